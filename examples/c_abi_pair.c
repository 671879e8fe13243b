/*
 * c_abi_pair.c -- the C-ABI of include/ofarn.h used from plain C, no Python and no torch: what a non-Python host would link.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_pair.c -o c_abi_pair -Lhackathonopticalflow_amd -lofarn \
 *       -Wl,-rpath,$PWD/hackathonopticalflow_amd
 *   ./c_abi_pair frames.raw W H flow.raw [levels]
 *
 * frames.raw: two uint8 frames of W x H back to back (prev, next).  flow.raw: float32 [H][W][2] written by the call that
 * replaces cv2.calcOpticalFlowFarneback(prev, next, None, 0.5, levels, 15, 3, 5, 1.2, 0) (DenseOF.py:147-156); the danger
 * points of pathfinder_viewer.py:159-176 on that flow are printed.  tests/test_gpu_parity.py runs it and compares flow.raw
 * bit for bit with the Python mirror's result.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ofarn.h"

static int die(const char *what, int rc)
{
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, ofarn_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s frames.raw W H flow.raw [levels]\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    if (w < 1 || h < 1) return 2;
    const size_t npx = (size_t)w * h;
    uint8_t *frames = (uint8_t *)malloc(2 * npx);
    float *flow = (float *)malloc(npx * 2 * sizeof(float));
    FILE *f = fopen(argv[1], "rb");
    if (!frames || !flow || !f || fread(frames, 1, 2 * npx, f) != 2 * npx) {
        fprintf(stderr, "cannot read two %dx%d frames from %s\n", w, h, argv[1]);
        return 2;
    }
    fclose(f);

    ofarn_params prm;
    ofarn_default_params(&prm);                 /* DenseOF.py:127-128 defaults */
    if (argc > 5) prm.levels = atoi(argv[5]);
    ofarn_ctx *ctx = NULL;
    int rc = ofarn_create(&prm, 0, w, h, 1, &ctx);
    if (rc) return die("ofarn_create", rc);
    rc = ofarn_calc(ctx, frames, frames + npx, w, h, w, flow);
    if (rc) return die("ofarn_calc", rc);
    printf("%s: %dx%d levels=%d device time %.3f ms\n", ofarn_version(), w, h, prm.levels, ofarn_last_device_ms(ctx));

    const int P = ofarn_grid_points(w, h, prm.grid_step, NULL);
    if (P > 0) {
        uint8_t *mask = (uint8_t *)calloc((size_t)P, 1), *v = (uint8_t *)calloc((size_t)P, 1);
        float *pts = (float *)malloc((size_t)P * 2 * sizeof(float));
        ofarn_grid_points(w, h, prm.grid_step, pts);
        rc = ofarn_grid_filter(ctx, flow, 1, w, h, mask, v, NULL);
        if (rc) return die("ofarn_grid_filter", rc);
        int kept = 0;
        for (int i = 0; i < P; i++) kept += mask[i];
        printf("danger points: %d of %d grid points", kept, P);
        for (int i = 0, shown = 0; i < P && shown < 4; i++)
            if (mask[i]) { printf("  (%g,%g V=%d)", pts[2 * i], pts[2 * i + 1], v[i]); shown++; }
        printf("\n");
        free(mask); free(v); free(pts);
    }
    f = fopen(argv[4], "wb");
    if (!f || fwrite(flow, sizeof(float), npx * 2, f) != npx * 2) {
        fprintf(stderr, "cannot write %s\n", argv[4]);
        return 2;
    }
    fclose(f);
    ofarn_destroy(ctx);
    free(frames); free(flow);
    return 0;
}
