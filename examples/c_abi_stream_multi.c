/*
 * c_abi_stream_multi.c -- the two call patterns of round 3 from plain C (no Python, no torch):
 *
 *   1. the reference's frame LOOP (DenseOF.py:491-525: one new frame per turn, prev_gray = gray) through ofarn_stream_next,
 *      with the flow written into page-locked memory from ofarn_host_alloc;
 *   2. the same frames as a BATCH of consecutive pairs sharded over G GPUs of this process through ofarn_multi_calc_batch
 *      (one context + host thread + stream per device, one RCCL all-gather of the danger maps).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_stream_multi.c -o c_abi_stream_multi -Lhackathonopticalflow_amd -lofarn \
 *       -Wl,-rpath,$PWD/hackathonopticalflow_amd
 *   ./c_abi_stream_multi frames.raw W H N out_prefix [levels] [n_gpus]
 *
 * frames.raw: N uint8 frames of W x H.  Writes <out_prefix>.stream.raw and <out_prefix>.multi.raw (float32 [N-1][H][W][2] each)
 * and <out_prefix>.mask.raw (uint8 [N-1][P], the gathered danger masks); exit code 0 only if the two flow files are identical.
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

static int dump(const char *prefix, const char *suffix, const void *p, size_t bytes)
{
    char name[1024];
    snprintf(name, sizeof name, "%s%s", prefix, suffix);
    FILE *f = fopen(name, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", name); return 1; }
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: %s frames.raw W H N out_prefix [levels] [n_gpus]\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]), n = atoi(argv[4]);
    if (w < 1 || h < 1 || n < 2) return 2;
    const size_t npx = (size_t)w * h, fbytes = npx * 2 * sizeof(float);
    uint8_t *frames = (uint8_t *)malloc((size_t)n * npx);
    FILE *f = fopen(argv[1], "rb");
    if (!frames || !f || fread(frames, 1, (size_t)n * npx, f) != (size_t)n * npx) {
        fprintf(stderr, "cannot read %d %dx%d frames from %s\n", n, w, h, argv[1]);
        return 2;
    }
    fclose(f);
    ofarn_params prm;
    ofarn_default_params(&prm);
    if (argc > 6) prm.levels = atoi(argv[6]);
    const int gpus = argc > 7 ? atoi(argv[7]) : 1;
    float *all_stream = (float *)malloc((size_t)(n - 1) * fbytes), *all_multi = (float *)malloc((size_t)(n - 1) * fbytes);
    if (!all_stream || !all_multi) return 2;

    /* 1. the frame loop */
    ofarn_ctx *ctx = NULL;
    int rc = ofarn_create(&prm, 0, w, h, 1, &ctx);
    if (rc) return die("ofarn_create", rc);
    void *pinned = NULL;
    if ((rc = ofarn_host_alloc(fbytes, &pinned))) return die("ofarn_host_alloc", rc);
    double ms = 0;
    for (int i = 0; i < n; i++) {
        rc = ofarn_stream_next(ctx, frames + (size_t)i * npx, w, h, w, (float *)pinned);
        if (rc < 0) return die("ofarn_stream_next", rc);
        if (i == 0 && rc != OFARN_STREAM_PRIMED) { fprintf(stderr, "first frame should prime the session\n"); return 1; }
        if (i > 0) {
            if (rc != OFARN_OK) { fprintf(stderr, "turn %d returned %d\n", i, rc); return 1; }
            memcpy(all_stream + (size_t)(i - 1) * npx * 2, pinned, fbytes);
            ms += ofarn_last_device_ms(ctx);
        }
    }
    printf("%s: frame loop, %d turns of %dx%d, levels=%d: %.3f ms of device time per turn\n", ofarn_version(), n - 1, w, h, prm.levels,
           ms / (n - 1));
    /* 1b. the same loop with the PAIR signature of the reference's call (DenseOF.py:519-525: flow = calculate_optical_flow(prev_gray,
     * gray); prev_gray = gray): ofarn_calc_reuse is stateless in effect -- it reuses the frame on the device only when every byte of
     * `prev` equals the frame it was last given as `next`.  Here: the loop (reused from the second call on), then the held frame
     * edited in place by one byte (noticed: not reused, and the flow is that of the edited frame = ofarn_calc's). */
    {
        int reused = 0, hits = 0;
        for (int i = 1; i < n; i++) {
            if ((rc = ofarn_calc_reuse(ctx, frames + (size_t)(i - 1) * npx, frames + (size_t)i * npx, w, h, w, w, (float *)pinned, &reused)))
                return die("ofarn_calc_reuse", rc);
            hits += reused;
            if (memcmp(all_stream + (size_t)(i - 1) * npx * 2, pinned, fbytes) != 0) { fprintf(stderr, "ofarn_calc_reuse differs from the loop at pair %d\n", i - 1); return 1; }
        }
        if (hits != n - 2) { fprintf(stderr, "expected %d reuses, got %d\n", n - 2, hits); return 1; }
        uint8_t *edited = (uint8_t *)malloc(npx);
        float *ref = (float *)malloc(fbytes);
        if (!edited || !ref) return 2;
        memcpy(edited, frames + (size_t)(n - 1) * npx, npx);        /* the frame the device holds now ... */
        edited[npx / 2 + 3] ^= 0x40;                                  /* ... with one byte changed */
        if ((rc = ofarn_calc_reuse(ctx, edited, frames, w, h, w, w, (float *)pinned, &reused))) return die("ofarn_calc_reuse", rc);
        if ((rc = ofarn_calc(ctx, edited, frames, w, h, w, ref))) return die("ofarn_calc", rc);
        if (reused || memcmp(ref, pinned, fbytes) != 0) { fprintf(stderr, "an edited frame was reused or gave another flow than ofarn_calc\n"); return 1; }
        unsigned long long h_ = 0, m_ = 0;
        ofarn_calc_reuse_info(ctx, &h_, &m_);
        printf("pair signature with exact reuse: %llu reused, %llu started over (one of them after a one-byte edit)\n", h_, m_);
        free(edited); free(ref);
    }
    ofarn_host_free(pinned);
    ofarn_destroy(ctx);

    /* 2. the same pairs as one batch over `gpus` devices */
    const int P = ofarn_grid_points(w, h, prm.grid_step, NULL);
    uint8_t *mask = (uint8_t *)calloc((size_t)(n - 1) * (P > 0 ? P : 1), 1), *v = (uint8_t *)calloc((size_t)(n - 1) * (P > 0 ? P : 1), 1);
    ofarn_multi *multi = NULL;
    if ((rc = ofarn_multi_create(&prm, NULL, gpus, w, h, 8, &multi))) return die("ofarn_multi_create", rc);
    if ((rc = ofarn_multi_calc_batch(multi, frames, n, w, h, OFARN_PAIRS_CONSECUTIVE, all_multi, mask, v)))
        return die("ofarn_multi_calc_batch", rc);
    int ver = 0;
    unsigned long long calls = 0;
    ofarn_multi_info(multi, &ver, &calls, &ms);
    long kept = 0;
    for (size_t i = 0; i < (size_t)(n - 1) * (size_t)(P > 0 ? P : 0); i++) kept += mask[i];
    printf("batch of %d consecutive pairs over %d device(s): RCCL %d, %llu ncclAllGather calls, %ld danger points kept\n", n - 1,
           ofarn_multi_device_count(multi), ver, calls, kept);
    for (int g = 0; g < gpus; g++) {
        int s0 = 0, cnt = 0;
        ofarn_shard_pairs(n - 1, g, gpus, &s0, &cnt);
        printf("  rank %d: pairs [%d, %d)\n", g, s0, s0 + cnt);
    }
    ofarn_multi_destroy(multi);

    const int same = memcmp(all_stream, all_multi, (size_t)(n - 1) * fbytes) == 0;
    printf("stream and multi-GPU batch flows %s\n", same ? "identical" : "DIFFER");
    if (dump(argv[5], ".stream.raw", all_stream, (size_t)(n - 1) * fbytes) || dump(argv[5], ".multi.raw", all_multi, (size_t)(n - 1) * fbytes) ||
        dump(argv[5], ".mask.raw", mask, (size_t)(n - 1) * (size_t)(P > 0 ? P : 0)))
        return 2;
    free(frames); free(all_stream); free(all_multi); free(mask); free(v);
    return same ? 0 : 1;
}
