/*
 * filter_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's own NumPy code around the flow call:
 *   measurement grid          pathfinder_viewer.py:252-267
 *   vector filter             pathfinder_viewer.py:159-176   (get_flow_lk, after the LK call)
 *   danger brightness V       pathfinder_viewer.py:204-217   (draw_sparse_lamps)
 * The NumPy lines themselves are re-typed in oracle/oracle.py (numpy IS importable here), so
 * this C twin is pinned against real NumPy float32/float64 semantics in tests/test_oracle_filter.py:
 * the mask is IEEE-only arithmetic (+,*,/,sqrt) and must match bit for bit.  The integer flow and V go
 * through arctan2 / cos / sin: NumPy's float32 loops for those are SIMD approximations (measured here: up to
 * 3.2 ulp for arctan2, CPU dependent), so the contract the HIP kernels are held to is the CORRECTLY ROUNDED
 * float32 value -- evaluated in double by the host libm and rounded once (OFO_CR_* below).  How often that
 * differs from this container's NumPy after the integer truncation is measured in tests/test_oracle_filter.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OFO_API __attribute__((visibility("default")))

/* correctly rounded float32 arctan2 / cos / sin (double evaluation, one rounding) */
static float ofo_cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
static float ofo_cr_cosf(float a) { return (float)cos((double)a); }
static float ofo_cr_sinf(float a) { return (float)sin((double)a); }

/* pathfinder_viewer.py:255-262: indent and np.mgrid[indent:size:step].astype(int) */
static int ofo_axis_points(int size, int step, int *out)
{
    double indent = ((size / step) % 2 == 1) ? (size % step) / 2.0 : ((size % step) + step) / 2.0;
    int n = (int)ceil((size - indent) / (step * 1.0));
    if (n < 0) n = 0;
    if (out) for (int i = 0; i < n; i++) out[i] = (int)(i * (double)step + indent);
    return n;
}

/* Returns P = nx*ny; pts (if not NULL) receives float32 (x,y) pairs, x-major (pathfinder_viewer.py:263-267). */
OFO_API int ofo_grid_points(int width, int height, int step, float *pts)
{
    int nx = ofo_axis_points(width, step, NULL), ny = ofo_axis_points(height, step, NULL);
    if (pts) {
        int *xs = (int *)malloc(sizeof(int) * (size_t)(nx + 1)), *ys = (int *)malloc(sizeof(int) * (size_t)(ny + 1));
        ofo_axis_points(width, step, xs); ofo_axis_points(height, step, ys);
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++) {
                pts[(i * ny + j) * 2] = (float)xs[i];
                pts[(i * ny + j) * 2 + 1] = (float)ys[j];
            }
        free(xs); free(ys);
    }
    return nx * ny;
}

static int cmp_f32(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* np.median(float32[P]) -> float32: mean of the two middle order statistics for even P
 * (float32 add, then /2), the middle one for odd P. */
static float ofo_median_f32(const float *sorted, int P)
{
    if (P & 1) return sorted[P / 2];
    float s = sorted[P / 2 - 1] + sorted[P / 2];
    return s / 2.0f;
}

/* np.percentile(float32[P], q) -> float32, method 'linear' (numpy 2.x lib/_function_base_impl.py:
 * percentile() divides q by a.dtype.type(100), so for float32 data EVERYTHING stays float32:
 *   quant = float32(q)/float32(100);  vi = float32(P-1)*quant;  t = vi - floor(vi);
 *   _lerp: diff = b-a;  a + diff*t, rewritten as b - diff*(1-t) where t >= 0.5. */
static float ofo_percentile_f32(const float *sorted, int P, float q)
{
    float quant = q / 100.0f;
    float vi = (float)(P - 1) * quant;
    float prevf = floorf(vi);
    long pi = (long)prevf, ni = pi + 1;
    if (vi >= (float)(P - 1)) { pi = P - 1; ni = P - 1; }   /* _get_indexes: above bounds -> last */
    if (vi < 0) { pi = 0; ni = 0; }
    if (ni > P - 1) ni = P - 1;
    float t = vi - prevf;
    float a = sorted[pi], b = sorted[ni];
    float diff = b - a;
    if (t >= 0.5f) return b - diff * (1.0f - t);
    return a + diff * t;
}

/* vectors float32[P][2] (fx,fy) sampled at pts float32[P][2] (x,y).
 * Outputs (each may be NULL): mask u8[P]; modulus float32[P] (equalised);
 * iflow int32[P][2] = next_pts - points for EVERY point (callers apply the mask);
 * v u8[P] = danger brightness for kept points, 0 elsewhere; thr[2] = {median, p99}. */
/* variant 0: pathfinder_viewer.py:173 (median*1.0 < mod) & (mod < P99);  1: DenseOF.py:228 mod > median*1.2 */
OFO_API int ofo_vector_filter2(const float *vec, const float *pts, int P, int width, int height, int variant,
                               uint8_t *mask, float *modulus_out, int32_t *iflow, uint8_t *v, double *thr)
{
    if (P <= 0) return 0;
    const int half_width = (int)(width / 2.0), half_height = (int)(height / 2.0);
    float *mod = (float *)malloc(sizeof(float) * (size_t)P);
    float *srt = (float *)malloc(sizeof(float) * (size_t)P);
    for (int i = 0; i < P; i++) {
        float fx = vec[i * 2], fy = vec[i * 2 + 1];
        float x = pts[i * 2], y = pts[i * 2 + 1];
        float ang = ofo_cr_atan2f(fy, fx);
        float m = sqrtf(fx * fx + fy * fy);
        float ddx = (float)half_width - x, ddy = (float)half_height - y;
        float mm = sqrtf(ddx * ddx + ddy * ddy);
        m = m / (5.0f + sqrtf(mm)) * 30.0f;
        mod[i] = m;
        if (iflow) {
            float gx = m * ofo_cr_cosf(ang), gy = m * ofo_cr_sinf(ang);
            int32_t nx = (int32_t)((x + gx) + 0.5f), ny = (int32_t)((y + gy) + 0.5f);
            int32_t px = (int32_t)(x + 0.5f), py = (int32_t)(y + 0.5f);
            iflow[i * 2] = nx - px; iflow[i * 2 + 1] = ny - py;
        }
    }
    memcpy(srt, mod, sizeof(float) * (size_t)P);
    int has_nan = 0;
    for (int i = 0; i < P; i++) if (mod[i] != mod[i]) has_nan = 1;
    float med, p99;
    if (has_nan) med = p99 = NAN;       /* np.median / np.percentile return NaN when the data hold one */
    else {
        qsort(srt, (size_t)P, sizeof(float), cmp_f32);
        med = ofo_median_f32(srt, P) * 1.0f;
        p99 = ofo_percentile_f32(srt, P, 99.0f);
    }
    if (thr) { thr[0] = med; thr[1] = p99; }
    for (int i = 0; i < P; i++) {
        int keep = variant == 1 ? (mod[i] > med * 1.2f) : ((med < mod[i]) && (mod[i] < p99));
        if (mask) mask[i] = (uint8_t)keep;
        if (v) {
            uint8_t val = 0;
            if (keep && iflow) {
                int32_t a = iflow[i * 2], b = iflow[i * 2 + 1];
                double mlen = sqrt((double)(a * a + b * b));
                double vv = 50 + mlen * 2; if (vv > 255) vv = 255;
                val = (uint8_t)vv;
            }
            v[i] = val;
        }
    }
    if (modulus_out) memcpy(modulus_out, mod, sizeof(float) * (size_t)P);
    free(srt); free(mod);
    return 0;
}

OFO_API int ofo_vector_filter(const float *vec, const float *pts, int P, int width, int height,
                              uint8_t *mask, float *modulus_out, int32_t *iflow, uint8_t *v, double *thr)
{
    return ofo_vector_filter2(vec, pts, P, width, height, 0, mask, modulus_out, iflow, v, thr);
}
