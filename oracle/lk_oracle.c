/*
 * lk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for the sparse path of the shipped
 * viewer:
 *
 *   cv2.calcOpticalFlowPyrLK(img2, img1, points_, None, winSize=(45, 45), maxLevel=2,
 *                            criteria=(TERM_CRITERIA_EPS | TERM_CRITERIA_COUNT, 10, 0.03))
 *       reference: pathfinder_viewer.py:153-158 (get_flow_lk), DenseOF.py:181-185, SparseOF.py:35-36
 *
 * The arithmetic is OpenCV 4.10 modules/video/src/lkpyramid.cpp (buildOpticalFlowPyramid, calcScharrDeriv,
 * LKTrackerInvoker) and modules/imgproc/src/pyramids.cpp (pyrDown, 8-bit), restated from memory.
 *
 * PARITY UNPINNED: cv2 is not installed, OpenCV's sources are not in /root/reference and the reference holds
 * no fixtures.  Pinned instead by tests/test_oracle_lk.py (pyrDown / Scharr against scipy closed forms,
 * translation ground truth, forward-backward consistency).
 *
 * Everything up to the tracker's sums is integer arithmetic and therefore order independent.  The sums
 * A11, A12, A22, b1, b2 are FLOAT accumulations of integer products over the window; their order is the one
 * recall-risk switch here (OpenCV's SSE/NEON paths keep four partial sums per row chunk, the scalar build adds
 * row by row):
 *   OFO_LK_SUM_SCALAR  row-major, one accumulator: the order of lkpyramid.cpp's scalar loop
 *   OFO_LK_SUM_COLUMNS one accumulator per window column (top to bottom), then the columns left to right:
 *                      the order the HIP kernel uses (one lane per window column)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OFO_API __attribute__((visibility("default")))

#define OFO_LK_SUM_SCALAR 0
#define OFO_LK_SUM_COLUMNS 1

#define OFO_LK_GET_MIN_EIGENVALS 8 /* cv2.OPTFLOW_LK_GET_MIN_EIGENVALS */
#define OFO_LK_USE_INITIAL_FLOW 4  /* cv2.OPTFLOW_USE_INITIAL_FLOW */

static int lk_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

/* pyrDown, CV_8U, BORDER_REFLECT_101: 5x5 [1 4 6 4 1]^2 / 256 with rounding, centred on (2x, 2y);
 * dst is ((sw+1)/2) x ((sh+1)/2). */
OFO_API void ofo_pyrdown_u8(const uint8_t *src, int sw, int sh, uint8_t *dst)
{
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; y++) {
        for (int j = 0; j < 5; j++) {
            const uint8_t *s = src + (size_t)lk_reflect101(2 * y - 2 + j, sh) * sw;
            int *r = rows + (size_t)j * dw;
            for (int x = 0; x < dw; x++) {
                int acc = 0;
                for (int i = 0; i < 5; i++) acc += k[i] * s[lk_reflect101(2 * x - 2 + i, sw)];
                r[x] = acc;
            }
        }
        for (int x = 0; x < dw; x++) {
            int acc = 0;
            for (int j = 0; j < 5; j++) acc += k[j] * rows[(size_t)j * dw + x];
            dst[(size_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
        }
    }
    free(rows);
}

/* calcScharrDeriv: dst int16 [h][w][2] = (dI/dx, dI/dy) with the 3-10-3 Scharr kernels, unscaled,
 * BORDER_REFLECT_101. */
OFO_API void ofo_scharr_deriv(const uint8_t *src, int w, int h, int16_t *dst)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = src + (size_t)lk_reflect101(y - 1, h) * w;
        const uint8_t *r1 = src + (size_t)y * w;
        const uint8_t *r2 = src + (size_t)lk_reflect101(y + 1, h) * w;
        for (int x = 0; x < w; x++) {
            const int xl = lk_reflect101(x - 1, w), xr = lk_reflect101(x + 1, w);
            /* trow0 = vertical smooth, trow1 = vertical difference */
            const int s_l = (r0[xl] + r2[xl]) * 3 + r1[xl] * 10, s_r = (r0[xr] + r2[xr]) * 3 + r1[xr] * 10;
            const int d_l = r2[xl] - r0[xl], d_c = r2[x] - r0[x], d_r = r2[xr] - r0[xr];
            dst[((size_t)y * w + x) * 2] = (int16_t)(s_r - s_l);
            dst[((size_t)y * w + x) * 2 + 1] = (int16_t)((d_r + d_l) * 3 + d_c * 10);
        }
    }
}

/* buildOpticalFlowPyramid: level i exists while both sides stay larger than the window. */
OFO_API int ofo_lk_levels(int w, int h, int win_w, int win_h, int max_level)
{
    int level = 0;
    while (level < max_level) {
        const int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win_w || nh <= win_h) break;
        w = nw;
        h = nh;
        level++;
    }
    return level;
}

typedef struct {
    int w, h;
    uint8_t *I, *J;  /* prev / next image of this level */
    int16_t *dI;     /* Scharr derivatives of I */
} lk_level;

static inline int img_at(const uint8_t *im, int w, int h, int x, int y) /* REFLECT_101 padding of the pyramid */
{
    return im[(size_t)lk_reflect101(y, h) * w + lk_reflect101(x, w)];
}
static inline int der_at(const int16_t *d, int w, int h, int x, int y, int c) /* BORDER_CONSTANT padding */
{
    if (x < 0 || x >= w || y < 0 || y >= h) return 0;
    return d[((size_t)y * w + x) * 2 + c];
}
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

/* float sum of n*m items in the selected order; item(y, x) supplied through arrays [h][w] */
static float lk_sum(const float *items, int ww, int wh, int mode)
{
    float s = 0.f;
    if (mode == OFO_LK_SUM_COLUMNS) {
        for (int x = 0; x < ww; x++) {
            float c = 0.f;
            for (int y = 0; y < wh; y++) c += items[(size_t)y * ww + x];
            s += c;
        }
    } else {
        for (int y = 0; y < wh; y++)
            for (int x = 0; x < ww; x++) s += items[(size_t)y * ww + x];
    }
    return s;
}

/* cv2.calcOpticalFlowPyrLK(prev, next, pts, ...).  pts/next_pts float[n][2]; status uint8[n]; err float[n].
 * criteria: max_count and epsilon as given (TERM_CRITERIA_COUNT | TERM_CRITERIA_EPS); flags: OFO_LK_*.
 * With OFO_LK_USE_INITIAL_FLOW next_pts holds the initial guesses on entry.  Returns 0, <0 on bad arguments. */
OFO_API int ofo_pyr_lk(const uint8_t *prev, const uint8_t *next, int W, int H, const float *pts, int n, int win_w, int win_h,
                       int max_level, int max_count, double epsilon, int flags, double min_eig_threshold, int sum_mode,
                       float *next_pts, uint8_t *status, float *err)
{
    if (!prev || !next || !pts || !next_pts || !status || !err) return -1;
    if (max_level < 0 || win_w <= 2 || win_h <= 2 || W < 1 || H < 1 || n < 0) return -2;
    max_count = max_count < 0 ? 0 : max_count > 100 ? 100 : max_count;
    epsilon = epsilon < 0 ? 0 : epsilon > 10 ? 10 : epsilon;
    epsilon *= epsilon;
    const int levels = ofo_lk_levels(W, H, win_w, win_h, max_level);
    lk_level *L = (lk_level *)calloc((size_t)levels + 1, sizeof(lk_level));
    for (int l = 0; l <= levels; l++) {
        L[l].w = l == 0 ? W : (L[l - 1].w + 1) / 2;
        L[l].h = l == 0 ? H : (L[l - 1].h + 1) / 2;
        const size_t npx = (size_t)L[l].w * L[l].h;
        L[l].I = (uint8_t *)malloc(npx);
        L[l].J = (uint8_t *)malloc(npx);
        L[l].dI = (int16_t *)malloc(npx * 2 * sizeof(int16_t));
        if (l == 0) {
            memcpy(L[l].I, prev, npx);
            memcpy(L[l].J, next, npx);
        } else {
            ofo_pyrdown_u8(L[l - 1].I, L[l - 1].w, L[l - 1].h, L[l].I);
            ofo_pyrdown_u8(L[l - 1].J, L[l - 1].w, L[l - 1].h, L[l].J);
        }
        ofo_scharr_deriv(L[l].I, L[l].w, L[l].h, L[l].dI);
    }
    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0.f; }
    const int area = win_w * win_h;
    int16_t *Iw = (int16_t *)malloc(sizeof(int16_t) * (size_t)area * 3);
    int16_t *dIw = Iw + area;            /* [area][2] */
    float *items = (float *)malloc(sizeof(float) * (size_t)area * 3);
    const float hwx = (win_w - 1) * 0.5f, hwy = (win_h - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = levels; level >= 0; level--) {
        const lk_level *lv = &L[level];
        const int cols = lv->w, rows = lv->h;
        for (int p = 0; p < n; p++) {
            const float sc = (float)(1. / (1 << level));
            float ppx = pts[2 * p] * sc, ppy = pts[2 * p + 1] * sc;
            float npx, npy;
            if (level == levels) {
                if (flags & OFO_LK_USE_INITIAL_FLOW) { npx = next_pts[2 * p] * sc; npy = next_pts[2 * p + 1] * sc; }
                else { npx = ppx; npy = ppy; }
            } else { npx = next_pts[2 * p] * 2.f; npy = next_pts[2 * p + 1] * 2.f; }
            next_pts[2 * p] = npx;
            next_pts[2 * p + 1] = npy;

            ppx -= hwx; ppy -= hwy;
            const int ipx = cv_floor_f(ppx), ipy = cv_floor_f(ppy);
            if (ipx < -win_w || ipx >= cols || ipy < -win_h || ipy >= rows) {
                if (level == 0) { status[p] = 0; err[p] = 0; }
                continue;
            }
            float a = ppx - ipx, b = ppy - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            for (int y = 0; y < win_h; y++)
                for (int x = 0; x < win_w; x++) {
                    const int X = ipx + x, Y = ipy + y;
                    const int ival = DESCALE(img_at(lv->I, cols, rows, X, Y) * iw00 + img_at(lv->I, cols, rows, X + 1, Y) * iw01 +
                                             img_at(lv->I, cols, rows, X, Y + 1) * iw10 + img_at(lv->I, cols, rows, X + 1, Y + 1) * iw11,
                                             W_BITS - 5);
                    const int ixval = DESCALE(der_at(lv->dI, cols, rows, X, Y, 0) * iw00 + der_at(lv->dI, cols, rows, X + 1, Y, 0) * iw01 +
                                              der_at(lv->dI, cols, rows, X, Y + 1, 0) * iw10 + der_at(lv->dI, cols, rows, X + 1, Y + 1, 0) * iw11,
                                              W_BITS);
                    const int iyval = DESCALE(der_at(lv->dI, cols, rows, X, Y, 1) * iw00 + der_at(lv->dI, cols, rows, X + 1, Y, 1) * iw01 +
                                              der_at(lv->dI, cols, rows, X, Y + 1, 1) * iw10 + der_at(lv->dI, cols, rows, X + 1, Y + 1, 1) * iw11,
                                              W_BITS);
                    const int o = y * win_w + x;
                    Iw[o] = (int16_t)ival;
                    dIw[2 * o] = (int16_t)ixval;
                    dIw[2 * o + 1] = (int16_t)iyval;
                    items[o] = (float)(ixval * ixval);
                    items[area + o] = (float)(ixval * iyval);
                    items[2 * area + o] = (float)(iyval * iyval);
                }
            const float A11 = lk_sum(items, win_w, win_h, sum_mode) * FLT_SCALE;
            const float A12 = lk_sum(items + area, win_w, win_h, sum_mode) * FLT_SCALE;
            const float A22 = lk_sum(items + 2 * area, win_w, win_h, sum_mode) * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win_w * win_h);
            if ((flags & OFO_LK_GET_MIN_EIGENVALS) != 0) err[p] = minEig;
            if (minEig < (float)min_eig_threshold || D < 1.1920929e-07f) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            npx -= hwx; npy -= hwy;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; j++) {
                const int inx = cv_floor_f(npx), iny = cv_floor_f(npy);
                if (inx < -win_w || inx >= cols || iny < -win_h || iny >= rows) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                a = npx - inx; b = npy - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                for (int y = 0; y < win_h; y++)
                    for (int x = 0; x < win_w; x++) {
                        const int X = inx + x, Y = iny + y, o = y * win_w + x;
                        const int diff = DESCALE(img_at(lv->J, cols, rows, X, Y) * iw00 + img_at(lv->J, cols, rows, X + 1, Y) * iw01 +
                                                 img_at(lv->J, cols, rows, X, Y + 1) * iw10 + img_at(lv->J, cols, rows, X + 1, Y + 1) * iw11,
                                                 W_BITS - 5) - Iw[o];
                        items[o] = (float)(diff * dIw[2 * o]);
                        items[area + o] = (float)(diff * dIw[2 * o + 1]);
                    }
                const float b1 = lk_sum(items, win_w, win_h, sum_mode) * FLT_SCALE;
                const float b2 = lk_sum(items + area, win_w, win_h, sum_mode) * FLT_SCALE;
                const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                npx += dx; npy += dy;
                next_pts[2 * p] = npx + hwx;
                next_pts[2 * p + 1] = npy + hwy;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabs(dx + pdx) < 0.01 && fabs(dy + pdy) < 0.01) {
                    next_pts[2 * p] -= dx * 0.5f;
                    next_pts[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[p] && level == 0 && (flags & OFO_LK_GET_MIN_EIGENVALS) == 0) {
                const float fx = next_pts[2 * p] - hwx, fy = next_pts[2 * p + 1] - hwy;
                const int inx = cv_floor_f(fx), iny = cv_floor_f(fy);
                if (inx < -win_w || inx >= cols || iny < -win_h || iny >= rows) { status[p] = 0; continue; }
                const float aa = fx - inx, bb = fy - iny;
                iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                for (int y = 0; y < win_h; y++)
                    for (int x = 0; x < win_w; x++) {
                        const int X = inx + x, Y = iny + y, o = y * win_w + x;
                        const int diff = DESCALE(img_at(lv->J, cols, rows, X, Y) * iw00 + img_at(lv->J, cols, rows, X + 1, Y) * iw01 +
                                                 img_at(lv->J, cols, rows, X, Y + 1) * iw10 + img_at(lv->J, cols, rows, X + 1, Y + 1) * iw11,
                                                 W_BITS - 5) - Iw[o];
                        items[o] = fabsf((float)diff);
                    }
                err[p] = lk_sum(items, win_w, win_h, sum_mode) * 1.f / (32 * win_w * win_h);
            }
        }
    }
    for (int l = 0; l <= levels; l++) { free(L[l].I); free(L[l].J); free(L[l].dI); }
    free(L); free(Iw); free(items);
    return 0;
}
