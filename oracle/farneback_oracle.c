/*
 * farneback_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic behind the one call the reference
 * makes on its dense path:
 *
 *     cv2.calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize,
 *                                  iterations, poly_n, poly_sigma, flags)
 *     reference: DenseOF.py:127-157 (wrapper), DenseOF.py:520 (call site).
 *
 * The arithmetic itself is not in /root/reference: it lives in the pip
 * dependency opencv-python~=4.10.0.84 (requirements.txt:1), i.e. OpenCV 4.10.0
 *   modules/video/src/optflowgf.cpp      (FarnebackOpticalFlowImpl::calc,
 *                                         FarnebackPrepareGaussian, FarnebackPolyExp,
 *                                         FarnebackUpdateMatrices, FarnebackUpdateFlow_Blur)
 *   modules/imgproc/src/smooth.dispatch.cpp (getGaussianKernel / GaussianBlur)
 *   modules/imgproc/src/resize.cpp          (INTER_LINEAR on CV_32F)
 * None of those files, and no cv2 wheel, exist in the build container, so this
 * file restates OpenCV's published algorithm following the
 * scalar C++ code paths as recalled: same float/double mix, same operation order,
 * no FMA contraction (build with -ffp-contract=off).  Points where the recollection
 * is uncertain are named switches below.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
 * this path and OpenCV cannot be run here.  What pins this file instead are
 * independent closed-form / scipy / numpy checks in tests/test_oracle_*.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (hackathonopticalflow_amd) never does.
 *
 * Three summation orders are offered for the winsize x winsize box average
 * (OFO_BOX_RUNNING / OFO_BOX_DIRECT / OFO_BOX_BLOCKED); see ofo_update_flow_blur().
 *
 * Known nuance that is NOT restated: for a level exactly half the frame in both directions
 * cv::resize() swaps INTER_LINEAR for the 2x2 INTER_AREA fast path.  Its vector form,
 * ((a+b)+(c+d))*0.25, rounds exactly like the bilinear form with weights 1/2 used here, but the
 * scalar tail of a row (the last (W/2) mod 4 columns) sums ((a+b)+c)+d.  Frame widths with
 * (W/2) mod 4 == 0 (640, 1920, 3840) are unaffected.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OFO_API __attribute__((visibility("default")))

/* ---- recall-risk switches, kept as named constants ---- */
#define OFO_LEVEL_LOOP_INCLUSIVE 1   /* k = levels..0 inclusive -> levels+1 scales   */
#define OFO_MIN_SIZE 32              /* optflowgf.cpp: const int min_size = 32       */
#define OFO_DET_EPS 1e-3             /* idet = 1/(g11*g22 - g12*g12 + 1e-3)          */
#define OFO_BORDER 5
static const float ofo_border_tab[OFO_BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};

/* GaussianBlur row pass of a SMALL symmetric float kernel.  filter.simd.hpp getLinearRowFilter() hands a symmetrical
 * CV_32F kernel with ksize <= 5 to SymmRowSmallFilter<float, float>, whose arithmetic is
 *     ksize 3:  D = S[0]*k0 + (S[-1] + S[1])*k1
 *     ksize 5:  D = S[0]*k0 + (S[-1] + S[1])*k1 + (S[-2] + S[2])*k2          (k0 = centre tap)
 * not the left-to-right sum of the general RowFilter used for wider kernels.  1 = that order (default since round 3),
 * 0 = plain left to right for every size (rounds 1-2).  Level 0 ([1/4, 1/2, 1/4] on byte values) is exact either way;
 * a level with sigma = 0.5 (ksize 3) or a ksize-5 level rounds differently.  Run-time switch: ofo_set_row_small_symm().
 * (Real wheels may differ further in the last bit: the AVX2 dispatch of these filters uses v_muladd = FMA, and builds
 * with IPP route GaussianBlur through ippiFilterGaussian.  Neither is restated: parity is unpinned, see above.) */
#define OFO_ROW_SMALL_SYMM 1
static int ofo_row_small_symm = OFO_ROW_SMALL_SYMM;

#define OFO_BOX_RUNNING 0   /* OpenCV's literal order: double running sums, float row differences */
#define OFO_BOX_DIRECT  1   /* same window, each sum taken directly in a fixed order                */
#define OFO_BOX_BLOCKED 2   /* column sums as block-restarted running sums (the HIP kernels' order)  */

typedef struct ofo_params {
    double pyr_scale;
    int levels;
    int winsize;
    int iterations;
    int poly_n;
    double poly_sigma;
    int flags;
} ofo_params;

/* cvRound(double): round-half-to-even under the default rounding mode (lrint). */
static inline int ofo_cvround(double v) { return (int)lrint(v); }
/* cvFloor(float) */
static inline int ofo_cvfloor(float v) { int i = (int)v; return i - (i > v); }

static inline int ofo_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------- */
/* A.2 level geometry                                                         */
/* ------------------------------------------------------------------------- */

/* Number of pyramid reductions actually used (optflowgf.cpp: the "levels = k" crop loop). */
OFO_API int ofo_crop_levels(int W, int H, double pyr_scale, int levels)
{
    int k; double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (W * scale < OFO_MIN_SIZE || H * scale < OFO_MIN_SIZE) break;
    }
    return k;
}

OFO_API void ofo_level_geom(int W, int H, double pyr_scale, int k,
                            int *w, int *h, double *sigma, int *ksize)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= pyr_scale;
    double s = (1. / scale - 1) * 0.5;
    int sz = ofo_cvround(s * 5) | 1;
    if (sz < 3) sz = 3;
    *sigma = s; *ksize = sz;
    *w = ofo_cvround(W * scale);
    *h = ofo_cvround(H * scale);
}

/* ------------------------------------------------------------------------- */
/* A.3 level image: convertTo(CV_32F) + GaussianBlur + resize(INTER_LINEAR)   */
/* ------------------------------------------------------------------------- */

/* getGaussianKernel(n, sigma, CV_32F): fixed table for sigma<=0 and odd n<=7,
 * else exp(-x^2/(2 sigma^2)) in double, normalised in double, cast to float.
 * (OpenCV 4.x evaluates the symmetric half and mirrors it; so does this.) */
OFO_API int ofo_gaussian_kernel(int n, double sigma, float *out)
{
    static const float tab[4][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    if (n <= 0) return -1;
    if ((n & 1) && n <= 7 && sigma <= 0) {
        for (int i = 0; i < n; i++) out[i] = tab[n >> 1][i];
        return 0;
    }
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        v[i] = exp(scale2X * x * x);
        sum += v[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) out[i] = (float)(v[i] * sum);
    free(v);
    return 0;
}

OFO_API void ofo_set_row_small_symm(int on) { ofo_row_small_symm = on != 0; }
OFO_API int ofo_get_row_small_symm(void) { return ofo_row_small_symm; }

/* Separable filter on float32, BORDER_REFLECT_101, row pass then column pass.
 * Order of operations per output sample (OpenCV's scalar filter engine):
 *   row    : s = kx[0]*S[0]; s += kx[k]*S[k]            (k = 1..n-1, left to right; ksize <= 5: OFO_ROW_SMALL_SYMM above)
 *   column : s = ky[r]*S[0]; s += ky[r+k]*(S[+k]+S[-k]) (k = 1..r, centre outwards)
 */
OFO_API void ofo_gaussian_blur(const float *src, int W, int H, int ksize, double sigma, float *dst)
{
    float *kx = (float *)malloc(sizeof(float) * (size_t)ksize);
    ofo_gaussian_kernel(ksize, sigma, kx);
    int r = ksize / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
    int *xi = (int *)malloc(sizeof(int) * (size_t)(W + 2 * r));
    for (int x = -r; x < W + r; x++) xi[x + r] = ofo_reflect101(x, W);
    float *ext = (float *)malloc(sizeof(float) * (size_t)(W + 2 * r));
    for (int y = 0; y < H; y++) {
        const float *s = src + (size_t)y * W;
        float *t = tmp + (size_t)y * W;
        for (int x = 0; x < W + 2 * r; x++) ext[x] = s[xi[x]];
        if (ofo_row_small_symm && ksize == 3) {          /* SymmRowSmallFilter, symmetrical, ksize == 3 */
            const float k0 = kx[r], k1 = kx[r + 1];
            for (int x = 0; x < W; x++) { const float *S = ext + x + r; t[x] = S[0] * k0 + (S[-1] + S[1]) * k1; }
            continue;
        }
        if (ofo_row_small_symm && ksize == 5) {          /* ksize == 5 */
            const float k0 = kx[r], k1 = kx[r + 1], k2 = kx[r + 2];
            for (int x = 0; x < W; x++) {
                const float *S = ext + x + r;
                t[x] = S[0] * k0 + (S[-1] + S[1]) * k1 + (S[-2] + S[2]) * k2;
            }
            continue;
        }
        for (int x = 0; x < W; x++) t[x] = kx[0] * ext[x];
        for (int k = 1; k < ksize; k++) {
            const float f = kx[k];
            const float *e = ext + k;
            for (int x = 0; x < W; x++) t[x] = t[x] + f * e[x];
        }
    }
    for (int y = 0; y < H; y++) {
        float *d = dst + (size_t)y * W;
        const float *c = tmp + (size_t)y * W;
        const float f0 = kx[r];
        for (int x = 0; x < W; x++) d[x] = f0 * c[x];
        for (int k = 1; k <= r; k++) {
            const float f = kx[r + k];
            const float *a = tmp + (size_t)ofo_reflect101(y + k, H) * W;
            const float *b = tmp + (size_t)ofo_reflect101(y - k, H) * W;
            for (int x = 0; x < W; x++) d[x] = d[x] + f * (a[x] + b[x]);
        }
    }
    free(ext); free(xi); free(tmp); free(kx);
}

/* resize(..., INTER_LINEAR) on CV_32F, cn interleaved channels.
 *   fx = (float)((dx+0.5)*scale_x - 0.5); sx = floor(fx); fx -= sx;
 *   sx < 0 -> sx = 0, fx = 0;  sx >= sw-1 -> sx = sw-1, fx = 0
 * horizontal pass  D = S[sx]*(1-fx) + S[sx+1]*fx, then vertical pass b0*row0 + b1*row1.
 * (For exact 2x decimation OpenCV switches to the 2x2-mean fast path,
 *  ((a+b)+(c+d))*0.25, which this formula reproduces bit for bit: all weights are 0.5.) */
OFO_API void ofo_resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    float *xa = (float *)malloc(sizeof(float) * (size_t)dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = ofo_cvfloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx; xa[dx] = fx;
    }
    float *row0 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    float *row1 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = ofo_cvfloor(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        const float *s0 = src + (size_t)sy * sw * cn;
        const float *s1 = src + (size_t)sy1 * sw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            float a1 = xa[dx], a0 = 1.f - a1;
            for (int c = 0; c < cn; c++) {
                row0[dx * cn + c] = s0[sx * cn + c] * a0 + s0[sx1 * cn + c] * a1;
                row1[dx * cn + c] = s1[sx * cn + c] * a0 + s1[sx1 * cn + c] * a1;
            }
        }
        float b1 = fy, b0 = 1.f - fy;
        float *d = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) d[i] = row0[i] * b0 + row1[i] * b1;
    }
    free(row1); free(row0); free(xa); free(xofs);
}

/* One pyramid level image of one frame (optflowgf.cpp calc(): convertTo, GaussianBlur, resize). */
OFO_API void ofo_level_image(const uint8_t *img, int W, int H, int stride,
                             int ksize, double sigma, int w, int h, float *out)
{
    float *f = (float *)malloc(sizeof(float) * (size_t)W * H);
    float *b = (float *)malloc(sizeof(float) * (size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) f[(size_t)y * W + x] = (float)img[(size_t)y * stride + x];
    ofo_gaussian_blur(f, W, H, ksize, sigma, b);
    if (w == W && h == H) memcpy(out, b, sizeof(float) * (size_t)W * H);
    else ofo_resize_linear(b, W, H, 1, out, w, h);
    free(b); free(f);
}

/* ------------------------------------------------------------------------- */
/* A.4 polynomial expansion                                                   */
/* ------------------------------------------------------------------------- */

/* 6x6 inverse by Gauss-Jordan with partial pivoting, double.  OpenCV uses
 * DECOMP_CHOLESKY; both are exact to ~1e-16 relative on this well-conditioned matrix. */
static void ofo_inv6(double A[6][6], double inv[6][6])
{
    double a[6][12];
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) { a[i][j] = A[i][j]; a[i][j + 6] = (i == j); }
    }
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (p != c) for (int j = 0; j < 12; j++) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        double d = 1. / a[c][c];
        for (int j = 0; j < 12; j++) a[c][j] *= d;
        for (int r = 0; r < 6; r++) if (r != c) {
            double f = a[r][c];
            if (f != 0) for (int j = 0; j < 12; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) inv[i][j] = a[i][j + 6];
}

/* FarnebackPrepareGaussian.  g, xg, xxg point at the CENTRE of arrays of 2n+1 floats.
 * ig = {ig11, ig03, ig33, ig55}. */
OFO_API void ofo_poly_prepare(int n, double sigma, float *g, float *xg, float *xxg, double *ig)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[6][6], invG[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    ofo_inv6(G, invG);
    ig[0] = invG[1][1]; ig[1] = invG[0][3]; ig[2] = invG[3][3]; ig[3] = invG[5][5];
}

/* FarnebackPolyExp: I float[h][w] -> R float[h][w][5] (interleaved, OpenCV channel order:
 * 0: y-linear, 1: x-linear, 2: yy, 3: xx, 4: xy).  Replicate borders.
 * The vertical pass is float32; the horizontal accumulators b1..b6 are double, but the
 * products feeding b2,b3,b5,b6 are float*float (as written in optflowgf.cpp), only
 * tg*g0 and tg*xxg[k] are double products. */
OFO_API void ofo_polyexp(const float *I, int w, int h, int n, double sigma, float *R)
{
    float *kbuf = (float *)malloc(sizeof(float) * (size_t)(n * 6 + 3));
    float *g = kbuf + n, *xg = g + n * 2 + 1, *xxg = xg + n * 2 + 1;
    double ig[4];
    ofo_poly_prepare(n, sigma, g, xg, xxg, ig);
    const double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
    float *_row = (float *)malloc(sizeof(float) * (size_t)(w + n * 2) * 3);
    float *row = _row + n * 3;

    for (int y = 0; y < h; y++) {
        float g0 = g[0], g1, g2;
        const float *srow0 = I + (size_t)y * w, *srow1 = 0;
        float *drow = R + (size_t)y * w * 5;

        for (int x = 0; x < w; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = I + (size_t)(y - k > 0 ? y - k : 0) * w;
            srow1 = I + (size_t)(y + k < h - 1 ? y + k : h - 1) * w;
            for (int x = 0; x < w; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
            }
        }
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[w * 3 + x] = row[w * 3 + x - 3];
        }
        for (int x = 0; x < w; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0,
                   b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0; b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(_row); free(kbuf);
}

/* ------------------------------------------------------------------------- */
/* A.5 FarnebackUpdateMatrices                                                */
/* ------------------------------------------------------------------------- */
OFO_API void ofo_update_matrices(const float *R0_, const float *R1, const float *flow_, float *M_,
                                 int width, int height, int y0, int y1)
{
    const size_t step1 = (size_t)width * 5;
    for (int y = y0; y < y1; y++) {
        const float *flow = flow_ + (size_t)y * width * 2;
        const float *R0 = R0_ + (size_t)y * width * 5;
        float *M = M_ + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = ofo_cvfloor(fx), yy1 = ofo_cvfloor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= yy1;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)yy1 < (unsigned)(height - 1)) {
                const float *ptr = R1 + (size_t)yy1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                      a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - OFO_BORDER) >= (unsigned)(width - OFO_BORDER * 2) ||
                (unsigned)(y - OFO_BORDER) >= (unsigned)(height - OFO_BORDER * 2)) {
                float scale = (x < OFO_BORDER ? ofo_border_tab[x] : 1.f) *
                              (x >= width - OFO_BORDER ? ofo_border_tab[width - x - 1] : 1.f) *
                              (y < OFO_BORDER ? ofo_border_tab[y] : 1.f) *
                              (y >= height - OFO_BORDER ? ofo_border_tab[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* A.6 FarnebackUpdateFlow_Blur                                               */
/* ------------------------------------------------------------------------- */
static inline void ofo_solve(double g11, double g12, double g22, double h1, double h2,
                             double scale, float *out)
{
    double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
    double h1_ = h1 * scale, h2_ = h2 * scale;
    double idet = 1. / (g11_ * g22_ - g12_ * g12_ + OFO_DET_EPS);
    out[0] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
    out[1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
}

/* box_mode OFO_BOX_RUNNING: statement-by-statement optflowgf.cpp: vsum[] is a double running
 *   sum down the image, updated with the FLOAT difference srow1[x]-srow0[x]; the horizontal
 *   window is a double running sum along the row.
 * box_mode OFO_BOX_DIRECT: the same (2m+1)x(2m+1) replicate-border window, but every sum is
 *   taken directly: column sum = sum_{j=-m..m} (double)M[clamp(y+j)][x] in that order, then
 *   window sum = sum_{i=-m..m} colsum[clamp(x+i)] in that order.
 * box_mode OFO_BOX_BLOCKED: like OpenCV the column sums are RUNNING sums in double (add the row
 *   entering the window, subtract the row leaving it), but restarted every B = 2m+1 rows so that a
 *   GPU strip can start anywhere without the history of the rows above it, and without OpenCV's
 *   float rounding of the row difference.  With padded rows r' = 0..h+2m-1, v[r'] =
 *   (double)M[clamp(r'-m)][x], and aligned blocks b = [B*b, B*b+B-1]:
 *       P_b(j) = v[Bb] + v[Bb+1] + ... + v[Bb+j]          accumulated left to right
 *       T_b    = P_b(B-1)
 *       S_b(0) = T_b;  S_b(j) = S_b(j-1) - v[Bb+j-1]       (what is left of block b from offset j)
 *       colsum(y) = T_b                    if y = B*b
 *                 = S_b(j) + P_{b+1}(j-1)  if y = B*b + j, 0 < j < B
 *   The horizontal pass adds the 2m+1 column sums in chunks of three, left to right:
 *       window = ((e0+e1)+e2) + ((e3+e4)+e5) + ...   (a shorter last chunk if 3 does not divide 2m+1)
 *   -- the fused kernel forms the chunk sums once per column with two whole-wave register shifts and
 *   reads five of them per window instead of fifteen column sums.  This is the order the HIP kernels
 *   use, so device results can be compared bit for bit.
 * OpenCV re-runs UpdateMatrices on row stripes as soon as the blur has passed them; a stripe is
 * only rewritten after its last reader, so that schedule equals the two-phase form used here. */
OFO_API void ofo_update_flow_blur(const float *R0, const float *R1, float *flow_, float *M,
                                  int width, int height, int block_size, int update_matrices,
                                  int box_mode)
{
    int m = block_size / 2;
    double scale = 1. / (block_size * block_size);
    double *_vsum = (double *)malloc(sizeof(double) * (size_t)(width + m * 2 + 2) * 5);
    double *vsum = _vsum + (m + 1) * 5;

    if (box_mode == OFO_BOX_RUNNING) {
        const float *srow0 = M;
        for (int x = 0; x < width * 5; x++) vsum[x] = srow0[x] * (m + 2);
        for (int y = 1; y < m; y++) {
            srow0 = M + (size_t)(y < height - 1 ? y : height - 1) * width * 5;
            for (int x = 0; x < width * 5; x++) vsum[x] += srow0[x];
        }
        for (int y = 0; y < height; y++) {
            double g11, g12, g22, h1, h2;
            float *flow = flow_ + (size_t)y * width * 2;
            srow0 = M + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * width * 5;
            const float *srow1 = M + (size_t)(y + m < height - 1 ? y + m : height - 1) * width * 5;
            for (int x = 0; x < width * 5; x++) vsum[x] += srow1[x] - srow0[x];
            for (int x = 0; x < (m + 1) * 5; x++) {
                vsum[-1 - x] = vsum[4 - x];
                vsum[width * 5 + x] = vsum[width * 5 + x - 5];
            }
            g11 = vsum[0] * (m + 2); g12 = vsum[1] * (m + 2); g22 = vsum[2] * (m + 2);
            h1 = vsum[3] * (m + 2); h2 = vsum[4] * (m + 2);
            for (int x = 1; x < m; x++) {
                g11 += vsum[x * 5]; g12 += vsum[x * 5 + 1]; g22 += vsum[x * 5 + 2];
                h1 += vsum[x * 5 + 3]; h2 += vsum[x * 5 + 4];
            }
            for (int x = 0; x < width; x++) {
                g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
                g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
                g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
                h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
                h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
                ofo_solve(g11, g12, g22, h1, h2, scale, flow + x * 2);
            }
        }
    } else {
        /* per column-channel state of the blocked running sums */
        const int B = 2 * m + 1;
        double *P = NULL, *S = NULL;
        if (box_mode == OFO_BOX_BLOCKED) {
            P = (double *)calloc((size_t)width * 5, sizeof(double));
            S = (double *)calloc((size_t)width * 5, sizeof(double));
        }
        for (int y = 0; y < height; y++) {
            float *flow = flow_ + (size_t)y * width * 2;
            if (box_mode == OFO_BOX_DIRECT) {
                const float *srow = M + (size_t)(y - m < 0 ? 0 : y - m) * width * 5;
                for (int x = 0; x < width * 5; x++) vsum[x] = (double)srow[x];
                for (int j = -m + 1; j <= m; j++) {
                    int yy = y + j; if (yy < 0) yy = 0; if (yy > height - 1) yy = height - 1;
                    srow = M + (size_t)yy * width * 5;
                    for (int x = 0; x < width * 5; x++) vsum[x] += (double)srow[x];
                }
            } else {
                /* padded rows arrive one per output row; output y is emitted when padded row
                 * t = y + B - 1 has arrived.  Rows 0..B-2 are consumed before the first output. */
                for (int t = (y == 0 ? 0 : y + B - 1); t <= y + B - 1; t++) {
                    const int j = t % B;                       /* offset of row t in its block */
                    int rnew = t - m; if (rnew < 0) rnew = 0; if (rnew > height - 1) rnew = height - 1;
                    int rold = t - B - m; if (rold < 0) rold = 0; if (rold > height - 1) rold = height - 1;
                    const float *snew = M + (size_t)rnew * width * 5;
                    const float *sold = M + (size_t)rold * width * 5;   /* row t-B: same offset, previous block */
                    for (int x = 0; x < width * 5; x++) {
                        const double vn = (double)snew[x];
                        P[x] = j == 0 ? vn : P[x] + vn;
                        if (j == B - 1) { vsum[x] = P[x]; S[x] = P[x]; }
                        else {
                            if (t >= B) S[x] = S[x] - (double)sold[x];
                            vsum[x] = S[x] + P[x];
                        }
                    }
                }
            }
            for (int x = 0; x < width; x++) {
                double s[5];
                if (box_mode == OFO_BOX_BLOCKED) {
                    /* chunks of three columns, left to right: ((e0+e1)+e2) + ((e3+e4)+e5) + ... */
                    for (int c = 0; c < 5; c++) {
                        for (int i0 = 0; i0 < B; i0 += 3) {
                            double ch = 0;
                            for (int i = i0; i < i0 + 3 && i < B; i++) {
                                int xx = x - m + i; if (xx < 0) xx = 0; if (xx > width - 1) xx = width - 1;
                                ch = i == i0 ? vsum[xx * 5 + c] : ch + vsum[xx * 5 + c];
                            }
                            s[c] = i0 == 0 ? ch : s[c] + ch;
                        }
                    }
                } else {
                    int xl = x - m < 0 ? 0 : x - m;
                    for (int c = 0; c < 5; c++) s[c] = vsum[xl * 5 + c];
                    for (int i = -m + 1; i <= m; i++) {
                        int xx = x + i; if (xx < 0) xx = 0; if (xx > width - 1) xx = width - 1;
                        for (int c = 0; c < 5; c++) s[c] += vsum[xx * 5 + c];
                    }
                }
                ofo_solve(s[0], s[1], s[2], s[3], s[4], scale, flow + x * 2);
            }
        }
        free(P); free(S);
    }
    free(_vsum);
    if (update_matrices) ofo_update_matrices(R0, R1, flow_, M, width, height, 0, height);
}

/* FarnebackUpdateFlow_GaussianBlur (flags & OPTFLOW_FARNEBACK_GAUSSIAN): the window is a separable
 * Gaussian, sigma = m*0.3, m = winsize/2, all in float32 (only the 2x2 solve is double):
 *   kernel[0] = 1, kernel[i] = (float)exp(-i*i/(2 sigma^2)); s = 1 + 2*sum (double); kernel[i] = (float)(kernel[i]/s... *(1/s))
 *   column: s0 = M[y][x]*k[0]; s0 += (M[min(y+i,h-1)][x] + M[max(y-i,0)][x]) * k[i]       (i = 1..m)
 *   row   : sum = v[x]*k[0];   sum += k[i] * (v[x-i] + v[x+i])   (replicate border)      (i = 1..m)
 * followed by the same regularised solve.  Two-phase like ofo_update_flow_blur. */
OFO_API void ofo_update_flow_gaussian(const float *R0, const float *R1, float *flow_, float *M,
                                      int width, int height, int block_size, int update_matrices)
{
    const int m = block_size / 2;
    double sigma = m * 0.3, s = 1;
    float *kernel = (float *)malloc(sizeof(float) * (size_t)(m + 1));
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);
    float *_vsum = (float *)malloc(sizeof(float) * (size_t)(width + m * 2 + 2) * 5);
    float *vsum = _vsum + (m + 1) * 5;
    float *hsum = (float *)malloc(sizeof(float) * (size_t)width * 5);
    for (int y = 0; y < height; y++) {
        float *flow = flow_ + (size_t)y * width * 2;
        const float *c = M + (size_t)y * width * 5;
        for (int x = 0; x < width * 5; x++) {
            float s0 = c[x] * kernel[0];
            for (int i = 1; i <= m; i++) {
                const float *a = M + (size_t)(y + i < height - 1 ? y + i : height - 1) * width * 5;
                const float *b = M + (size_t)(y - i > 0 ? y - i : 0) * width * 5;
                s0 += (a[x] + b[x]) * kernel[i];
            }
            vsum[x] = s0;
        }
        for (int x = 0; x < m * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        for (int x = 0; x < width * 5; x++) {
            float sum = vsum[x] * kernel[0];
            for (int i = 1; i <= m; i++) sum += kernel[i] * (vsum[x - i * 5] + vsum[x + i * 5]);
            hsum[x] = sum;
        }
        for (int x = 0; x < width; x++) {
            double g11 = hsum[x * 5], g12 = hsum[x * 5 + 1], g22 = hsum[x * 5 + 2];
            double h1 = hsum[x * 5 + 3], h2 = hsum[x * 5 + 4];
            double idet = 1. / (g11 * g22 - g12 * g12 + OFO_DET_EPS);
            flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
    }
    free(hsum); free(_vsum); free(kernel);
    if (update_matrices) ofo_update_matrices(R0, R1, flow_, M, width, height, 0, height);
}

/* ------------------------------------------------------------------------- */
/* A.2 the level loop                                                         */
/* ------------------------------------------------------------------------- */

/* Optional per-level capture for stage-by-stage parity tests.  Any pointer may be NULL.
 * Arrays are indexed by level k (0 = full resolution); the caller sizes them from ofo_level_geom. */
typedef struct ofo_capture {
    float **I0, **I1;     /* [k] -> float[h][w]      level images           */
    float **R0, **R1;     /* [k] -> float[h][w][5]   polynomial expansions  */
    float **M_first;      /* [k] -> float[h][w][5]   M before iteration 0   */
    float **flow_init;    /* [k] -> float[h][w][2]   flow entering level k  */
    float **flow_out;     /* [k] -> float[h][w][2]   flow leaving level k   */
} ofo_capture;

int ofo_resize_area(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh);   /* frontend_oracle.c */

OFO_API int ofo_farneback_ex(const uint8_t *prev, const uint8_t *next, int W, int H, int stride,
                             const ofo_params *p, int box_mode, float *flow0, const ofo_capture *cap)
{
    if (!prev || !next || !flow0 || !p) return -1;
    if (!(p->pyr_scale < 1) || W <= 0 || H <= 0) return -2;
    if (p->flags & ~(256 | 4)) return -3;   /* OPTFLOW_FARNEBACK_GAUSSIAN | OPTFLOW_USE_INITIAL_FLOW */
    /* winsize < 2 gives m = 0, for which optflowgf.cpp's running-sum initialisation ((m+2) copies
     * of row 0) no longer describes a window at all (it yields M[y][x]+M[0][x]+M[y][0]+M[0][0]);
     * that artefact is not restated. */
    if (p->winsize < 2 || p->iterations < 0 || p->poly_n < 1 || p->levels < 0) return -4;
    const uint8_t *img[2] = {prev, next};
    int levels = ofo_crop_levels(W, H, p->pyr_scale, p->levels);
    float *prevFlow = NULL; int pw = 0, ph = 0;

    for (int k = OFO_LEVEL_LOOP_INCLUSIVE ? levels : levels - 1; k >= 0; k--) {
        int width, height, smooth_sz; double sigma;
        ofo_level_geom(W, H, p->pyr_scale, k, &width, &height, &sigma, &smooth_sz);
        size_t npx = (size_t)width * height;
        float *flow = k > 0 ? (float *)malloc(sizeof(float) * npx * 2) : flow0;
        if (!prevFlow && (p->flags & 4)) {
            /* OPTFLOW_USE_INITIAL_FLOW: resize(flow0, flow, Size(width, height), 0, 0, INTER_AREA); flow *= scale;
             * flow0 is the caller's in/out buffer (read here at the coarsest level, written at level 0).
             * Same size (k == 0): resize() copies onto itself and scale is 1. */
            if (k > 0) {
                double scale = 1;
                for (int i = 0; i < k; i++) scale *= p->pyr_scale;
                if (ofo_resize_area(flow0, W, H, 2, flow, width, height)) return -5;
                const float mul = (float)scale;
                for (size_t i = 0; i < npx * 2; i++) flow[i] = flow[i] * mul;
            }
        } else if (!prevFlow) memset(flow, 0, sizeof(float) * npx * 2);
        else {
            ofo_resize_linear(prevFlow, pw, ph, 2, flow, width, height);
            /* "flow *= 1./pyrScale_" on CV_32F is convertTo(-1, alpha): elem * (float)alpha */
            const float mul = (float)(1. / p->pyr_scale);
            for (size_t i = 0; i < npx * 2; i++) flow[i] = flow[i] * mul;
        }
        if (cap && cap->flow_init && cap->flow_init[k]) memcpy(cap->flow_init[k], flow, sizeof(float) * npx * 2);

        float *R[2];
        float *I = (float *)malloc(sizeof(float) * npx);
        for (int i = 0; i < 2; i++) {
            R[i] = (float *)malloc(sizeof(float) * npx * 5);
            ofo_level_image(img[i], W, H, stride, smooth_sz, sigma, width, height, I);
            if (cap) {
                float **dstI = i == 0 ? cap->I0 : cap->I1;
                if (dstI && dstI[k]) memcpy(dstI[k], I, sizeof(float) * npx);
            }
            ofo_polyexp(I, width, height, p->poly_n, p->poly_sigma, R[i]);
            if (cap) {
                float **dstR = i == 0 ? cap->R0 : cap->R1;
                if (dstR && dstR[k]) memcpy(dstR[k], R[i], sizeof(float) * npx * 5);
            }
        }
        free(I);
        float *M = (float *)malloc(sizeof(float) * npx * 5);
        ofo_update_matrices(R[0], R[1], flow, M, width, height, 0, height);
        if (cap && cap->M_first && cap->M_first[k]) memcpy(cap->M_first[k], M, sizeof(float) * npx * 5);
        for (int i = 0; i < p->iterations; i++) {
            if (p->flags & 256)   /* OPTFLOW_FARNEBACK_GAUSSIAN */
                ofo_update_flow_gaussian(R[0], R[1], flow, M, width, height, p->winsize, i < p->iterations - 1);
            else
                ofo_update_flow_blur(R[0], R[1], flow, M, width, height, p->winsize,
                                     i < p->iterations - 1, box_mode);
        }
        if (cap && cap->flow_out && cap->flow_out[k]) memcpy(cap->flow_out[k], flow, sizeof(float) * npx * 2);
        free(M); free(R[0]); free(R[1]);
        if (prevFlow) free(prevFlow);
        prevFlow = flow; pw = width; ph = height;
        if (k == 0) prevFlow = NULL;   /* flow0 is caller memory */
    }
    if (prevFlow) free(prevFlow);
    return 0;
}

OFO_API int ofo_farneback(const uint8_t *prev, const uint8_t *next, int W, int H, int stride,
                          const ofo_params *p, int box_mode, float *flow0)
{
    return ofo_farneback_ex(prev, next, W, H, stride, p, box_mode, flow0, NULL);
}

/* Batch driver for the CPU baseline: frames u8[n_frames][H][W]; pairs_mode 0: pair i = frames
 * (2i, 2i+1); pairs_mode 1: pair i = frames (i, i+1) (video order, DenseOF.py:525).
 * OpenMP across pairs when built with -fopenmp; nthreads <= 0 means omp default. */
OFO_API int ofo_farneback_batch(const uint8_t *frames, int n_pairs, int pairs_mode, int W, int H,
                                const ofo_params *p, int box_mode, float *flow, int nthreads)
{
    int rc = 0;
    size_t fsz = (size_t)W * H;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n_pairs; i++) {
        const uint8_t *a = frames + (pairs_mode ? (size_t)i : (size_t)2 * i) * fsz;
        const uint8_t *b = a + fsz;
        int r = ofo_farneback(a, b, W, H, W, p, box_mode, flow + (size_t)i * fsz * 2);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

OFO_API int ofo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
