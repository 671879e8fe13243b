/*
 * frontend_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for the steps either side
 * of the dense-flow call:
 *
 *   ofo_bgr2gray        cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) on uint8 frames (DenseOF.py:481,510)
 *   ofo_resize_area     cv::resize(..., INTER_AREA) on float images, the step OPTFLOW_USE_INITIAL_FLOW
 *                       applies to the caller's flow at the coarsest scale (optflowgf.cpp calc())
 *   ofo_hsv2bgr_u8      cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) on uint8 (DenseOF.py:121, draw_hsv)
 *
 * PARITY UNPINNED: these restate OpenCV 4.10 (imgproc/src/color_rgb.simd.hpp, color_hsv.simd.hpp,
 * resize.cpp) from memory; cv2 is not installed, OpenCV's sources are not in /root/reference and the
 * reference holds no fixtures for them.  What is checked instead is listed in
 * tests/test_oracle_frontend.py.  Recall-risk switches are the named constants below.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OFO_API __attribute__((visibility("default")))

/* ---- BGR -> gray --------------------------------------------------------------------------------
 * color_rgb.simd.hpp, RGB2Gray<uchar>: fixed point with 15 fractional bits since OpenCV 3.4.2 / 4.0
 * (RY15, GY15, BY15; gray_shift = 15); earlier releases used 14 bits (R2Y, G2Y, B2Y; yuv_shift).
 * The vector path (v_dotprod on int16 pairs + rounding shift) computes the same integers. */
#define OFO_GRAY_VARIANT_15BIT 0
#define OFO_GRAY_VARIANT_14BIT 1
static const int k_gray15[3] = {3735, 19235, 9798}; /* B, G, R; sum 32768 */
static const int k_gray14[3] = {1868, 9617, 4899};  /* B, G, R; sum 16384 */

OFO_API void ofo_gray_coeffs(int variant, int *cb, int *cg, int *cr, int *shift)
{
    const int *c = variant == OFO_GRAY_VARIANT_14BIT ? k_gray14 : k_gray15;
    *cb = c[0];
    *cg = c[1];
    *cr = c[2];
    *shift = variant == OFO_GRAY_VARIANT_14BIT ? 14 : 15;
}

/* bgr: uint8 [h][stride] with 3 bytes per pixel (B, G, R); gray: uint8 [h][w] dense. */
OFO_API void ofo_bgr2gray(const uint8_t *bgr, int w, int h, int stride, int variant, uint8_t *gray)
{
    int cb, cg, cr, shift;
    ofo_gray_coeffs(variant, &cb, &cg, &cr, &shift);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = bgr + (size_t)y * stride;
        uint8_t *d = gray + (size_t)y * w;
        for (int x = 0; x < w; x++, s += 3) /* CV_DESCALE(b*cb + g*cg + r*cr, shift) */
            d[x] = (uint8_t)((s[0] * cb + s[1] * cg + s[2] * cr + (1 << (shift - 1))) >> shift);
    }
}

/* ---- resize INTER_AREA, float, cn channels --------------------------------------------------------
 * resize.cpp.  Shrinking by an integer factor in both directions takes ResizeAreaFast_Invoker
 * (plain sum over the scale_x*scale_y cell times 1/area; the 2x2 vector form exists for 1, 3 and 4
 * channels only, so 2-channel flow always takes the scalar loop); otherwise
 * computeResizeAreaTab + ResizeArea_Invoker (row accumulation `buf += S*alpha` in table order, then
 * `sum += beta*buf` over the rows of the cell).  Only shrinking (dst <= src in both directions) is
 * restated: that is the only use on this path. */
#define OFO_AREA_FAST_UNROLL4 1 /* CV_ENABLE_UNROLLED: sums taken four at a time (a+b+c+d added to the running sum) */

typedef struct {
    int si, di;
    float alpha;
} ofo_dec_alpha;

static int area_tab(int ssize, int dsize, double scale, ofo_dec_alpha *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cell = fmin(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx;
            tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cell);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx;
            tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cell);
        }
        if (fsx2 - sx2 > 1e-3) {
            tab[k].di = dx;
            tab[k].si = sx2;
            tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell);
        }
    }
    return k;
}

/* Exposes the decimation table (the device side builds its own; tests compare them). */
OFO_API int ofo_area_tab(int ssize, int dsize, int cap, int *si, int *di, float *alpha)
{
    double scale = 1. / ((double)dsize / ssize);
    ofo_dec_alpha *tab = (ofo_dec_alpha *)malloc(sizeof(ofo_dec_alpha) * ((size_t)ssize * 2 + 2));
    int n = area_tab(ssize, dsize, scale, tab);
    for (int i = 0; i < n && i < cap; i++) {
        si[i] = tab[i].si;
        di[i] = tab[i].di;
        alpha[i] = tab[i].alpha;
    }
    free(tab);
    return n;
}

/* returns 0 ok, -1 unsupported (enlarging) */
OFO_API int ofo_resize_area(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    if (dw > sw || dh > sh || dw < 1 || dh < 1) return -1;
    /* resize(): inv_scale = dsize/ssize, scale = 1/inv_scale (not ssize/dsize: the two differ in the last bit
     * for some ratios, which decides between the fast and the general branch) */
    double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    int iscale_x = (int)(scale_x > 0 ? floor(scale_x + 0.5) : ceil(scale_x - 0.5)); /* saturate_cast<int>: cvRound */
    int iscale_y = (int)floor(scale_y + 0.5);
    /* cvRound is round-half-even; scale >= 1 here and x.5 cannot be within DBL_EPSILON of an integer, so floor(x+.5) decides the same */
    int fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (fast && iscale_x == 1 && iscale_y == 1) {
        memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return 0;
    }
    if (fast) {
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / area;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    const float *S = src + ((size_t)dy * iscale_y * sw + (size_t)dx * iscale_x) * cn + c;
                    float sum = 0;
                    int k = 0;
#if OFO_AREA_FAST_UNROLL4
                    for (; k <= area - 4; k += 4) {
                        const float *p0 = S + ((size_t)(k / iscale_x) * sw + (k % iscale_x)) * cn;
                        const float *p1 = S + ((size_t)((k + 1) / iscale_x) * sw + ((k + 1) % iscale_x)) * cn;
                        const float *p2 = S + ((size_t)((k + 2) / iscale_x) * sw + ((k + 2) % iscale_x)) * cn;
                        const float *p3 = S + ((size_t)((k + 3) / iscale_x) * sw + ((k + 3) % iscale_x)) * cn;
                        sum += *p0 + *p1 + *p2 + *p3;
                    }
#endif
                    for (; k < area; k++)
                        sum += S[((size_t)(k / iscale_x) * sw + (k % iscale_x)) * cn];
                    const float v = sum * scale;
                    dst[((size_t)dy * dw + dx) * cn + c] = v;
                }
        return 0;
    }
    ofo_dec_alpha *xtab = (ofo_dec_alpha *)malloc(sizeof(ofo_dec_alpha) * ((size_t)sw * 2 + 2));
    ofo_dec_alpha *ytab = (ofo_dec_alpha *)malloc(sizeof(ofo_dec_alpha) * ((size_t)sh * 2 + 2));
    const int nx = area_tab(sw, dw, scale_x, xtab);
    const int ny = area_tab(sh, dh, scale_y, ytab);
    float *buf = (float *)malloc(sizeof(float) * (size_t)dw * cn * 2);
    float *sum = buf + (size_t)dw * cn;
    for (int i = 0; i < dw * cn; i++) sum[i] = 0.f;
    int prev_dy = ytab[0].di;
    for (int j = 0; j < ny; j++) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di;
        const float *S = src + (size_t)ytab[j].si * sw * cn;
        for (int i = 0; i < dw * cn; i++) buf[i] = 0.f;
        for (int k = 0; k < nx; k++) {
            const float alpha = xtab[k].alpha;
            for (int c = 0; c < cn; c++)
                buf[xtab[k].di * cn + c] = buf[xtab[k].di * cn + c] + S[xtab[k].si * cn + c] * alpha;
        }
        if (dy != prev_dy) {
            float *D = dst + (size_t)prev_dy * dw * cn;
            for (int i = 0; i < dw * cn; i++) {
                D[i] = sum[i];
                sum[i] = beta * buf[i];
            }
            prev_dy = dy;
        } else {
            for (int i = 0; i < dw * cn; i++) sum[i] += beta * buf[i];
        }
    }
    {
        float *D = dst + (size_t)prev_dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) D[i] = sum[i];
    }
    free(buf);
    free(xtab);
    free(ytab);
    return 0;
}

/* ---- HSV -> BGR, uint8 ----------------------------------------------------------------------------
 * color_hsv.simd.hpp, HSV2RGB_b over HSV2RGB_native with hrange = 180: h, s/255, v/255 in float, sector
 * table, result * 255 rounded to nearest even (saturate_cast<uchar> = cvRound). */
static inline uint8_t sat_u8_round(float v)
{
    long r = lrintf(v); /* round half to even under the default rounding mode, as cvRound */
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

OFO_API void ofo_hsv2bgr_u8(const uint8_t *hsv, size_t npx, uint8_t *bgr)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 180.f;
    for (size_t i = 0; i < npx; i++) {
        float h = hsv[3 * i], s = hsv[3 * i + 1] * (1.f / 255.f), v = hsv[3 * i + 2] * (1.f / 255.f);
        float b, g, r;
        if (s == 0)
            b = g = r = v;
        else {
            float tab[4];
            h *= hscale;
            h = fmodf(h, 6.f);
            int sector = (int)floorf(h);
            h -= sector;
            if ((unsigned)sector >= 6u) {
                sector = 0;
                h = 0.f;
            }
            tab[0] = v;
            tab[1] = v * (1.f - s);
            tab[2] = v * (1.f - s * h);
            tab[3] = v * (1.f - s * (1.f - h));
            b = tab[sector_data[sector][0]];
            g = tab[sector_data[sector][1]];
            r = tab[sector_data[sector][2]];
        }
        bgr[3 * i] = sat_u8_round(b * 255.f);
        bgr[3 * i + 1] = sat_u8_round(g * 255.f);
        bgr[3 * i + 2] = sat_u8_round(r * 255.f);
    }
}
