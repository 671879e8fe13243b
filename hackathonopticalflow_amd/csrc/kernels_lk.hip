// kernels_lk.hip -- sparse pyramidal Lucas-Kanade at a set of points:
//
//   cv2.calcOpticalFlowPyrLK(img2, img1, points_, None, winSize=(45, 45), maxLevel=2,
//                            criteria=(EPS | COUNT, 10, 0.03))               pathfinder_viewer.py:153-158
//
// restating OpenCV 4.10 lkpyramid.cpp / pyramids.cpp (see oracle/lk_oracle.c, which these kernels match bit
// for bit in its OFO_LK_SUM_COLUMNS order):
//   k_pyrdown_u8   pyrDown on uint8 (5x5 binomial, (sum + 128) >> 8, BORDER_REFLECT_101)
//   k_scharr       calcScharrDeriv: int16 (dI/dx, dI/dy), 3-10-3 Scharr, unscaled
//   k_lk_track     LKTrackerInvoker for one pyramid level: one wave per (pair, point), one lane per window
//                  column; the window of the first image (value and both derivatives, int16) lives in LDS,
//                  the float sums are taken per column top to bottom and the columns then left to right.
// Pyramid images are read with reflect-101 indices and the derivatives with a zero border, which is what the
// padded pyramid of buildOpticalFlowPyramid (BORDER_REFLECT_101 / BORDER_CONSTANT) holds around each level.
#include "farneback_device.h"
#include "ofarn_internal.h"

namespace ofarn {

__global__ __launch_bounds__(256) void k_pyrdown_u8(const uint8_t *__restrict__ src, int sw, int sh, uint8_t *__restrict__ dst,
                                                    int dw, int dh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const uint8_t *s = src + (size_t)blockIdx.z * sw * sh;
    int cx[5];
#pragma unroll
    for (int i = 0; i < 5; i++) cx[i] = reflect101(2 * x - 2 + i, sw);
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t *r = s + (size_t)reflect101(2 * y - 2 + j, sh) * sw;
        const int row = r[cx[0]] + r[cx[4]] + 4 * (r[cx[1]] + r[cx[3]]) + 6 * r[cx[2]];
        acc += (j == 0 || j == 4) ? row : (j == 2 ? 6 * row : 4 * row);
    }
    dst[(size_t)blockIdx.z * dw * dh + (size_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
}

void launch_pyrdown_u8(hipStream_t s, const uint8_t *src, int sw, int sh, uint8_t *dst, int nframes)
{
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, nframes);
    hipLaunchKernelGGL(k_pyrdown_u8, grid, dim3(256), 0, s, src, sw, sh, dst, dw, dh);
}

// frames z0, z0 + zstep, ... (only the frames that serve as the FIRST image of a pair need derivatives)
__global__ __launch_bounds__(256) void k_scharr(const uint8_t *__restrict__ src, int w, int h, short2 *__restrict__ dst, int z0,
                                                int zstep)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t fz = (size_t)z0 + (size_t)blockIdx.z * zstep;
    const uint8_t *s = src + fz * w * h;
    const uint8_t *r0 = s + (size_t)reflect101(y - 1, h) * w, *r1 = s + (size_t)y * w, *r2 = s + (size_t)reflect101(y + 1, h) * w;
    const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
    const int s_l = (r0[xl] + r2[xl]) * 3 + r1[xl] * 10, s_r = (r0[xr] + r2[xr]) * 3 + r1[xr] * 10;
    const int d_l = r2[xl] - r0[xl], d_c = r2[x] - r0[x], d_r = r2[xr] - r0[xr];
    dst[fz * w * h + (size_t)y * w + x] = make_short2((short)(s_r - s_l), (short)((d_r + d_l) * 3 + d_c * 10));
}

void launch_scharr(hipStream_t s, const uint8_t *src, int w, int h, int16_t *dst, int z0, int zstep, int count)
{
    if (count <= 0) return;
    dim3 grid((w + 63) / 64, (h + 3) / 4, count);
    hipLaunchKernelGGL(k_scharr, grid, dim3(256), 0, s, src, w, h, reinterpret_cast<short2 *>(dst), z0, zstep);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lk_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
constexpr int LK_CHUNK = 15;   // window rows whose loads are in flight together (45 = 3 x 15)

// float sum over the window columns in lane order (every lane computes the same value from LDS)
__device__ __forceinline__ float lk_colsum(const float *red, int ww)
{
    float s = 0.f;
    for (int x = 0; x < ww; x++) s += red[x];
    return s;
}

__global__ __launch_bounds__(64) void k_lk_track(LkLevelArgs A)
{
    extern __shared__ unsigned char lk_smem[];
    const int ww = A.win_w, wh = A.win_h;
    short *sI = reinterpret_cast<short *>(lk_smem);                  // [wh][ww]
    short2 *sD = reinterpret_cast<short2 *>(sI + ((ww * wh + 1) & ~1));   // [wh][ww]
    float *red = reinterpret_cast<float *>(sD + ww * wh);            // [3][64]
    const int lane = threadIdx.x;
    const bool act = lane < ww;
    const int pt = blockIdx.x, pair = blockIdx.y;
    const int cols = A.w, rows = A.h;
    const size_t npx = (size_t)cols * rows;
    const int fi = pair * A.fstep + A.i_off, fj = pair * A.fstep + A.j_off;
    const uint8_t *I = A.img + (size_t)fi * npx, *J = A.img + (size_t)fj * npx;
    const short2 *dI = reinterpret_cast<const short2 *>(A.deriv) + (size_t)fi * npx;
    const float2 p0 = reinterpret_cast<const float2 *>(A.pts)[(size_t)pair * A.pts_stride + pt];
    float2 *np_out = reinterpret_cast<float2 *>(A.next_pts) + (size_t)pair * A.npts + pt;
    uint8_t *st_out = A.status + (size_t)pair * A.npts + pt;
    float *err_out = A.err + (size_t)pair * A.npts + pt;

    const float hwx = (float)(ww - 1) * 0.5f, hwy = (float)(wh - 1) * 0.5f;
    const float sc = A.scale;                                       // (float)(1. / (1 << level))
    float ppx = p0.x * sc, ppy = p0.y * sc;
    float npx_, npy_;
    if (A.level == A.top_level) {
        if (A.flags & 4) { const float2 g = *np_out; npx_ = g.x * sc; npy_ = g.y * sc; }
        else { npx_ = ppx; npy_ = ppy; }
    } else { const float2 g = *np_out; npx_ = g.x * 2.f; npy_ = g.y * 2.f; }
    if (A.level == A.top_level && lane == 0) { *st_out = 1; *err_out = 0.f; }
    float2 result = make_float2(npx_, npy_);
    bool ok = true;   // status can only turn false at level 0, the last one to run
    auto finish = [&](bool status_now) {
        if (lane == 0) {
            *np_out = result;
            if (A.level == 0 && !status_now) *st_out = 0;
        }
    };

    ppx -= hwx; ppy -= hwy;
    const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
    if (ipx < -ww || ipx >= cols || ipy < -wh || ipy >= rows) {
        if (A.level == 0 && lane == 0) *err_out = 0.f;
        finish(false);
        return;
    }
    float a = ppx - (float)ipx, b = ppy - (float)ipy;
    const float wsc = 16384.f;
    int iw00 = __float2int_rn((1.f - a) * (1.f - b) * wsc);
    int iw01 = __float2int_rn(a * (1.f - b) * wsc);
    int iw10 = __float2int_rn((1.f - a) * b * wsc);
    int iw11 = 16384 - iw00 - iw01 - iw10;

    // ---- window of the first image: value and derivatives (int16) into LDS, covariance sums per column
    float a11 = 0.f, a12 = 0.f, a22 = 0.f;
    if (act) {
        const int X0 = ipx + lane, X1 = X0 + 1;
        const int cx0 = reflect101(X0, cols), cx1 = reflect101(X1, cols);
        const bool in0 = X0 >= 0 && X0 < cols, in1 = X1 >= 0 && X1 < cols;
        auto pix = [&](int Y, int &v0, int &v1) {
            const uint8_t *r = I + (size_t)reflect101(Y, rows) * cols;
            v0 = r[cx0]; v1 = r[cx1];
        };
        auto der = [&](int Y, short2 &d0, short2 &d1) {
            const bool yin = Y >= 0 && Y < rows;
            d0 = (yin && in0) ? dI[(size_t)Y * cols + X0] : make_short2(0, 0);
            d1 = (yin && in1) ? dI[(size_t)Y * cols + X1] : make_short2(0, 0);
        };
        for (int yb = 0; yb < wh; yb += LK_CHUNK) {
            int p0[LK_CHUNK + 1], p1[LK_CHUNK + 1];
            short2 q0[LK_CHUNK + 1], q1[LK_CHUNK + 1];
#pragma unroll
            for (int i = 0; i <= LK_CHUNK; i++) {
                pix(ipy + yb + i, p0[i], p1[i]);
                der(ipy + yb + i, q0[i], q1[i]);
            }
#pragma unroll
            for (int i = 0; i < LK_CHUNK; i++) {
                const int y = yb + i;
                if (y < wh) {
                    const int ival = lk_descale(p0[i] * iw00 + p1[i] * iw01 + p0[i + 1] * iw10 + p1[i + 1] * iw11, 9);
                    const int ixval = lk_descale(q0[i].x * iw00 + q1[i].x * iw01 + q0[i + 1].x * iw10 + q1[i + 1].x * iw11, 14);
                    const int iyval = lk_descale(q0[i].y * iw00 + q1[i].y * iw01 + q0[i + 1].y * iw10 + q1[i + 1].y * iw11, 14);
                    sI[y * ww + lane] = (short)ival;
                    sD[y * ww + lane] = make_short2((short)ixval, (short)iyval);
                    a11 += (float)(ixval * ixval);
                    a12 += (float)(ixval * iyval);
                    a22 += (float)(iyval * iyval);
                }
            }
        }
    }
    red[lane] = a11; red[64 + lane] = a12; red[128 + lane] = a22;
    __syncthreads();
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float A11 = lk_colsum(red, ww) * FLT_SCALE, A12 = lk_colsum(red + 64, ww) * FLT_SCALE, A22 = lk_colsum(red + 128, ww) * FLT_SCALE;
    __syncthreads();
    float D = A11 * A22 - A12 * A12;
    const float minEig = __fdiv_rn(A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12), (float)(2 * ww * wh));
    if ((A.flags & 8) && lane == 0) *err_out = minEig;
    if (minEig < A.min_eig || D < 1.1920929e-07f) {
        finish(false);
        return;
    }
    D = __fdiv_rn(1.f, D);
    npx_ -= hwx; npy_ -= hwy;
    float pdx = 0.f, pdy = 0.f;
    // window difference sums at (fx, fy) in the second image; mode 0: b1, b2 (float products), 1: sum |diff|
    auto window_pass = [&](float fx, float fy, int inx, int iny, int mode, float &o1, float &o2) {
        const float aa = fx - (float)inx, bb = fy - (float)iny;
        const int w00 = __float2int_rn((1.f - aa) * (1.f - bb) * wsc);
        const int w01 = __float2int_rn(aa * (1.f - bb) * wsc);
        const int w10 = __float2int_rn((1.f - aa) * bb * wsc);
        const int w11 = 16384 - w00 - w01 - w10;
        float s1 = 0.f, s2 = 0.f;
        if (act) {
            const int cx0 = reflect101(inx + lane, cols), cx1 = reflect101(inx + lane + 1, cols);
            // rows in chunks of LK_CHUNK: all the byte loads of a chunk are issued before its arithmetic, so a pass pays a
            // few memory round trips instead of one per window row
            for (int yb = 0; yb < wh; yb += LK_CHUNK) {
                int v0[LK_CHUNK + 1], v1[LK_CHUNK + 1];
#pragma unroll
                for (int i = 0; i <= LK_CHUNK; i++) {
                    const uint8_t *r = J + (size_t)reflect101(iny + yb + i, rows) * cols;
                    v0[i] = r[cx0];
                    v1[i] = r[cx1];
                }
#pragma unroll
                for (int i = 0; i < LK_CHUNK; i++) {
                    const int y = yb + i;
                    if (y < wh) {
                        const int diff = lk_descale(v0[i] * w00 + v1[i] * w01 + v0[i + 1] * w10 + v1[i + 1] * w11, 9) - (int)sI[y * ww + lane];
                        if (mode == 0) {
                            const short2 d = sD[y * ww + lane];
                            s1 += (float)(diff * (int)d.x);
                            s2 += (float)(diff * (int)d.y);
                        } else
                            s1 += fabsf((float)diff);
                    }
                }
            }
        }
        red[lane] = s1; red[64 + lane] = s2;
        __syncthreads();
        o1 = lk_colsum(red, ww);
        o2 = mode == 0 ? lk_colsum(red + 64, ww) : 0.f;
        __syncthreads();
    };
    for (int j = 0; j < A.max_count; j++) {
        const int inx = (int)floorf(npx_), iny = (int)floorf(npy_);
        if (inx < -ww || inx >= cols || iny < -wh || iny >= rows) { ok = false; break; }
        float b1, b2;
        window_pass(npx_, npy_, inx, iny, 0, b1, b2);
        b1 *= FLT_SCALE; b2 *= FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        npx_ += dx; npy_ += dy;
        result = make_float2(npx_ + hwx, npy_ + hwy);
        if ((double)dx * dx + (double)dy * dy <= A.eps2) break;
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
            result.x -= dx * 0.5f;
            result.y -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
    if (ok && A.level == 0 && (A.flags & 8) == 0) {
        const float fx = result.x - hwx, fy = result.y - hwy;
        const int inx = (int)floorf(fx), iny = (int)floorf(fy);
        if (inx < -ww || inx >= cols || iny < -wh || iny >= rows) ok = false;
        else {
            float e, unused;
            window_pass(fx, fy, inx, iny, 1, e, unused);
            if (lane == 0) *err_out = __fdiv_rn(e * 1.f, (float)(32 * ww * wh));
        }
    }
    finish(ok);
}

size_t lk_track_lds_bytes(int win_w, int win_h)
{
    const size_t n = (size_t)win_w * win_h;
    return ((n + 1) & ~(size_t)1) * sizeof(short) + n * sizeof(short2) + 3 * 64 * sizeof(float);
}

void launch_lk_track(hipStream_t s, const LkLevelArgs &A, int npairs)
{
    if (A.npts == 0 || npairs == 0) return;
    hipLaunchKernelGGL(k_lk_track, dim3(A.npts, npairs), dim3(64), lk_track_lds_bytes(A.win_w, A.win_h), s, A);
}

}  // namespace ofarn
