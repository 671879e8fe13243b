// kernels_tile.hip -- one Farneback iteration (FarnebackUpdateMatrices + FarnebackUpdateFlow_Blur, optionally the flow upsample)
// for SMALL grids: a single pair, or the coarse levels of a small batch.
//
// The marching kernel (k_flow_iter, kernels_fast.hip) is built for throughput: a block walks down at least 2m+1 + 2m rows, one
// row after the other.  When a level is a few thousand pixels and there is one pair, a handful of such blocks is all there is on
// the chip, every one of them runs its 29 rows at one wave per SIMD (a row is ~270 VALU instructions at ~5 cycles of issue latency
// each), and the launch takes 26-31 us whatever the level's size: 18 of these launches are 0.56 of the 0.70 ms a 1080p pair takes.
// Here the same arithmetic is laid out for latency: a block owns a tile of TW x (2m+1) output pixels, computes the matrices of the
// (TW + 2m) x (2(2m+1) - 1) pixels its window sums need with one thread per pixel (all gathers of the tile in flight at once),
// keeps them in LDS, and sums them in the SAME order as the marching kernel and the oracle's OFO_BOX_BLOCKED:
//   columns: block-restarted running sums in double, blocks of 2m+1 padded rows aligned at padded row 0
//            T = v[0] + ... + v[B-1] (left to right), S(j) = S(j-1) - v[j-1], P'(j) = v[B] + ... + v[B+j-1], colsum(j) = S(j) + P'(j)
//   rows:    the 2m+1 column sums in chunks of three, left to right
// so the results are bit-identical to the other two paths (tests/test_gpu_parity.py runs the pipeline tests on both).
#include "flow_iter_common.h"
#include "ofarn_internal.h"

#include <cstdlib>

namespace ofarn {

constexpr int TI_TW = 32;   // output columns per tile

template <int M_, int MODE>   // MODE 0: flow_in == 0;  1: flow_in = upsample(coarse)*mul;  2: flow_in from HBM
__global__ __launch_bounds__(256) void k_flow_iter_tile(const float *__restrict__ R, int fstep, const float2 *__restrict__ flow_in,
                                                        float2 *__restrict__ flow_out, int w, int h, double scale, UpsampleArgs up)
{
    constexpr int B = 2 * M_ + 1, IW = TI_TW + 2 * M_, NT = 2 * B - 1;
    __shared__ float sM[5][NT][IW];      // matrices of padded rows yb .. yb+2B-2 at columns x0-m .. x0+TW+m-1 (clamped)
    __shared__ double sV[5][B][IW];      // column sums of the B output rows

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TI_TW, yb = blockIdx.y * B;
    const size_t npx = (size_t)w * h;
    const size_t p = blockIdx.z;
    const float *R0, *R1;
    pair_frames(R, fstep, p, npx, R0, R1);
    const float2 *fin = MODE == 2 ? flow_in + p * npx : nullptr;
    const float2 *coarse = MODE == 1 ? up.coarse + p * (size_t)up.cw * up.ch : nullptr;

    // ---- phase 1: FarnebackUpdateMatrices of every pixel the tile's windows touch, one thread per pixel
    for (int i = tid; i < NT * IW; i += 256) {
        const int t = i / IW, ix = i - t * IW;
        const int x = clampi(x0 - M_ + ix, 0, w - 1), y = clampi(yb + t - M_, 0, h - 1);
        float dx = 0.f, dy = 0.f;
        if (MODE == 2) {
            const float2 f = fin[(size_t)y * w + x];
            dx = f.x; dy = f.y;
        } else if (MODE == 1) {
            // resize(INTER_LINEAR) of the coarse flow, then * 1/pyr_scale: the operations of k_flow_upsample / k_flow_iter's MODE 1
            const int sx = up.xofs[x], sx1 = sx + 1 < up.cw ? sx + 1 : up.cw - 1;
            const float a1 = up.xa[x], a0 = 1.f - a1;
            int sy;
            float b1;
            resize_coord(y, up.yscale, up.ch, sy, b1);
            const int sy1 = sy + 1 < up.ch ? sy + 1 : up.ch - 1;
            const float b0 = 1.f - b1;
            const float2 p00 = coarse[(size_t)sy * up.cw + sx], p01 = coarse[(size_t)sy * up.cw + sx1];
            const float2 p10 = coarse[(size_t)sy1 * up.cw + sx], p11 = coarse[(size_t)sy1 * up.cw + sx1];
            const float r0x = p00.x * a0 + p01.x * a1, r0y = p00.y * a0 + p01.y * a1;
            const float r1x = p10.x * a0 + p11.x * a1, r1y = p10.y * a0 + p11.y * a1;
            dx = (r0x * b0 + r1x * b1) * up.mul;
            dy = (r0y * b0 + r1y * b1) * up.mul;
        }
        float m[5];
        update_matrices_px(R0, R1, npx, w, h, x, y, dx, dy, m);
#pragma unroll
        for (int c = 0; c < 5; c++) sM[c][t][ix] = m[c];
    }
    __syncthreads();

    // ---- phase 2: column sums, one thread per (channel, column); order of k_blur_solve / OFO_BOX_BLOCKED
    for (int i = tid; i < 5 * IW; i += 256) {
        const int c = i / IW, ix = i - c * IW;
        auto v = [&](int t) { return (double)sM[c][t][ix]; };
        double P = v(0);
#pragma unroll
        for (int j = 1; j < B; j++) P = P + v(j);
        sV[c][0][ix] = P;
        double S = P, Pn = 0;
#pragma unroll
        for (int j = 1; j < B; j++) {
            S = S - v(j - 1);
            const double vn = v(B + j - 1);
            Pn = j == 1 ? vn : Pn + vn;
            sV[c][j][ix] = S + Pn;
        }
    }
    __syncthreads();

    // ---- phase 3: row sums in chunks of three, 1/winsize^2, regularised 2x2 solve
    for (int i = tid; i < TI_TW * B; i += 256) {
        const int ly = i / TI_TW, ox = i - ly * TI_TW;
        const int gx = x0 + ox, gy = yb + ly;
        if (gx >= w || gy >= h) continue;
        double g[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double *rowv = &sV[c][ly][ox];
            double s = 0;
#pragma unroll
            for (int i0 = 0; i0 < B; i0 += 3) {
                double ch = rowv[i0];
                if (i0 + 1 < B) ch += rowv[i0 + 1];
                if (i0 + 2 < B) ch += rowv[i0 + 2];
                s = i0 == 0 ? ch : s + ch;
            }
            g[c] = s * scale;
        }
        const double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
        float2 o;
        o.x = (float)((g[0] * g[4] - g[1] * g[3]) * idet);
        o.y = (float)((g[2] * g[3] - g[1] * g[4]) * idet);
        flow_out[p * npx + (size_t)gy * w + gx] = o;
    }
}

// Instantiated for the window half-widths whose two LDS arrays fit 64 KB: m = 3 .. 7 (winsize 6 .. 15).
bool flow_iter_tile_supported(int winsize)
{
    const int m = winsize / 2;
    return m >= 3 && m <= 7;
}

// The marching kernel wins as soon as its blocks fill the chip; below that a launch is latency bound and the tile kernel wins.
// `marching_blocks`: the grid the marching kernel would be launched with.  tile_mode 0 / 1 forces the choice (the context's
// "tile" option: OFARN_TILE when the context is created, or ofarn_set_option; tests, A/B runs), -1 = by grid size.
bool flow_iter_tile_preferred(long marching_blocks, int tile_mode)
{
    if (tile_mode >= 0) return tile_mode != 0;
    return marching_blocks <= (long)march_cu_count();
}

template <int M_>
static void launch_flow_iter_tile_m(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w, int h,
                                    int npairs, int winsize, int mode, const float *coarse, int cw, int ch, const int *d_xofs,
                                    const float *d_xa, float mul)
{
    constexpr int B = 2 * M_ + 1;
    dim3 grid((unsigned)((w + TI_TW - 1) / TI_TW), (unsigned)((h + B - 1) / B), npairs);
    const double scale = 1. / ((double)winsize * winsize);
    const double yscale = ch > 0 ? 1. / ((double)h / ch) : 1.;
    UpsampleArgs up{reinterpret_cast<const float2 *>(coarse), cw, ch, d_xofs, d_xa, yscale, mul, nullptr};
    const float2 *fin = reinterpret_cast<const float2 *>(flow_in);
    float2 *fout = reinterpret_cast<float2 *>(flow_out);
    if (mode == 0)
        hipLaunchKernelGGL((k_flow_iter_tile<M_, 0>), grid, dim3(256), 0, s, R, fstep, fin, fout, w, h, scale, up);
    else if (mode == 1)
        hipLaunchKernelGGL((k_flow_iter_tile<M_, 1>), grid, dim3(256), 0, s, R, fstep, fin, fout, w, h, scale, up);
    else
        hipLaunchKernelGGL((k_flow_iter_tile<M_, 2>), grid, dim3(256), 0, s, R, fstep, fin, fout, w, h, scale, up);
}

void launch_flow_iter_tile(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w, int h,
                           int npairs, int winsize, int mode, const float *coarse, int cw, int ch, const int *d_xofs,
                           const float *d_xa, float mul)
{
    switch (winsize / 2) {
    case 3: launch_flow_iter_tile_m<3>(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, d_xa, mul); break;
    case 4: launch_flow_iter_tile_m<4>(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, d_xa, mul); break;
    case 5: launch_flow_iter_tile_m<5>(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, d_xa, mul); break;
    case 6: launch_flow_iter_tile_m<6>(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, d_xa, mul); break;
    case 7: launch_flow_iter_tile_m<7>(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, d_xa, mul); break;
    default: break;
    }
}

}  // namespace ofarn
