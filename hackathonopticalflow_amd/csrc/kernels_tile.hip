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
#include "farneback_device.h"
#include "flow_iter_common.h"
#include "ofarn_internal.h"

#include <cstdlib>

namespace ofarn {

// Output columns per tile.  30, not 32: with m = 7 the two LDS arrays are then 51.9 KB instead of 54.3 KB, so THREE blocks fit a CU's 160 KB
// instead of two (a 960 x 540 level of one pair: 68 instead of 79 us for its three iterations; 24 columns: 70 us).
#ifndef OFARN_TILE_TW
#define OFARN_TILE_TW 30
#endif
constexpr int TI_TW = OFARN_TILE_TW;
#ifndef OFARN_TILE_PB
#define OFARN_TILE_PB 3     // pixels whose loads a thread of phase 1 keeps in flight together (measured: 3 and 6 alike, 1 slower)
#endif

// One tile: output pixels [x0, x0 + TW) x [yb, yb + B) of pair p.  sM: matrices of padded rows yb .. yb+2B-2 at columns x0-m ..
// x0+TW+m-1 (clamped); sV: column sums of the B output rows.  (A persistent kernel that called this body tile after tile for every
// iteration of the coarse levels behind a device-wide barrier was built in round 4 and measured 40-50 us slower per 1080p turn than
// the separate launches: commit cfc2e1e (tag r04-coop-levels-experiment), profiles/r04_coop_ab.txt.)
template <int M_, int MODE>   // MODE 0: flow_in == 0;  1: flow_in = upsample(coarse)*mul;  2: flow_in from HBM
__device__ __forceinline__ void flow_iter_tile_body(float (&sM)[5][4 * M_ + 1][TI_TW + 2 * M_], double (&sV)[5][2 * M_ + 1][TI_TW + 2 * M_],
                                                    const float *__restrict__ R, int fstep, const float2 *__restrict__ flow_in,
                                                    float2 *__restrict__ flow_out, int w, int h, double scale, const UpsampleArgs &up,
                                                    int x0, int yb, size_t p)
{
    constexpr int B = 2 * M_ + 1, IW = TI_TW + 2 * M_, NT = 2 * B - 1;
    const int tid = threadIdx.x;
    const size_t npx = (size_t)w * h;
    const float *R0, *R1;
    pair_frames(R, fstep, p, npx, R0, R1);
    const float2 *fin = MODE == 2 ? flow_in + p * npx : nullptr;
    const float2 *coarse = MODE == 1 ? up.coarse + p * (size_t)up.cw * up.ch : nullptr;

    // ---- phase 1: FarnebackUpdateMatrices of every pixel the tile's windows touch, one thread per pixel, PB pixels per thread at a
    // time: their flow loads are issued together, then their gathers (R0 + the four taps of R1), then the arithmetic -- two memory round
    // trips per batch instead of two per pixel (a tile has 5.2 pixels per thread; the launch is latency bound).  gather_issue +
    // matrices_finish are the marching kernel's split of update_matrices_px: the same operations in the same order.
    constexpr int PB = OFARN_TILE_PB;
    for (int base = tid; base < NT * IW; base += 256 * PB) {
        int px[PB], py[PB], pt[PB], pix[PB];
        bool live[PB];
#pragma unroll
        for (int q = 0; q < PB; q++) {
            const int i = base + q * 256;
            live[q] = i < NT * IW;
            const int ii = live[q] ? i : 0;
            pt[q] = ii / IW; pix[q] = ii - pt[q] * IW;
            px[q] = clampi(x0 - M_ + pix[q], 0, w - 1); py[q] = clampi(yb + pt[q] - M_, 0, h - 1);
        }
        float dx[PB], dy[PB];
        if (MODE == 2) {
            float2 f[PB];
#pragma unroll
            for (int q = 0; q < PB; q++) f[q] = fin[(size_t)py[q] * w + px[q]];
#pragma unroll
            for (int q = 0; q < PB; q++) { dx[q] = f[q].x; dy[q] = f[q].y; }
        } else if (MODE == 1) {
            // resize(INTER_LINEAR) of the coarse flow, then * 1/pyr_scale: the operations of k_flow_upsample / k_flow_iter's MODE 1
            float2 p00[PB], p01[PB], p10[PB], p11[PB];
            float a1[PB], b1[PB];
#pragma unroll
            for (int q = 0; q < PB; q++) {
                const int sx = up.xofs[px[q]], sx1 = sx + 1 < up.cw ? sx + 1 : up.cw - 1;
                a1[q] = up.xa[px[q]];
                int sy;
                resize_coord(py[q], up.yscale, up.ch, sy, b1[q]);
                const int sy1 = sy + 1 < up.ch ? sy + 1 : up.ch - 1;
                p00[q] = coarse[(size_t)sy * up.cw + sx]; p01[q] = coarse[(size_t)sy * up.cw + sx1];
                p10[q] = coarse[(size_t)sy1 * up.cw + sx]; p11[q] = coarse[(size_t)sy1 * up.cw + sx1];
            }
#pragma unroll
            for (int q = 0; q < PB; q++) {
                const float a0 = 1.f - a1[q], b0 = 1.f - b1[q];
                const float r0x = p00[q].x * a0 + p01[q].x * a1[q], r0y = p00[q].y * a0 + p01[q].y * a1[q];
                const float r1x = p10[q].x * a0 + p11[q].x * a1[q], r1y = p10[q].y * a0 + p11[q].y * a1[q];
                dx[q] = (r0x * b0 + r1x * b1[q]) * up.mul;
                dy[q] = (r0y * b0 + r1y * b1[q]) * up.mul;
            }
        } else {
#pragma unroll
            for (int q = 0; q < PB; q++) { dx[q] = 0.f; dy[q] = 0.f; }
        }
        GatherRaw g[PB];
#pragma unroll
        for (int q = 0; q < PB; q++) gather_issue(R0, R1, npx, w, h, px[q], py[q], dx[q], dy[q], g[q]);
#pragma unroll
        for (int q = 0; q < PB; q++) {
            float m[5];
            matrices_finish(g[q], border_x(px[q], w), border_applies(px[q], w), h, py[q], m);
            if (live[q]) {
#pragma unroll
                for (int c = 0; c < 5; c++) sM[c][pt[q]][pix[q]] = m[c];
            }
        }
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

template <int M_, int MODE>
__global__ __launch_bounds__(256) void k_flow_iter_tile(const float *__restrict__ R, int fstep, const float2 *__restrict__ flow_in,
                                                        float2 *__restrict__ flow_out, int w, int h, double scale, UpsampleArgs up)
{
    constexpr int B = 2 * M_ + 1, IW = TI_TW + 2 * M_, NT = 2 * B - 1;
    __shared__ float sM[5][NT][IW];
    __shared__ double sV[5][B][IW];
    flow_iter_tile_body<M_, MODE>(sM, sV, R, fstep, flow_in, flow_out, w, h, scale, up, blockIdx.x * TI_TW, blockIdx.y * B, blockIdx.z);
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
