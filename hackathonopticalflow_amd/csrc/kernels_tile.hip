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
// x0+TW+m-1 (clamped); sV: column sums of the B output rows.  Called by the one-tile-per-block kernel below and, tile after tile, by
// the persistent kernel that runs all iterations of the coarse levels in one launch (k_flow_levels_coop).
// Flow accesses of the tile body.  COH = false: plain loads and stores (one launch per iteration: the kernel boundary orders them).
// COH = true (k_flow_levels_coop): relaxed ATOMIC 8-byte loads and stores at agent scope -- the memory model's coherent accesses
// (sc1: they neither hit a stale line of a CU's L1 / an XCD's L2 nor leave a dirty one behind), so that flow written by one block
// before a device-wide barrier is what every other block reads behind it WITHOUT any cache write-back or invalidate at the barrier
// (fences there cost 25 us per barrier issued by one thread per block and 130 us issued by all: measured, profiles/r04_coop_ab.txt).
template <bool COH>
__device__ __forceinline__ float2 ld_flow(const float2 *p)
{
    if (!COH) return *p;
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float2 r;
    r.x = __uint_as_float((unsigned)(v & 0xffffffffull));
    r.y = __uint_as_float((unsigned)(v >> 32));
    return r;
}
template <bool COH>
__device__ __forceinline__ void st_flow(float2 *p, float2 v)
{
    if (!COH) { *p = v; return; }
    const unsigned long long u = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int M_, int MODE, bool COH = false>   // MODE 0: flow_in == 0;  1: flow_in = upsample(coarse)*mul;  2: flow_in from HBM
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
            for (int q = 0; q < PB; q++) f[q] = ld_flow<COH>(&fin[(size_t)py[q] * w + px[q]]);
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
                p00[q] = ld_flow<COH>(&coarse[(size_t)sy * up.cw + sx]); p01[q] = ld_flow<COH>(&coarse[(size_t)sy * up.cw + sx1]);
                p10[q] = ld_flow<COH>(&coarse[(size_t)sy1 * up.cw + sx]); p11[q] = ld_flow<COH>(&coarse[(size_t)sy1 * up.cw + sx1]);
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
        st_flow<COH>(&flow_out[p * npx + (size_t)gy * w + gx], o);
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

// ---------------------------------------------------------------------------------------------
// k_flow_levels_coop: EVERY iteration of several consecutive pyramid levels of ONE pair in one launch (round 4, VERDICT r3 next #4a).
// A 1080p pair's levels 2-5 are 288, 72, 20 and 6 tiles; as twelve separate launches they cost 7-11 us each on a mostly idle chip,
// most of it launch-to-launch latency, plus 12-16 us wherever a level's first iteration waits on an event.  Here a persistent grid
// (all blocks co-resident: the host sizes it from the occupancy query) walks the same (level, iteration) sequence, a device-wide
// barrier between steps: arrive = one atomic add per block on a counter in HBM, wait = poll until the counter reaches this step's
// target (the launch's base + step * blocks; the counter only grows, so no reset between launches); the flow that crosses a barrier
// is written and read with coherent (agent-scope atomic) accesses, so the barrier itself does no cache maintenance.  Same tile body, same buffers
// and ping-pong order as the separate launches: bit-identical.  The wait is BOUNDED: a block that has polled for longer than
// `timeout_ticks` of the 100 MHz wall clock raises *fail and leaves, every other block leaves at its next poll -- results are then
// undefined, the host sees the flag after its synchronisation, reruns the levels with separate launches and stops using this kernel
// on the context (a grid that is not fully resident -- a shared GPU -- cannot deadlock the device).
// ---------------------------------------------------------------------------------------------
// The barrier, split in its two halves.  arrive: one relaxed atomic add per ACTIVE block (a block that had a tile in the step) on a
// counter in HBM, once the block's (coherent) flow stores have completed.  wait: thread 0 polls the counter until it reaches the
// cumulative number of arrivals of all steps so far; a block with no tile in the next step does not wait for it at all (the counter
// only grows, so waiting later for a later target covers it) -- a 6-tile level costs 6 arrivals, not one per block of the grid.
// bar[0] = arrivals, bar[1] = the give-up flag the blocks poll now and then (device memory: polling the host-visible copy over PCIe
// from every block cost 115 us per barrier); *fail_host is only WRITTEN, by the block that gives up.
__device__ __forceinline__ void coop_arrive(unsigned long long *bar)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's stores have completed (s_waitcnt vmcnt(0))
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool coop_wait(unsigned long long *bar, unsigned long long target, unsigned *fail_host, unsigned long long timeout_ticks)
{
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned long long t0 = 0;
        for (unsigned n = 0; __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; n++) {
            if ((n & 15u) != 15u) continue;
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { ok = 0; break; }
            const unsigned long long t = wall_clock64();
            if (t0 == 0) t0 = t;
            else if (t - t0 > timeout_ticks) {
                __hip_atomic_store(bar + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(fail_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <int M_>
__global__ __launch_bounds__(256) void k_flow_levels_coop(CoopArgs a)
{
    constexpr int B = 2 * M_ + 1, IW = TI_TW + 2 * M_, NT = 2 * B - 1;
    __shared__ float sM[5][NT][IW];
    __shared__ double sV[5][B][IW];
    const float *prev = a.prev;            // final flow of the level above (nullptr: the first level here is the coarsest)
    int pw = a.pw, ph = a.ph;
    unsigned long long arrivals = a.base;  // what the counter reads once every step so far has been finished by all its blocks
    bool behind = false;                   // this block has not yet waited for the steps before the current one
    for (int li = 0; li < a.nlev; li++) {
        const CoopLevel &L = a.lv[li];
        const float *cur = (li == 0 && !prev) ? a.init : nullptr;
        const int tx = (L.w + TI_TW - 1) / TI_TW, ty = (L.h + B - 1) / B, ntiles = tx * ty;
        const bool active = (int)blockIdx.x < ntiles;
        const unsigned long long act = (unsigned long long)min(ntiles, (int)gridDim.x);
        for (int it = 0; it < a.iterations; it++) {
            // the buffer walk of run_wave: the coarse flow is only read by iteration 0
            const float *busy = (it == 0 && prev) ? prev : cur;
            float *out = (it == a.iterations - 1 && li == a.nlev - 1 && a.final_out) ? a.final_out : (busy == a.flowA ? a.flowB : a.flowA);
            const int mode = it == 0 ? (prev ? 1 : (cur ? 2 : 0)) : 2;
            if (active) {
                if (behind && !coop_wait(a.bar, arrivals, a.fail, a.timeout_ticks)) return;
                behind = false;
                UpsampleArgs up{reinterpret_cast<const float2 *>(prev), pw, ph, L.xofs, L.xa, ph > 0 ? 1. / ((double)L.h / ph) : 1., a.mul, nullptr};
                const float2 *fin = reinterpret_cast<const float2 *>(cur);
                float2 *fout = reinterpret_cast<float2 *>(out);
                for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
                    const int by = t / tx, bx = t - by * tx;
                    if (mode == 0) flow_iter_tile_body<M_, 0, true>(sM, sV, L.R, L.fstep, fin, fout, L.w, L.h, a.scale, up, bx * TI_TW, by * B, 0);
                    else if (mode == 1) flow_iter_tile_body<M_, 1, true>(sM, sV, L.R, L.fstep, fin, fout, L.w, L.h, a.scale, up, bx * TI_TW, by * B, 0);
                    else flow_iter_tile_body<M_, 2, true>(sM, sV, L.R, L.fstep, fin, fout, L.w, L.h, a.scale, up, bx * TI_TW, by * B, 0);
                    __syncthreads();          // sM / sV are reused by the block's next tile
                }
                if (!(li == a.nlev - 1 && it == a.iterations - 1)) coop_arrive(a.bar);      // nobody waits behind the last step
            }
            cur = out;
            arrivals += act;
            behind = true;
        }
        prev = cur; pw = L.w; ph = L.h;
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

// Blocks of k_flow_levels_coop<m> that are resident at once on the current device (0: not instantiated for this window).
int flow_levels_coop_capacity(int winsize)
{
    int per_cu = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (winsize / 2) {
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_levels_coop<3>, 256, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_levels_coop<4>, 256, 0); break;
    case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_levels_coop<5>, 256, 0); break;
    case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_levels_coop<6>, 256, 0); break;
    case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_flow_levels_coop<7>, 256, 0); break;
    default: return 0;
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu * march_cu_count();
}

int flow_levels_coop_tiles(int w, int h, int winsize)
{
    const int B = 2 * (winsize / 2) + 1;
    return ((w + TI_TW - 1) / TI_TW) * ((h + B - 1) / B);
}

void launch_flow_levels_coop(hipStream_t s, const CoopArgs &a, int winsize, int blocks)
{
    switch (winsize / 2) {
    case 3: hipLaunchKernelGGL(k_flow_levels_coop<3>, dim3(blocks), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(k_flow_levels_coop<4>, dim3(blocks), dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL(k_flow_levels_coop<5>, dim3(blocks), dim3(256), 0, s, a); break;
    case 6: hipLaunchKernelGGL(k_flow_levels_coop<6>, dim3(blocks), dim3(256), 0, s, a); break;
    case 7: hipLaunchKernelGGL(k_flow_levels_coop<7>, dim3(blocks), dim3(256), 0, s, a); break;
    default: break;
    }
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
