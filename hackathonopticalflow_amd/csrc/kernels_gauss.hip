// kernels_gauss.hip -- fused Farneback iteration with the Gaussian window (flags & OPTFLOW_FARNEBACK_GAUSSIAN):
// FarnebackUpdateMatrices + FarnebackUpdateFlow_GaussianBlur of one iteration in one kernel, M never in HBM.
//
// Same marching structure as k_flow_iter (kernels_fast.hip): a block of 256 threads owns march_out_width(m) output columns plus an
// m-column halo per side, one thread per column, marching down a strip; the gather of the next matrix row is in flight
// while the current row is processed.  What differs is the window arithmetic (optflowgf.cpp, float32 throughout):
//   column pass  s0 = M[y]*k[0];   s0 = s0 + (M[y+i] + M[y-i]) * k[i]       i = 1..m   (rows clamped: replicate border)
//   row pass     sum = v[x]*k[0];  sum = sum + k[i] * (v[x-i] + v[x+i])     i = 1..m   (columns clamped)
// A weighted window has no running-sum form: every output row needs all 2m+1 matrix rows of its column.  They live in
// REGISTERS, 5 channels x (2m + K) rows per thread, as a window that slides by K rows at a time: the marching loop is
// unrolled K times, step u of a group writes slot 2m+u and reads slots u .. u+2m -- all fixed registers -- and after the
// K steps the last 2m rows move down by K slots (5 * 2m / K moves per row; a window sliding every row would cost 5 * 2m,
// dynamic register indexing a v_movrel per access, a phase-dispatched ring 2m+1 copies of the row body).
// The column sums of a row are exchanged through a double-buffered LDS line behind an LDS-only barrier,
// the row pass + the regularised 2x2 solve of the PREVIOUS row sit in the same straight-line code (skewed pipeline).
// Results equal the unfused k_update_matrices + k_gauss_solve and the CPU oracle bit for bit.
#include "flow_iter_common.h"
#include "ofarn_internal.h"

#include <type_traits>

namespace ofarn {

template <int M_> struct GaussTaps { float k[M_ + 1]; };

#ifndef OFARN_GAUSS_WAVES
#define OFARN_GAUSS_WAVES 2
#endif

// K = rows per slide of the register window (see the header).
template <int M_> struct GaussGroup { static constexpr int value = M_ <= 7 ? 5 : 3; };

template <int M_, int MODE>   // MODE 0: flow_in == 0;  1: flow_in = upsample(coarse)*mul;  2: flow_in from HBM
__global__ __launch_bounds__(FI_THREADS, OFARN_GAUSS_WAVES) void k_flow_iter_gauss(const float *__restrict__ R, int fstep,
                                                                                   const float2 *__restrict__ flow_in,
                                                                                   float2 *__restrict__ flow_out, int w, int h,
                                                                                   int strip_h, GaussTaps<M_> taps, UpsampleArgs up)
{
    constexpr int TAPS = 2 * M_ + 1;
    constexpr int OUTW = march_out_width(M_);
    __shared__ float sV[2][5][FI_THREADS];

    const int tid = threadIdx.x;
    unsigned bidx, bidy, bidz;
    xcd_remap(bidx, bidy, bidz);
    const int x = (int)bidx * OUTW - M_ + tid;
    const int xc = clampi(x, 0, w - 1);
    const float bx = border_x(xc, w);
    const bool ax = border_applies(xc, w);
    const int y0 = (int)bidy * strip_h;
    const int y1 = min(y0 + strip_h, h);
    const size_t npx = (size_t)w * h;
    const size_t p = bidz;
    const float *R0, *R1;
    pair_frames(R, fstep, p, npx, R0, R1);
    const float2 *fin = MODE == 2 ? flow_in + p * npx : nullptr;
    float2 *fout = flow_out + p * npx;

    int usx = 0, usx1 = 0;
    float ua1 = 0.f;
    const float2 *coarse = nullptr;
    if (MODE == 1) {
        usx = up.xofs[xc];
        usx1 = usx + 1 < up.cw ? usx + 1 : up.cw - 1;
        ua1 = up.xa[xc];
        coarse = up.coarse + p * (size_t)up.cw * up.ch;
    }
    struct FlowRaw { float2 p00, p01, p10, p11; float b1; };
    auto flow_issue = [&](int yy, FlowRaw &fr) {
        if (MODE == 2) fr.p00 = ldg_f2(fin, ((unsigned)yy * (unsigned)w + (unsigned)xc) * 8u);
        else if (MODE == 1) {
            int sy;
            float b1;
            resize_coord(yy, up.yscale, up.ch, sy, b1);
            sy = __builtin_amdgcn_readfirstlane(sy);
            fr.b1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b1)));
            const int sy1 = sy + 1 < up.ch ? sy + 1 : up.ch - 1;
            const unsigned r0 = (unsigned)sy * (unsigned)up.cw, r1 = (unsigned)sy1 * (unsigned)up.cw;
            fr.p00 = ldg_f2(coarse, (r0 + (unsigned)usx) * 8u); fr.p01 = ldg_f2(coarse, (r0 + (unsigned)usx1) * 8u);
            fr.p10 = ldg_f2(coarse, (r1 + (unsigned)usx) * 8u); fr.p11 = ldg_f2(coarse, (r1 + (unsigned)usx1) * 8u);
        }
    };
    auto flow_finish = [&](const FlowRaw &fr, float &dx, float &dy) {
        dx = 0.f; dy = 0.f;
        if (MODE == 2) { dx = fr.p00.x; dy = fr.p00.y; }
        else if (MODE == 1) {
            const float ua0 = 1.f - ua1;
            const float b1 = fr.b1, b0 = 1.f - b1;
            const float r0x = fr.p00.x * ua0 + fr.p01.x * ua1, r0y = fr.p00.y * ua0 + fr.p01.y * ua1;
            const float r1x = fr.p10.x * ua0 + fr.p11.x * ua1, r1y = fr.p10.y * ua0 + fr.p11.y * ua1;
            dx = (r0x * b0 + r1x * b1) * up.mul;
            dy = (r0y * b0 + r1y * b1) * up.mul;
        }
    };
    auto row_of = [&](int r) { return clampi(r, 0, h - 1); };

    // win[c][i]: matrix channel c of the window's i-th padded row; every index below is a compile-time constant, so the
    // array is 5 * (TAPS - 1 + K) plain registers.
    constexpr int K = GaussGroup<M_>::value;
    float win[5][TAPS - 1 + K];
#pragma unroll
    for (int c = 0; c < 5; c++)
#pragma unroll
        for (int i = 0; i < TAPS - 1 + K; i++) win[c][i] = 0.f;

    GatherRaw raw;
    FlowRaw fr{};
    {
        float dx, dy;
        flow_issue(row_of(y0 - M_), fr);
        flow_finish(fr, dx, dy);
        __builtin_amdgcn_sched_barrier(0);
        flow_issue(row_of(y0 - M_ + 1), fr);
        __builtin_amdgcn_sched_barrier(0);
        gather_issue(R0, R1, npx, w, h, xc, row_of(y0 - M_), dx, dy, raw);
        __builtin_amdgcn_sched_barrier(0);
    }

    const int nsteps = (y1 - y0) + TAPS - 1;   // padded rows y0 .. y1+2m-1; step s emits output row y0 + s - 2m
    const int tc = clampi(tid, M_, FI_THREADS - M_ - 1);   // halo threads redo a neighbour's row pass (no branch)
    const bool writer = tid >= M_ && tid < M_ + OUTW && x < w;

    // row pass + solve of the row whose column sums are in sV[b]
    auto hsum_row = [&](const int b, const int y) {
        float g[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float *v = &sV[b][c][tc];
            float sum = v[0] * taps.k[0];
#pragma unroll
            for (int i = 1; i <= M_; i++) sum = sum + taps.k[i] * (v[-i] + v[i]);
            g[c] = sum;
        }
        const double g11 = g[0], g12 = g[1], g22 = g[2], h1 = g[3], h2 = g[4];
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        float2 o;
        o.x = (float)((g11 * h2 - g12 * h1) * idet);
        o.y = (float)((g22 * h1 - g12 * h2) * idet);
        if (writer) stg_f2(fout, ((unsigned)y * (unsigned)w + (unsigned)x) * 8u, o);
    };

    int buf = 0;
    bool have_prev = false;
    int yprev = 0;
    // step u of a group of K: the new row goes to slot TAPS-1+u, the column pass reads slots u .. u+TAPS-1 (centre u+M_)
    auto do_step = [&](auto uc, const int step) {
        constexpr int U = decltype(uc)::value;
        const int t = y0 + step;
        float m[5];
        matrices_finish(raw, bx, ax, h, row_of(t - M_), m);
        {
            float dx, dy;
            flow_finish(fr, dx, dy);
            __builtin_amdgcn_sched_barrier(0);
            flow_issue(row_of(t + 2 - M_), fr);
            __builtin_amdgcn_sched_barrier(0);
            gather_issue(R0, R1, npx, w, h, xc, row_of(t + 1 - M_), dx, dy, raw);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool emit = step >= TAPS - 1;            // uniform
        if (have_prev) hsum_row(buf ^ 1, yprev);
#pragma unroll
        for (int c = 0; c < 5; c++) win[c][TAPS - 1 + U] = m[c];
        if (emit) {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                float s0 = win[c][U + M_] * taps.k[0];
#pragma unroll
                for (int i = 1; i <= M_; i++) s0 = s0 + (win[c][U + M_ + i] + win[c][U + M_ - i]) * taps.k[i];
                sV[buf][c][tid] = s0;
            }
            barrier_lds_only();
            have_prev = true;
            yprev = t - (TAPS - 1);
            buf ^= 1;
        }
    };
    auto group = [&](auto self, auto uc, const int step) -> void {
        constexpr int U = decltype(uc)::value;
        if constexpr (U < K) {
            if (step + U < nsteps) do_step(uc, step + U);     // uniform; only the last group of a strip is partial
            self(self, std::integral_constant<int, U + 1>{}, step);
        }
    };
    for (int step = 0; step < nsteps; step += K) {
        group(group, std::integral_constant<int, 0>{}, step);
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int i = 0; i < TAPS - 1; i++) win[c][i] = win[c][i + K];
    }
    if (have_prev) hsum_row(buf ^ 1, yprev);
}

// Instantiated for m = winsize/2 = 3..8 (winsize 6..17): the window is 5 * (2m + K) registers (three waves per SIMD up to
// m = 7, two for m = 8 and for the upsampling first iteration of m = 7); wider Gaussian windows take the unfused kernels.
bool flow_iter_gauss_supported(int winsize)
{
    const int m = winsize / 2;
    return m >= 3 && m <= 8;
}

// Resident blocks per CU of one instantiation on the current device (it decides the strip height): 3 where the window
// fits 168 registers, else 2.  Asked of the runtime once per (instantiation, device).
template <int M_, int MODE>
static int gauss_blocks_per_cu()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int nb = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (!nb) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k_flow_iter_gauss<M_, MODE>), FI_THREADS, 0) != hipSuccess ||
            nb < 1)
            nb = OFARN_GAUSS_WAVES;
        __atomic_store_n(&cached[dev], nb, __ATOMIC_RELAXED);
    }
    return nb;
}

template <int M_, int MODE>
static void launch_flow_iter_gauss_mm(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w,
                                      int h, int npairs, const float *h_kern, const float *coarse, int cw, int ch,
                                      const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul)
{
    constexpr int OUTW = march_out_width(M_);
    constexpr int B = 2 * M_ + 1;
    GaussTaps<M_> taps;
    for (int i = 0; i <= M_; i++) taps.k[i] = h_kern[i];
    const int strip_h = best_strip_units(h, 1, B - 1, (int)((w + OUTW - 1) / OUTW) * npairs, gauss_blocks_per_cu<M_, MODE>());
    dim3 grid((unsigned)((w + OUTW - 1) / OUTW), (unsigned)((h + strip_h - 1) / strip_h), npairs);
    const double yscale = ch > 0 ? 1. / ((double)h / ch) : 1.;
    (void)d_yofs; (void)d_ya;     // the row table is recomputed in the kernel (resize_coord)
    UpsampleArgs up{reinterpret_cast<const float2 *>(coarse), cw, ch, d_xofs, d_xa, yscale, mul, nullptr};
    hipLaunchKernelGGL((k_flow_iter_gauss<M_, MODE>), grid, dim3(FI_THREADS), 0, s, R, fstep, reinterpret_cast<const float2 *>(flow_in),
                       reinterpret_cast<float2 *>(flow_out), w, h, strip_h, taps, up);
}

template <int M_>
static void launch_flow_iter_gauss_m(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w,
                                     int h, int npairs, const float *h_kern, int mode, const float *coarse, int cw, int ch,
                                     const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul)
{
    if (mode == 0)
        launch_flow_iter_gauss_mm<M_, 0>(s, R, fstep, flow_in, flow_out, w, h, npairs, h_kern, coarse, cw, ch, d_xofs, d_xa, d_yofs, d_ya, mul);
    else if (mode == 1)
        launch_flow_iter_gauss_mm<M_, 1>(s, R, fstep, flow_in, flow_out, w, h, npairs, h_kern, coarse, cw, ch, d_xofs, d_xa, d_yofs, d_ya, mul);
    else
        launch_flow_iter_gauss_mm<M_, 2>(s, R, fstep, flow_in, flow_out, w, h, npairs, h_kern, coarse, cw, ch, d_xofs, d_xa, d_yofs, d_ya, mul);
}

// h_kern: host pointer to the m+1 taps of FarnebackUpdateFlow_GaussianBlur (kernel[0] first).  Modes as launch_flow_iter.
void launch_flow_iter_gauss(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w, int h,
                            int npairs, int winsize, const float *h_kern, int mode, const float *coarse, int cw, int ch,
                            const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul)
{
#define OFARN_FG_CASE(M)                                                                                              \
    case M:                                                                                                           \
        launch_flow_iter_gauss_m<M>(s, R, fstep, flow_in, flow_out, w, h, npairs, h_kern, mode, coarse, cw, ch, d_xofs, \
                                    d_xa, d_yofs, d_ya, mul);                                                          \
        break;
    switch (winsize / 2) {
        OFARN_FG_CASE(3) OFARN_FG_CASE(4) OFARN_FG_CASE(5) OFARN_FG_CASE(6) OFARN_FG_CASE(7) OFARN_FG_CASE(8)
        default: break;
    }
#undef OFARN_FG_CASE
}

}  // namespace ofarn
