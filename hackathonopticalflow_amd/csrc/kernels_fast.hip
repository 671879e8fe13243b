// kernels_fast.hip -- fused kernels the default parameter set dispatches to.
//
// k_flow_iter: one Farneback iteration = FarnebackUpdateMatrices + FarnebackUpdateFlow_Blur fused
// (stages C + D; with MODE 1 also stage E, the flow upsample).  M never goes to HBM.
//
//   * A block of 256 threads owns a strip of march_out_width(m) columns (240 for m = winsize/2 = 7) plus an m-column halo
//     on each side, one thread per column, and MARCHES down `strip_h` rows.
//   * Each thread computes G11,G12,G22,h1,h2 for its column at row y+m (bilinear gather of R1,
//     combine with R0, border damping) and keeps the last 2m+1 rows of its column in registers:
//     the box filter's vertical pass is 3 f64 ops per channel and never touches LDS.
//   * The column sums (double) of one row are exchanged through a double-buffered LDS line; the
//     horizontal pass, the 1/winsize^2 scale and the regularised 2x2 solve follow, then the
//     (dx, dy) row is written (coalesced float2).
//   * flow is read from flow_in and written to flow_out (ping-pong): neighbouring strips read
//     each other's halo columns, so the update cannot be in place.
//
// Summation order is the oracle's OFO_BOX_BLOCKED order (column sums: running sums in double
// restarted every 2m+1 rows; then columns x-m..x+m in chunks of three, left to right, in double), so
// results equal the unfused generic kernels and the CPU oracle bit for bit.
//
// Roofline: HBM-bound by design: per pixel and iteration it reads flow 8 B + R0 20 B + R1 20 B
// (gathered, mostly sequential) and writes 8 B, x ((240+2m)/240) x ((strip_h+2m)/strip_h) halo
// re-reads that L2 mostly absorbs; ~150 f64 adds per pixel keep the f64 pipe about half busy.
#include "farneback_device.h"
#include "flow_iter_common.h"
#include "ofarn_internal.h"

#include <cstdio>
#include <type_traits>

namespace ofarn {

// Register FIFO with a UNIFORM runtime index: one 16- or 32-wide vector per channel, which the
// backend keeps in VGPRs and addresses relative to M0 (v_movrels/v_movreld) -- no scratch memory, no
// select chains, and the marching loop can stay rolled.
typedef float ofarn_f16v __attribute__((ext_vector_type(16)));
typedef float ofarn_f32v __attribute__((ext_vector_type(32)));
template <int B> struct FifoVec { typedef ofarn_f32v type; };
template <> struct FifoVec<3> { typedef ofarn_f16v type; };
template <> struct FifoVec<5> { typedef ofarn_f16v type; };
template <> struct FifoVec<7> { typedef ofarn_f16v type; };
template <> struct FifoVec<9> { typedef ofarn_f16v type; };
template <> struct FifoVec<11> { typedef ofarn_f16v type; };
template <> struct FifoVec<13> { typedef ofarn_f16v type; };
template <> struct FifoVec<15> { typedef ofarn_f16v type; };

// OFARN_FI_WAVES: minimum waves per SIMD asked of the register allocator; OFARN_FI_REGCH: channels whose
// 15-row FIFO lives in registers (the rest in LDS).  3 / 3 gives 3 waves per SIMD and 3 blocks per CU.
#ifndef OFARN_FI_WAVES
#define OFARN_FI_WAVES 3
#endif
#ifndef OFARN_FI_REGCH
#define OFARN_FI_REGCH 3
#endif
#if OFARN_FI_WAVES > 0
#define FI_BOUNDS __launch_bounds__(FI_THREADS, OFARN_FI_WAVES)
#else
#define FI_BOUNDS __launch_bounds__(FI_THREADS)
#endif
constexpr int FI_REGCH = OFARN_FI_REGCH;

// OFARN_STAMPS: diagnostic build with in-kernel s_memtime stamps; its code lives in experiments/flow_iter_stamps.inc
#ifdef OFARN_STAMPS
#define OFARN_EXP_PART 1
#include "experiments/flow_iter_stamps.inc"
#else
#define STAMP(k) do { } while (0)
#endif   // channels whose row FIFO lives in registers

// OFARN_HSUM3: horizontal pass in chunks of three columns.  t3[x] = (V[x] + V[x+1]) + V[x+2] is formed in
// registers with two whole-wave DPP shifts (v_mov_b32_dpp wave_shl:1; the two lanes at a wave's end take
// the next wave's first two values from a tiny LDS edge buffer), written to an LDS line, and the window
// sum is ((((t3[x-m] + t3[x-m+3]) + ...) five reads for m = 7 instead of fifteen.
#ifndef OFARN_HSUM3
#define OFARN_HSUM3 1
#endif

// lane i <- src[lane i+1]; lane 63 (no source lane) keeps `edge`
__device__ __forceinline__ double wave_shl1(double edge, double src)
{
    const int lo = __builtin_amdgcn_update_dpp((int)__double2loint(edge), (int)__double2loint(src), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)__double2hiint(edge), (int)__double2hiint(src), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// 32-bit whole-wave shifts: lane i <- src[lane i+1] (lane 63 keeps `edge`) and lane i <- src[lane i-1] (lane 0 keeps `edge`)
__device__ __forceinline__ unsigned wave_shl1_u32(unsigned edge, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)src, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned wave_shr1_u32(unsigned edge, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)src, 0x138, 0xF, 0xF, false);
}

template <int M_, int MODE>   // MODE 0: flow_in == 0;  1: flow_in = upsample(coarse)*mul;  2: flow_in from HBM
__global__ FI_BOUNDS void k_flow_iter(const float *__restrict__ R, int fstep,
                                                          const float2 *__restrict__ flow_in,
                                                          float2 *__restrict__ flow_out, int w, int h,
                                                          int strip_h, double scale, UpsampleArgs up)
{
    constexpr int TAPS = 2 * M_ + 1;
    constexpr int OUTW = march_out_width(M_);
    __shared__ double sV[2][5][FI_THREADS];

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

    // per-thread constants of the on-the-fly upsample (resize INTER_LINEAR, horizontal weights)
    int usx = 0, usx1 = 0;
    float ua1 = 0.f;
    const float2 *coarse = nullptr;
    if (MODE == 1) {
        usx = up.xofs[xc];
        usx1 = usx + 1 < up.cw ? usx + 1 : up.cw - 1;
        ua1 = up.xa[xc];
        coarse = up.coarse + p * (size_t)up.cw * up.ch;
    }

    // ---- software pipeline -------------------------------------------------------------------
    // Row r of the matrices needs the flow at (xc, r), then a flow-dependent gather.  Both are long
    // latency, so they are issued one iteration (gather) and two iterations (flow) ahead of use and
    // land while the f64 box sums of the current row run.
    // b1 (the row's vertical weight) is uniform: it lives in an SGPR, not in a VGPR of the 168-register budget
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

    // Column sums: block-restarted running sums in double (oracle OFO_BOX_BLOCKED).  Padded row t
    // is matrix row clamp(t - M_); blocks of B = TAPS padded rows are aligned at t = 0, and y0 is a
    // multiple of B.  fifo[c][j] holds M_c of the row at offset j of the previous block (the value
    // that leaves the window), P the prefix of the current block, S what is left of the previous one.
    // The loop stays rolled (the unrolled form is ~50 KB of code and thrashes the instruction cache);
    // the register FIFO is a vector per channel indexed by the uniform block offset.
    constexpr int B = TAPS;
    // Channels 0..FI_REGCH-1 keep their FIFO in registers, the rest in LDS: the register budget
    // decides how many waves fit a SIMD (<= 168 VGPRs for three), the LDS budget how many blocks a CU.
    typename FifoVec<B>::type fifo[FI_REGCH];
    __shared__ float sF[5 - FI_REGCH][B][FI_THREADS];
    double P[5], S[5];
#pragma unroll
    for (int c = 0; c < 5; c++) { P[c] = 0; S[c] = 0; }
#pragma unroll
    for (int c = 0; c < FI_REGCH; c++) fifo[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 5 - FI_REGCH; c++)
        for (int q = 0; q < B; q++) sF[c][q][tid] = 0.f;
    // Software pipeline: the gather of matrix row r+1 (and the flow of row r+2 it will depend on)
    // is issued before the box sums of row r, so it lands while they run.
    GatherRaw rawA;
    FlowRaw frA{};
    {
        float dx, dy;
        flow_issue(row_of(y0 - M_), frA);
        flow_finish(frA, dx, dy);
        __builtin_amdgcn_sched_barrier(0);
        flow_issue(row_of(y0 - M_ + 1), frA);
        __builtin_amdgcn_sched_barrier(0);
        gather_issue(R0, R1, npx, w, h, xc, row_of(y0 - M_), dx, dy, rawA);
        __builtin_amdgcn_sched_barrier(0);
    }
    // here: rawA holds the matrix row of padded row t = y0, frA the flow of the row after it

    const int nsteps = (y1 - y0) + B - 1;   // padded rows y0 .. y1+B-2; output y = t-(B-1)
    int buf = 0;

#ifdef OFARN_STAMPS
#define OFARN_EXP_PART 2
#include "experiments/flow_iter_stamps.inc"
#endif
    bool have_prev = false;                  // a row's column sums are in sV[buf ^ 1] awaiting their horizontal pass
    int yprev = 0;
    constexpr bool H3 = OFARN_HSUM3 && (TAPS % 3 == 0);
    __shared__ double sE[H3 ? 2 : 1][5][FI_THREADS / 64 + 1][2];   // first two column sums of every wave (edge buffer)
    double Vp[5] = {0, 0, 0, 0, 0};          // H3: this thread's column sums of the previous row
    const int tc = clampi(tid, M_, FI_THREADS - M_ - 1);   // halo threads redo a neighbour's sums (no branch)
    const bool writer = tid >= M_ && tid < M_ + OUTW && x < w;
    auto hsum_row = [&](const int b, const int y) {
        double g[5];
        double chain = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            // One opaque LDS base register per channel (folded into a single base, the 8-bit ds_read2
            // offsets do not reach the other channels and every read needs its own add).  The base is
            // also chained on the previous channel's sum: left alone the scheduler issues all 75 reads
            // up front (150 VGPRs -> spills); chained, at most one channel's reads are in flight and
            // the other row's independent work fills the latency.
            __attribute__((address_space(3))) const double *v =
                (__attribute__((address_space(3))) const double *)&sV[b][c][tc - M_];
            asm volatile("" : "+v"(v), "+v"(chain));
            double s2 = 0;
#pragma unroll
            for (int i0 = 0; i0 < TAPS; i0 += 3) {   // chunks of three, left to right
                double ch = v[i0];
                if (i0 + 1 < TAPS) ch += v[i0 + 1];
                if (i0 + 2 < TAPS) ch += v[i0 + 2];
                s2 = i0 == 0 ? ch : s2 + ch;
            }
            g[c] = s2 * scale;
            chain = s2;
        }
        const double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
        float2 o;
        o.x = (float)((g[0] * g[4] - g[1] * g[3]) * idet);
        o.y = (float)((g[2] * g[3] - g[1] * g[4]) * idet);
        if (writer) stg_f2(fout, ((unsigned)y * (unsigned)w + (unsigned)x) * 8u, o);
    };
    // H3 stages.  tsum_row: chunk sums of the row whose column sums are in Vp (edges in sE[e]) -> sV[bt].
    auto tsum_row = [&](const int e, const int bt) {
        const int wv = tid >> 6;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double e1 = sE[e][c][wv + 1][0], e2 = sE[e][c][wv + 1][1];   // uniform addresses: broadcast reads
            const double a = Vp[c];
            const double s1 = wave_shl1(e1, a);
            const double s2 = wave_shl1(e2, s1);
            sV[bt][c][tid] = (a + s1) + s2;
        }
    };
    // hsum3_row: window sum of row y out of the chunk sums in sV[b], solve, store.
    auto hsum3_row = [&](const int b, const int y) {
        double g[5];
        double chain = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            __attribute__((address_space(3))) const double *v =
                (__attribute__((address_space(3))) const double *)&sV[b][c][tc - M_];
            asm volatile("" : "+v"(v), "+v"(chain));
            double s2 = v[0];
#pragma unroll
            for (int i = 3; i < TAPS; i += 3) s2 += v[i];
            g[c] = s2 * scale;
            chain = s2;
        }
        const double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
        float2 o;
        o.x = (float)((g[0] * g[4] - g[1] * g[3]) * idet);
        o.y = (float)((g[2] * g[3] - g[1] * g[4]) * idet);
        if (writer) stg_f2(fout, ((unsigned)y * (unsigned)w + (unsigned)x) * 8u, o);
    };
    // KIND 0: first row of a block (j == 0), 1: middle rows, 2: last row (j == B-1) -- compile-time so
    // the P/S updates need no selects.
    auto do_row = [&](auto kind_c, const int step, const int j, GatherRaw &raw, FlowRaw &fr) {
        constexpr int KIND = decltype(kind_c)::value;
        const int t = y0 + step;
        float m[5], old[5];
        STAMP(7);
#ifdef OFARN_STAMPS
#define OFARN_EXP_PART 3
#include "experiments/flow_iter_stamps.inc"
#endif
        STAMP(0);
#ifdef OFARN_EXP_SKELETON   // wrong-result experiment, kept out of this file: experiments/flow_iter_skeleton.inc
#define OFARN_EXP_PART 1
#include "experiments/flow_iter_skeleton.inc"
#endif
        matrices_finish(raw, bx, ax, h, row_of(t - M_), m);
        STAMP(1);
        {
            // vmcnt counts loads in issue order: the flow load goes FIRST in every step so that the
            // wait for it (two steps later) does not also wait for the younger gathers behind it.
            float dx, dy;
            flow_finish(fr, dx, dy);
            __builtin_amdgcn_sched_barrier(0);
            flow_issue(row_of(t + 2 - M_), fr);
            __builtin_amdgcn_sched_barrier(0);
            gather_issue(R0, R1, npx, w, h, xc, row_of(t + 1 - M_), dx, dy, raw);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(2);
        {
            const int ju = __builtin_amdgcn_readfirstlane(j);
#pragma unroll
            for (int c = 0; c < FI_REGCH; c++) { old[c] = fifo[c][ju]; fifo[c][ju] = m[c]; }
#pragma unroll
            for (int c = FI_REGCH; c < 5; c++) { old[c] = sF[c - FI_REGCH][ju][tid]; sF[c - FI_REGCH][ju][tid] = m[c]; }
        }
        const bool emit = step >= B - 1;           // uniform: output row y = t-(B-1) >= y0
        // Skewed pipeline: the horizontal pass + solve of the PREVIOUS output row (its column sums were
        // exchanged one step ago) sits in the same straight-line code as this row's column sums, so
        // the scheduler can interleave the LDS reads and f64 add chains of one with the f32 math,
        // register-FIFO traffic and f64 updates of the other.  One barrier per step.
        if constexpr (H3) {
            // three rows in flight: window sums + solve of row t-B-1 (chunk sums written one step ago), chunk
            // sums of row t-B (column sums in Vp, edges written one step ago), column sums of row t-(B-1)
            const int par = step & 1;
            if (step >= B + 1) hsum3_row(par ^ 1, t - B - 1);
            if (step >= B) tsum_row(par ^ 1, par);
            STAMP(5);
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const double vn = (double)m[c];
                double V;
                if constexpr (KIND == 0) P[c] = vn; else P[c] = P[c] + vn;
                if constexpr (KIND == 2) { V = P[c]; S[c] = P[c]; }
                else { S[c] = S[c] - (double)old[c]; V = S[c] + P[c]; }
                Vp[c] = V;
            }
            if (emit && (tid & 63) < 2) {
#pragma unroll
                for (int c = 0; c < 5; c++) sE[par][c][tid >> 6][tid & 63] = Vp[c];
            }
            STAMP(3);
            if (!emit) return;
            barrier_lds_only();
            STAMP(4);
            return;
        }
        if (have_prev) hsum_row(buf ^ 1, yprev);
        STAMP(5);
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double vn = (double)m[c];
            double V;
            if constexpr (KIND == 0) P[c] = vn; else P[c] = P[c] + vn;
            if constexpr (KIND == 2) { V = P[c]; S[c] = P[c]; }
            else { S[c] = S[c] - (double)old[c]; V = S[c] + P[c]; }
            if (emit) sV[buf][c][tid] = V;
        }
        STAMP(3);
        if (!emit) return;
        barrier_lds_only();
        STAMP(4);
        have_prev = true;
        yprev = t - (B - 1);
        buf ^= 1;
    };

#if defined(OFARN_EXP_SKELETON) && OFARN_EXP_SKELETON == 3
#define OFARN_EXP_PART 3
#include "experiments/flow_iter_skeleton.inc"
#endif
    {
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        int step = 0;
        while (step < nsteps) {
            do_row(K0{}, step, 0, rawA, frA);
            step++;
            for (int jj = 1; jj < B - 1 && step < nsteps; jj++, step++) do_row(K1{}, step, jj, rawA, frA);
            if (step < nsteps) {
                do_row(K2{}, step, B - 1, rawA, frA);
                step++;
            }
        }
#ifdef OFARN_EXP_SKELETON
        if (false)
#endif
        if constexpr (H3) {
            // drain: chunk sums of the last row, then the two outstanding window sums
            const int par = nsteps & 1;
            if (nsteps >= B + 1) hsum3_row(par ^ 1, y0 + nsteps - B - 1);
            if (nsteps >= B) {
                tsum_row(par ^ 1, par);
                barrier_lds_only();
                hsum3_row(par, y0 + nsteps - B);
            }
        } else if (have_prev) hsum_row(buf ^ 1, yprev);     // drain the pipeline: last output row
    }
#ifdef OFARN_STAMPS
#define OFARN_EXP_PART 4
#include "experiments/flow_iter_stamps.inc"
#endif
}

// ---------------------------------------------------------------------------------------------
// k_polyexp_march: FarnebackPolyExp (stage B) with compile-time radius N, marching layout.
//   SRC 0: reads the float level image I[h][w] (levels >= 1).
//   SRC 1: level 0 fused with stage A: reads the uint8 frame and applies the 3-tap Gaussian
//          ([1/4,1/2,1/4], BORDER_REFLECT_101, row pass then column pass) on the fly -- at scale 1
//          resize() is the identity, so the level image never exists in HBM.
// One thread per column (march_out_width(N) = 240 outputs per block row), marching down strip_h rows with the last
// 2N+1 level-image values of its column in registers: the float32 vertical pass is private, the
// (r0, r1, r2) line is exchanged through a double-buffered LDS line, the horizontal pass
// accumulates in double in optflowgf.cpp's order.  Bit-identical to k_polyexp / the oracle.
// ---------------------------------------------------------------------------------------------
#ifdef OFARN_PE_WAVES      // experiment: ask the register allocator for this many waves per SIMD (default: 74 VGPRs = 6 waves)
#define OFARN_PE_OCC __attribute__((amdgpu_waves_per_eu(OFARN_PE_WAVES, OFARN_PE_WAVES)))
#else
#define OFARN_PE_OCC
#define OFARN_PE_WAVES 6
#endif
// Output columns per block.  Round 4: 240 ends a block's row segment on a whole 128-byte line in the float4
// plane (3840 B) but on a HALF line in the scalar plane (960 B = 7.5 lines); 192 ends both on whole lines (3072 + 768 B) and the bare
// store pattern runs at 5.6 instead of 4.65 TB/s (tools/microbench/hbm_rw.hip, profiles/r04_hbm_rw.txt).  OFARN_PE_ALIGNED: the
// writers are lanes 0 .. OUTW-1 -- whole waves, each wave's 64 columns starting on a line in both planes -- and the 2N halo columns
// are handled by the first lanes of the following wave (default: halo lanes first, as in the iteration kernels).
#ifndef OFARN_PE_OUTW
#define OFARN_PE_OUTW 0
#endif
#ifndef OFARN_PE_ALIGNED
#define OFARN_PE_ALIGNED 0
#endif
#ifndef OFARN_PE_THREADS       // block size of the marching polynomial expansion (experiment: 384 threads / 320 columns, 512 / 480)
#define OFARN_PE_THREADS FI_THREADS
#endif
constexpr int pe_out_width(int n) { return OFARN_PE_OUTW ? OFARN_PE_OUTW : march_out_width(n); }
template <int N, int SRC>
__global__ __launch_bounds__(OFARN_PE_THREADS) OFARN_PE_OCC void k_polyexp_march(const void *__restrict__ src, size_t src_stride,
                                                              float *__restrict__ R, int w, int h, int strip_h,
                                                              PolyCoef c, float k0, float k1, float k2, int nt)
{
    constexpr int TAPS = 2 * N + 1;
    constexpr int OUTW = pe_out_width(N);
    static_assert(OUTW + 2 * N <= OFARN_PE_THREADS, "block too narrow for the output width");
    __shared__ float sRow[2][3][OFARN_PE_THREADS];

    const int tid = threadIdx.x;
#if OFARN_PE_ALIGNED
    // column relative to the block's first output column, and the lane's place in the LDS line (= column + N)
    const int col = tid < OUTW ? tid : (tid < OUTW + N ? tid - OUTW - N : tid - N);
    const int li = tid < OUTW ? tid + N : (tid < OUTW + N ? tid - OUTW : tid);
#else
    const int col = tid - N, li = tid;
#endif
    const int x = blockIdx.x * OUTW + col;
    const int xc = clampi(x, 0, w - 1);
    const int y0 = blockIdx.y * strip_h;
    const int y1 = min(y0 + strip_h, h);
    const size_t npx = (size_t)w * h;
    const float *img = SRC == 0 ? reinterpret_cast<const float *>(src) + (size_t)blockIdx.z * src_stride : nullptr;
    const uint8_t *frm = SRC == 1 ? reinterpret_cast<const uint8_t *>(src) + (size_t)blockIdx.z * src_stride : nullptr;
    float *out = R + (size_t)blockIdx.z * r_frame_stride(npx);

    // Source of the level-image value of one row of this column, split in two so that the loads of row y+1 are ISSUED
    // before the stores of row y and their data is first touched in the next turn of the loop: vmcnt counts loads and
    // stores together and retires them in order, so a load issued behind a row's stores cannot be waited for without
    // waiting for those stores too -- every row then pays the full write latency.
    //   SRC 0: the float level image.  SRC 1: level 0, 3-tap blur state (row pass of rows yy-1, yy, yy+1) from the bytes.
    const int xl = reflect101(xc - 1, w), xr = reflect101(xc + 1, w);
    float rpm = 0.f, rpc = 0.f, rpp = 0.f, icur = 0.f, pf = 0.f;
    // SRC 1, marching.  Default: three byte loads per lane and row (f[xl], f[xc], f[xr]).  -DOFARN_PE_ONE_LOAD builds the form VERDICT r2
    // asked for: ONE byte per lane and row, the neighbours taken from the neighbouring lanes with whole-wave DPP shifts, except where a
    // neighbouring lane does not hold the neighbouring pixel -- lanes 0 and 63 of a wave and the lanes at or beyond the frame's left /
    // right edge (clamped columns, reflect-101 neighbours: there xl == xr) fetch one extra byte, f[xe], in a second, almost empty load.
    // Bit-identical, SQ_INSTS_VMEM_RD 3.28 -> 2.20 per pixel -- and 1.5 % SLOWER in a same-box A/B (5.98 / 6.00 / 6.01 against
    // 5.82 / 5.96 / 5.95 ms per 512 frames, alternating runs): the texture addresser's time for the byte loads was never the limit, and
    // the divergent second load + two DPP moves + two selects per row cost a little.  Kept as a build option for the record.
#ifdef OFARN_PE_ONE_LOAD
    const int lane = tid & 63;
    const bool edge_col = x <= 0 || x >= w - 1;              // xl == xr here
    const bool need_e = edge_col || lane == 0 || lane == 63;
    const int xe = (lane == 63 && !edge_col) ? xr : xl;
#endif
    unsigned nc = 0, ne = 0;
#ifndef OFARN_PE_ONE_LOAD
    unsigned ne2 = 0;
#endif
    int ry = -0x40000000;
    auto rowpass3 = [&](uint8_t a, uint8_t b, uint8_t cc) {
        float s = k0 * (float)a;
        s = s + k1 * (float)b;
        s = s + k2 * (float)cc;
        return s;
    };
    auto rowpass = [&](int row) {
        const uint8_t *f = frm + (size_t)row * w;
        return rowpass3(f[xl], f[xc], f[xr]);
    };
    // issue_next(yy): loads for the value finish(yy) will return (yy in [0, h-1], non-decreasing from call to call)
    auto issue_next = [&](int yy) {
        if (SRC == 0) pf = img[(size_t)yy * w + xc];
        else if (yy == ry + 1) {                         // uniform; the other cases need no new row or take the slow path
            const uint8_t *f = frm + (size_t)reflect101(yy + 1, h) * w;
            nc = f[xc];
#ifndef OFARN_PE_ONE_LOAD        // default: three byte loads per lane and row (see below)
            ne = f[xl];
            ne2 = f[xr];
#else
            if (need_e) ne = f[xe];
#endif
        }
    };
    auto finish = [&](int yy) -> float {
        if (SRC == 0) return pf;
        if (yy == ry) return icur;
        if (yy == ry + 1) {
#ifndef OFARN_PE_ONE_LOAD
            const uint8_t nl = (uint8_t)ne, nr = (uint8_t)ne2;
#else
            const unsigned dl = wave_shr1_u32(ne, nc), dr = wave_shl1_u32(ne, nc);   // lane 0 / 63 keep their own extra byte
            const uint8_t nl = (uint8_t)(edge_col ? ne : dl), nr = (uint8_t)(edge_col ? ne : dr);
#endif
            rpm = rpc; rpc = rpp; rpp = rowpass3(nl, (uint8_t)nc, nr);
        }
        else { rpm = rowpass(reflect101(yy - 1, h)); rpc = rowpass(yy); rpp = rowpass(reflect101(yy + 1, h)); }
        ry = yy;
        icur = k1 * rpc + k2 * (rpp + rpm);
        return icur;
    };

    float win[TAPS];
    for (int j = 1; j < TAPS; j++) {
        const int yy = clampi(y0 - N + (j - 1), 0, h - 1);
        issue_next(yy);
        const float v = finish(yy);
#pragma unroll
        for (int q = 1; q < TAPS; q++) if (q == j) win[q] = v;
    }

    int buf = 0;
    const bool writer = col >= 0 && col < OUTW && x < w;
    float pv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};     // the previous row's result, stored one turn late (see below)
    issue_next(clampi(y0 + N, 0, h - 1));
    for (int y = y0; y < y1; y++) {
        const float v = finish(clampi(y + N, 0, h - 1));
        issue_next(clampi(y + 1 + N, 0, h - 1));
        // The stores of row y-1 go out HERE, behind the loads of row y+1 and a whole row of arithmetic before the next
        // wait on a load: the compiler's wait for those loads is a plain vmcnt(0), which also waits for every store issued
        // before it -- stores issued at the end of a row would be waited for at once, with their full write latency exposed.
        if (y > y0 && writer) {
            if (nt) store_r_nt(out, npx, (unsigned)(y - 1) * (unsigned)w + (unsigned)x, pv);
            else store_r(out, npx, (unsigned)(y - 1) * (unsigned)w + (unsigned)x, pv);
        }
#pragma unroll
        for (int j = 0; j < TAPS - 1; j++) win[j] = win[j + 1];
        win[TAPS - 1] = v;
        float r0 = win[N] * c.g[0], r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int k = 1; k <= N; k++) {
            const float a = win[N - k], b = win[N + k];
            const float pp = a + b;
            r0 = r0 + c.g[k] * pp;
            r1 = r1 + c.xg[k] * (b - a);
            r2 = r2 + c.xxg[k] * pp;
        }
        sRow[buf][0][li] = r0;
        sRow[buf][1][li] = r1;
        sRow[buf][2][li] = r2;
        barrier_lds_only();   // __syncthreads() would also drain vmcnt
        if (writer) {
            const float *p0 = &sRow[buf][0][li], *p1 = &sRow[buf][1][li], *p2 = &sRow[buf][2][li];
            const float g0 = c.g[0];
            double b1 = p0[0] * g0, b2 = 0, b3 = p1[0] * g0, b4 = 0, b5 = p2[0] * g0, b6 = 0;
#pragma unroll
            for (int k = 1; k <= N; k++) {
                const float gk = c.g[k], xgk = c.xg[k], xxgk = c.xxg[k];
                const double tg = p0[k] + p0[-k];
                // tg and the taps are float values held in double: their product is exact in double (24 + 24 bits), so the
                // fused multiply-add rounds exactly like OpenCV's separate multiply and add -- one instruction instead of two
                b1 = __builtin_fma(tg, (double)gk, b1);
                b4 = __builtin_fma(tg, (double)xxgk, b4);
                b2 += (p0[k] - p0[-k]) * xgk;
                b3 += (p1[k] + p1[-k]) * gk;
                b6 += (p1[k] - p1[-k]) * xgk;
                b5 += (p2[k] + p2[-k]) * gk;
            }
            pv[0] = (float)(b3 * c.ig11);
            pv[1] = (float)(b2 * c.ig11);
            pv[2] = (float)(b1 * c.ig03 + b5 * c.ig33);
            pv[3] = (float)(b1 * c.ig03 + b4 * c.ig33);
            pv[4] = (float)(b6 * c.ig55);
        }
        buf ^= 1;
    }
    if (y1 > y0 && writer) {
        if (nt) store_r_nt(out, npx, (unsigned)(y1 - 1) * (unsigned)w + (unsigned)x, pv);
        else store_r(out, npx, (unsigned)(y1 - 1) * (unsigned)w + (unsigned)x, pv);
    }
}

// ---------------------------------------------------------------------------------------------
// k_level_hpass_lds: stage A pass 1 for levels >= 1 (same arithmetic as k_level_hpass).
// One block per frame row: the uint8 row is converted to float once and staged in LDS together
// with its reflect-101 border, so the ksize-tap loop has no border logic and no byte loads.
// LDS index i is skewed to i + (i >> 5): level columns are 2^k source pixels apart, which would
// put every lane on one bank; with the skew any power-of-two stride is conflict free.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int skew(int i) { return i + (i >> 5); }

__global__ __launch_bounds__(256) void k_level_hpass_lds(const uint8_t *__restrict__ frames, size_t frame_stride,
                                                          int W, int H, const float *__restrict__ kern, int ksize,
                                                          const int *__restrict__ xofs, int dw,
                                                          float2 *__restrict__ tmp, int symm)
{
    extern __shared__ float srow[];   // skew(W + 2r) floats, then ksize kernel taps
    const int r = ksize >> 1;
    const int y = blockIdx.y;
    const uint8_t *row = frames + (size_t)blockIdx.z * frame_stride + (size_t)y * W;
    const int tid = threadIdx.x;
    const int ext = W + 2 * r;
    float *sk = srow + skew(ext) + 1;
    for (int i = tid; i < ext; i += 256) srow[skew(i)] = (float)row[reflect101(i - r, W)];
    for (int i = tid; i < ksize; i += 256) sk[i] = kern[i];
    __syncthreads();
    float2 *dst = tmp + ((size_t)blockIdx.z * H + y) * dw;
    for (int dx = tid; dx < dw; dx += 256) {
        const int sx = xofs[dx];
        const int sx1 = sx + 1 < W ? sx + 1 : W - 1;
        // source column c sits at extended index c + r; tap t reads c - r + t -> extended c + t
        if (symm && (ksize == 3 || ksize == 5)) {
            float a[5], b[5];
            for (int t = 0; t < ksize; t++) { a[t] = srow[skew(sx + t)]; b[t] = srow[skew(sx1 + t)]; }
            dst[dx] = make_float2(row_small_symm(a, sk, ksize), row_small_symm(b, sk, ksize));
            continue;
        }
        float s0 = sk[0] * srow[skew(sx)];
        float s1 = sk[0] * srow[skew(sx1)];
        for (int t = 1; t < ksize; t++) {
            const float f = sk[t];
            s0 = s0 + f * srow[skew(sx + t)];
            s1 = s1 + f * srow[skew(sx1 + t)];
        }
        dst[dx] = make_float2(s0, s1);
    }
}

// ---------------------------------------------------------------------------------------------
// k_level_hpass_multi: stage A pass 1 for ALL levels >= 1 in one launch.  One block per frame row:
// the uint8 row is read once, converted and staged in LDS with a reflect-101 border of the widest
// kernel, then every level's sampled columns are filtered out of LDS (same arithmetic and order
// as k_level_hpass).  Work item = (level, level column, left/right source column).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_hpass_multi(const uint8_t *__restrict__ frames, size_t frame_stride,
                                                            int W, int H, HLevels L)
{
    extern __shared__ float srow[];
    const int y = blockIdx.y;
    const uint8_t *row = frames + (size_t)blockIdx.z * frame_stride + (size_t)y * W;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const int rmax = L.rmax, ext = W + 2 * rmax;
    float *sk = srow + skew(ext) + 1;        // all levels' kernel taps, back to back
    if ((rmax & 3) == 0 && (W & 3) == 0 && ((uintptr_t)row & 3) == 0) {
        // interior four pixels at a time: one dword load, four v_cvt_f32_ubyteN, one 16-byte LDS write (the border
        // width is a multiple of 4, so a group never straddles a skew step); the two borders element by element
        for (int c = 4 * tid; c < W; c += 4 * NT) {
            const uint32_t d = *reinterpret_cast<const uint32_t *>(row + c);
            *reinterpret_cast<float4 *>(&srow[skew(c + rmax)]) =
                make_float4((float)(d & 255u), (float)((d >> 8) & 255u), (float)((d >> 16) & 255u), (float)(d >> 24));
        }
        for (int i = tid; i < 2 * rmax; i += NT) {
            const int e = i < rmax ? i : W + i;          // extended index: left border, then right border
            srow[skew(e)] = (float)row[reflect101(e - rmax, W)];
        }
    } else
        for (int i = tid; i < ext; i += NT) srow[skew(i)] = (float)row[reflect101(i - rmax, W)];
    {
        int off = 0;
        for (int l = 0; l < L.n; l++) {
            for (int i = tid; i < L.lv[l].ksize; i += NT) sk[off + i] = L.lv[l].kern[i];
            off += L.lv[l].ksize;
        }
    }
    __syncthreads();
    int koff = 0;
    for (int l = 0; l < L.n; l++) {
        const HLevel lv = L.lv[l];
        const int r = lv.ksize >> 1;
        const float *kk = sk + koff;
        koff += lv.ksize;
        float2 *dst = reinterpret_cast<float2 *>(lv.dst) + ((size_t)blockIdx.z * H + y) * lv.dw;
        for (int dx = tid; dx < lv.dw; dx += NT) {
            const int sx = lv.xofs[dx];
            const int p = sx - r + rmax;             // extended index of tap 0 of the left column
            float acc0, acc1;
            if (L.symm && (lv.ksize == 3 || lv.ksize == 5)) {
                float a[6];
                for (int t = 0; t <= lv.ksize; t++) a[t] = srow[skew(p + t)];
                acc0 = row_small_symm(a, kk, lv.ksize);
                acc1 = sx + 1 < W ? row_small_symm(a + 1, kk, lv.ksize) : acc0;
            } else if (sx + 1 < W) {
                // right column = left column + 1: its tap t reads what the left column's tap t+1 reads.
                // Taps go in batches of 8 so that the 16 LDS reads of a batch are in flight together
                // (one wait per batch instead of one LDS round trip per tap).
                float v0 = srow[skew(p)], v1 = srow[skew(p + 1)];
                acc0 = kk[0] * v0;
                acc1 = kk[0] * v1;
                int t = 1;
                for (; t + 8 <= lv.ksize; t += 8) {
                    float vv[8], ff[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { vv[u] = srow[skew(p + t + 1 + u)]; ff[u] = kk[t + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        v0 = v1;
                        v1 = vv[u];
                        acc0 = acc0 + ff[u] * v0;
                        acc1 = acc1 + ff[u] * v1;
                    }
                }
                for (; t < lv.ksize; t++) {
                    v0 = v1;
                    v1 = srow[skew(p + t + 1)];
                    const float f = kk[t];
                    acc0 = acc0 + f * v0;
                    acc1 = acc1 + f * v1;
                }
            } else {                                 // clamped: both columns are W-1
                acc0 = kk[0] * srow[skew(p)];
                for (int t = 1; t < lv.ksize; t++) acc0 = acc0 + kk[t] * srow[skew(p + t)];
                acc1 = acc0;
            }
            dst[dx] = make_float2(acc0, acc1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_level_direct<S, K>: the whole of stage A for one level whose size is EXACTLY 1/S of the frame in both
// directions (S = 2, 4, 8 with pyr_scale 0.5 and sizes divisible by S): uint8 frame in, level image out,
// nothing in between goes to HBM.  Same arithmetic, in the same order, as row pass + column pass + resize:
//   row pass    t[v][c]  = k[0]*f[c-r] + k[1]*f[c-r+1] + ...                     (left to right, reflect-101)
//   column pass b[row]   = k[r]*t[row] + sum_i k[r+i]*(t[row+i] + t[row-i])      (i = 1..r)
//   resize      out      = (b00*a0 + b01*a1)*w0 + (b10*a0 + b11*a1)*w1,  a = w = 1/2 here
// A level pixel (x, y) samples source columns c0 = S*x + S/2 - 1, c0 + 1 and source rows sy = S*y + S/2 - 1,
// sy + 1.  One thread per level column marches down a strip of level rows; the row-pass results of the last
// 3*S source rows live in a register ring (K + 1 <= 3*S rows are needed per output) that is addressed
// statically because the marching loop is unrolled by the ring period of three output rows; each output
// row costs S new row passes (K + 1 byte loads each).
// ---------------------------------------------------------------------------------------------
template <int K> struct TapsArg { float k[K]; };

__device__ __forceinline__ int reflect101_once(int p, int len)   // valid for -len < p < 2*len - 1
{
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

// EDGE = false: level columns 1 .. w-2, whose K+1 source bytes lie inside the row: they are fetched as aligned
//               dwords (the window start S*x + S/2 - 1 - r has the same alignment for every x when S >= 4) and
//               unpacked with v_cvt_f32_ubyteN.
// EDGE = true : level columns 0 and w-1 (the only ones whose taps cross the left / right border for these
//               (S, K)), byte loads at reflect-101 indices; thread = (strip, column), a tiny second launch.
// SYMM (K == 3 only): the row pass in SymmRowSmallFilter's order (row_small_symm) -- a template parameter, not a run-time flag: a
// run-time branch inside the row pass kept the compiler from batching the rows' loads (1.15 -> 1.63 ms per 1024 frames at level 1).
template <int S, int K, bool EDGE, bool SYMM>
__global__ __launch_bounds__(EDGE ? 64 : 256) void k_level_direct(const uint8_t *__restrict__ frames, size_t frame_stride,
                                                                  int W, int H, TapsArg<K> taps, float *__restrict__ I, int w,
                                                                  int h, int strip)
{
    constexpr int R = 3 * S, r = K / 2;
    static_assert(K + 1 <= R && (K & 1), "ring of 3*S rows must hold the K+1 rows of one output");
    static_assert(!SYMM || K == 3, "the small-symmetric order exists for 3 taps here");
    int x, dy0;
    if (EDGE) {
        const int sidx = blockIdx.y * 32 + (threadIdx.x >> 1);
        x = (threadIdx.x & 1) ? w - 1 : 0;
        dy0 = sidx * strip;
        if (dy0 >= h || (w == 1 && (threadIdx.x & 1))) return;
    } else {
        x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
        dy0 = blockIdx.y * strip;
        if (x > w - 2) return;
    }
    const int dy1 = min(dy0 + strip, h);
    const uint8_t *f = frames + (size_t)blockIdx.z * frame_stride;
    float *out = I + (size_t)blockIdx.z * w * h;
    const int cl = S * x + S / 2 - 1 - r;            // source column of tap 0 of the left sampled column
    float ring[R][2];

    // The K+1 source bytes of a row pass arrive as ND dwords: aligned ones for S >= 4 (the window start has the same alignment for
    // every x), one unaligned dword for S = 2 (K + 1 = 4 bytes from cl; global loads need no alignment); EDGE: bytes at reflected indices.
    constexpr int OFF = S >= 4 ? (((S / 2 - 1 - r) % 4) + 4) % 4 : 0;     // cl - (cl rounded down to a multiple of 4)
    constexpr int ND = EDGE ? K + 1 : (OFF + K + 1 + 3) / 4;
    auto rowload = [&](int v, uint32_t (&d)[ND]) {
        const uint8_t *row = f + (size_t)reflect101_once(v, H) * W;
        if (EDGE) {
#pragma unroll
            for (int t = 0; t <= K; t++) d[t] = row[reflect101_once(cl + t, W)];
        } else {
#pragma unroll
            for (int i = 0; i < ND; i++) __builtin_memcpy(&d[i], row + (cl - OFF) + 4 * i, 4);
        }
    };
    auto rowcalc = [&](const uint32_t (&d)[ND], float &o0, float &o1) {
        float px[K + 1];
#pragma unroll
        for (int t = 0; t <= K; t++)
            px[t] = EDGE ? (float)d[t] : (float)((d[(OFF + t) / 4] >> (8 * ((OFF + t) % 4))) & 255u);
        if (SYMM) {                                    // SymmRowSmallFilter's order (row_small_symm)
            o0 = px[1] * taps.k[1] + (px[0] + px[2]) * taps.k[2];
            o1 = px[2] * taps.k[1] + (px[1] + px[3]) * taps.k[2];
            return;
        }
        float a0 = taps.k[0] * px[0], a1 = taps.k[0] * px[1];
#pragma unroll
        for (int t = 1; t < K; t++) {
            a0 = a0 + taps.k[t] * px[t];
            a1 = a1 + taps.k[t] * px[t + 1];
        }
        o0 = a0;
        o1 = a1;
    };
    const int vbase = S * dy0 + S / 2 - 1 - r;       // virtual source row of ring slot 0
    // warm-up: the K+1-S rows the first output needs beyond its own S new ones
    {
        constexpr int NW = K + 1 - S;
        uint32_t wd[NW > 0 ? NW : 1][ND];
#pragma unroll
        for (int j = 0; j < NW; j++) rowload(vbase + j, wd[j]);
#pragma unroll
        for (int j = 0; j < NW; j++) rowcalc(wd[j], ring[j % R][0], ring[j % R][1]);
    }
    // The S new source rows of an output row are LOADED one output row ahead: their latency then overlaps the previous row's
    // row passes, column pass and store instead of standing in front of every row pass (the kernel is bound by exactly that
    // latency: its waves wait 84 % of their cycles).
    // (S = 2 only: at S = 4 and 8 the 2 x S x ND registers of such a prefetch cost more occupancy than the overlap returns --
    // 0.39 / 0.46 instead of 0.37 / 0.38 ms per 512 frames -- and the compiler batches the S row loads of an output row by itself.)
    constexpr bool PREFETCH = S == 2 && !EDGE;
    uint32_t nxt[PREFETCH ? S : 1][ND];
    if (PREFETCH && dy0 < dy1) {
#pragma unroll
        for (int jj = 0; jj < S; jj++) rowload(vbase + K + 1 - S + jj, nxt[PREFETCH ? jj : 0]);
    }
    for (int dy = dy0; dy < dy1; dy += 3) {
        const int vg = vbase + S * (dy - dy0);       // virtual row of slot 0 for this group of three outputs
#pragma unroll
        for (int b = 0; b < 3; b++) {
            if (dy + b < dy1) {
                uint32_t cur[S][ND];
                if (PREFETCH) {
#pragma unroll
                    for (int jj = 0; jj < S; jj++)
#pragma unroll
                        for (int i = 0; i < ND; i++) cur[jj][i] = nxt[PREFETCH ? jj : 0][i];
                    if (dy + b + 1 < dy1) {
#pragma unroll
                        for (int jj = 0; jj < S; jj++) rowload(vg + S * (b + 1) + K + 1 - S + jj, nxt[PREFETCH ? jj : 0]);
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < S; jj++) rowload(vg + S * b + K + 1 - S + jj, cur[jj]);
                }
#pragma unroll
                for (int jj = 0; jj < S; jj++) {
                    const int j = K + 1 - S + jj;
                    rowcalc(cur[jj], ring[(S * b + j) % R][0], ring[(S * b + j) % R][1]);
                }
                float bb[2][2];
#pragma unroll
                for (int q = 0; q < 2; q++)          // sampled rows sy (q = 0) and sy + 1 (q = 1)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        float acc = taps.k[r] * ring[(S * b + r + q) % R][c];
#pragma unroll
                        for (int i = 1; i <= r; i++)
                            acc = acc + taps.k[r + i] * (ring[(S * b + r + q + i) % R][c] + ring[(S * b + r + q - i) % R][c]);
                        bb[q][c] = acc;
                    }
                const float row0 = bb[0][0] * 0.5f + bb[0][1] * 0.5f;
                const float row1 = bb[1][0] * 0.5f + bb[1][1] * 0.5f;
                out[(size_t)(dy + b) * w + x] = row0 * 0.5f + row1 * 0.5f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_level_hdirect<S, K>: stage A row pass for a level whose WIDTH is exactly 1/S of the frame's, S = 16, 32, 64
// (K = 39, 79, 159 taps), straight from the uint8 frame: no LDS staging, no barrier.  One thread per (frame row,
// level column): the K+1 source bytes of its two sampled columns start at S*x + S/2 - 1 - K/2, which is a multiple
// of 4 for these (S, K), so they arrive as (K+1)/4 aligned dwords and are unpacked with v_cvt_f32_ubyteN; taps are
// kernel arguments (SGPRs).  Same arithmetic and order as k_level_hpass*.  EDGE: columns 0 and w-1, whose taps
// cross the frame border, in a second small launch -- blockIdx.x picks the side, so the reflect-101 source index of every
// tap is a compile-time offset from the start (left) or the end (right) of the row: the bytes arrive as aligned dwords
// all issued at once, like the interior's (80 dependent byte loads per thread made this launch slower than the interior's).
// ---------------------------------------------------------------------------------------------
constexpr int HD_ROWS = 4;

template <int S, int K, bool EDGE>
__global__ __launch_bounds__(EDGE ? 64 : 128) void k_level_hdirect(const uint8_t *__restrict__ frames, size_t frame_stride, int W,
                                                                   int H, TapsArg<K> taps, float2 *__restrict__ tmp, int w)
{
    constexpr int r = K / 2;
    static_assert((S / 2 - 1 - r) % 4 == 0 && (K + 1) % 4 == 0, "window must start on a dword and span whole dwords");
    int x, y;
    if (EDGE) {
        x = blockIdx.x ? w - 1 : 0;
        y = blockIdx.y * 64 + threadIdx.x;
    } else {
        x = 1 + blockIdx.x * blockDim.x + threadIdx.x;
        y = blockIdx.y * HD_ROWS;
        if (x > w - 2) return;
    }
    const int cl = S * x + S / 2 - 1 - r;
    // interior threads take HD_ROWS consecutive frame rows each (fewer, longer blocks; the next row's loads overlap this row's sums)
    for (int rr = 0; rr < (EDGE ? 1 : HD_ROWS); rr++, y++) {
    if (y >= H) return;
    const uint8_t *row = frames + (size_t)blockIdx.z * frame_stride + (size_t)y * W;
    float a0 = 0.f, a1 = 0.f;
    if (EDGE) {
        // left:  tap t reads column |c0 + t| with c0 = S/2-1-r < 0            -> bytes 0 .. c0+K of the row
        // right: tap t reads column W + p, p = cR + t, cR = -S/2-1-r, reflected to W - 2 - p where p >= 0
        //                                                                     -> bytes W+cR .. W-1 of the row
        constexpr int c0 = S / 2 - 1 - r, cR = -S / 2 - 1 - r;
        static_assert(c0 < 0 && (-cR) % 4 == 0 && c0 + K < 2 * S && cR + K < -cR, "edge windows must fold once, inside 2*S columns");
        constexpr int NDL = (c0 + K) / 4 + 1, NDR = -cR / 4;
        uint32_t d[NDL > NDR ? NDL : NDR];
        float px[K + 1];
        if (blockIdx.x == 0) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(row);
#pragma unroll
            for (int i = 0; i < NDL; i++) d[i] = q[i];
#pragma unroll
            for (int t = 0; t <= K; t++) {
                const int idx = c0 + t < 0 ? -(c0 + t) : c0 + t;
                px[t] = (float)((d[idx / 4] >> (8 * (idx % 4))) & 255u);
            }
        } else {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(row + (W + cR));
#pragma unroll
            for (int i = 0; i < NDR; i++) d[i] = q[i];
#pragma unroll
            for (int t = 0; t <= K; t++) {
                const int p = cR + t;
                const int idx = (p >= 0 ? -2 - p : p) - cR;
                px[t] = (float)((d[idx / 4] >> (8 * (idx % 4))) & 255u);
            }
        }
        a0 = taps.k[0] * px[0];
        a1 = taps.k[0] * px[1];
#pragma unroll
        for (int t = 1; t < K; t++) {
            a0 = a0 + taps.k[t] * px[t];
            a1 = a1 + taps.k[t] * px[t + 1];
        }
    } else {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(row + cl);
        uint32_t d[(K + 1) / 4];
#pragma unroll
        for (int i = 0; i < (K + 1) / 4; i++) d[i] = q[i];
#pragma unroll
        for (int t = 0; t <= K; t++) {
            const float v = (float)((d[t / 4] >> (8 * (t % 4))) & 255u);
            if (t == 0) a0 = taps.k[0] * v;
            else if (t < K) a0 = a0 + taps.k[t] * v;
            if (t == 1) a1 = taps.k[0] * v;
            else if (t > 1) a1 = a1 + taps.k[t - 1] * v;
        }
    }
    tmp[((size_t)blockIdx.z * H + y) * w + x] = make_float2(a0, a1);
    }
}

bool level_hdirect_supported(const void *frames, int W, int w, int ksize)
{
    if (w < 2 || ((uintptr_t)frames & 3) || (W & 3)) return false;
    return (W == 16 * w && ksize == 39) || (W == 32 * w && ksize == 79) || (W == 64 * w && ksize == 159);
}

template <int S, int K>
static void launch_level_hdirect_sk(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                                    const float *h_kern, float *tmp, int w)
{
    TapsArg<K> taps;
    for (int i = 0; i < K; i++) taps.k[i] = h_kern[i];
    float2 *t2 = reinterpret_cast<float2 *>(tmp);
    if (w > 2) {
        const int nt = w - 2 > 64 ? 128 : 64;
        dim3 grid((unsigned)((w - 2 + nt - 1) / nt), (unsigned)((H + HD_ROWS - 1) / HD_ROWS), nframes);
        hipLaunchKernelGGL((k_level_hdirect<S, K, false>), grid, dim3(nt), 0, s, frames, frame_stride, W, H, taps, t2, w);
    }
    dim3 egrid(2, (unsigned)((H + 63) / 64), nframes);     // x: left / right border column (w >= 2 here)
    hipLaunchKernelGGL((k_level_hdirect<S, K, true>), egrid, dim3(64), 0, s, frames, frame_stride, W, H, taps, t2, w);
}

void launch_level_hdirect(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                          const float *h_kern, int ksize, float *tmp, int w)
{
    if (W == 16 * w && ksize == 39) launch_level_hdirect_sk<16, 39>(s, frames, frame_stride, W, H, nframes, h_kern, tmp, w);
    else if (W == 32 * w && ksize == 79) launch_level_hdirect_sk<32, 79>(s, frames, frame_stride, W, H, nframes, h_kern, tmp, w);
    else if (W == 64 * w && ksize == 159) launch_level_hdirect_sk<64, 159>(s, frames, frame_stride, W, H, nframes, h_kern, tmp, w);
}

static inline unsigned cdivu(int a, int b) { return (unsigned)((a + b - 1) / b); }
int best_strip_units(int nunits, int unit, int warm, int blocks_per_strip_row, int blocks_per_cu);

// Direct level build: supported when the level is exactly 1/S of the frame (S = 2, 4, 8), the kernel has the
// size getGaussianKernel gives for that scale (3, 9, 19 taps) and the frames are aligned for dword loads.
bool level_direct_supported(const void *frames, int W, int H, int w, int h, int ksize)
{
    if (w < 2 || h < 1 || ((uintptr_t)frames & 7)) return false;
    return (W == 2 * w && H == 2 * h && ksize == 3) || (W == 4 * w && H == 4 * h && ksize == 9) ||
           (W == 8 * w && H == 8 * h && ksize == 19);
}

template <int S, int K, bool SYMM>
static void launch_level_direct_sk(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                                   const float *h_kern, float *I, int w, int h)
{
    TapsArg<K> taps;
    for (int i = 0; i < K; i++) taps.k[i] = h_kern[i];
    // strips of a multiple of three level rows; each strip re-does K+1-S row passes of warm-up
    const int units = (h + 2) / 3;
    if (w > 2) {
        // narrow levels: smaller blocks keep the lanes busy (a 80-column level fills 78 of 128 lanes, not of 256)
        const int nt = w - 2 > 128 ? 256 : (w - 2 > 64 ? 128 : 64);
        const int strip = 3 * best_strip_units(units, 3 * S, K + 1 - S, (int)cdivu(w - 2, nt) * nframes, 8 * (256 / nt));
        dim3 grid(cdivu(w - 2, nt), cdivu(h, strip), nframes);
        hipLaunchKernelGGL((k_level_direct<S, K, false, SYMM>), grid, dim3(nt), 0, s, frames, frame_stride, W, H, taps, I, w, h, strip);
    }
    // border columns 0 and w-1: 32 strips of at least 12 rows per block of 64 threads
    int estrip = 3 * ((units + 31) / 32);
    if (estrip < 12) estrip = 12;
    dim3 egrid(1, cdivu((int)cdivu(h, estrip), 32), nframes);
    hipLaunchKernelGGL((k_level_direct<S, K, true, SYMM>), egrid, dim3(64), 0, s, frames, frame_stride, W, H, taps, I, w, h, estrip);
}

void launch_level_direct(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                         const float *h_kern, int ksize, float *I, int w, int h, int symm)
{
    if (W == 2 * w && ksize == 3) {
        if (symm) launch_level_direct_sk<2, 3, true>(s, frames, frame_stride, W, H, nframes, h_kern, I, w, h);
        else launch_level_direct_sk<2, 3, false>(s, frames, frame_stride, W, H, nframes, h_kern, I, w, h);
    } else if (W == 4 * w && ksize == 9) launch_level_direct_sk<4, 9, false>(s, frames, frame_stride, W, H, nframes, h_kern, I, w, h);
    else if (W == 8 * w && ksize == 19) launch_level_direct_sk<8, 19, false>(s, frames, frame_stride, W, H, nframes, h_kern, I, w, h);
}

// The fused kernel is instantiated for the window half-widths m = winsize/2 = 3..10 (winsize 6..21); other window
// sizes take the generic unfused kernels.
bool flow_iter_supported(int winsize)
{
    const int m = winsize / 2;
    return m >= 3 && m <= 10;
}

template <int M_>
static void launch_flow_iter_m(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w,
                               int h, int npairs, int winsize, int mode, const float *coarse, int cw, int ch,
                               const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul, int tile_mode)
{
    constexpr int OUTW = march_out_width(M_);
    constexpr int B = 2 * M_ + 1;
    // strips start on block boundaries of the blocked column sums (multiples of B rows); their
    // number is chosen to minimise (rounds of resident blocks) x (rows marched per block)
    const int nblk = (h + B - 1) / B;
    int strip_units = best_strip_units(nblk, B, B - 1, (int)cdivu(w, OUTW) * npairs, 3);
#ifdef OFARN_EXP_STRIP_ENV      // experiment: strip height of the level-0 launch from the environment (units of B rows)
    if (const char *e = getenv("OFARN_FI_STRIP_UNITS")) if (w >= 1920 && atoi(e) > 0) strip_units = atoi(e);
#endif
    const int strip_h = B * strip_units;
    dim3 grid(cdivu(w, OUTW), cdivu(h, strip_h), npairs);
    // a grid that leaves most CUs without a block is latency bound: the tile kernel (kernels_tile.hip) does the same arithmetic
    // with all of a tile's gathers in flight at once
    if (flow_iter_tile_supported(winsize) && flow_iter_tile_preferred((long)grid.x * grid.y * grid.z, tile_mode)) {
        launch_flow_iter_tile(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, d_xa, mul);
        return;
    }
    const double scale = 1. / ((double)winsize * winsize);
    // resize_tables(): inv_scale = (double)dsize / ssize; scale = 1. / inv_scale
    const double yscale = ch > 0 ? 1. / ((double)h / ch) : 1.;
    (void)d_yofs; (void)d_ya;     // the row table is recomputed in the kernel (resize_coord)
    UpsampleArgs up{reinterpret_cast<const float2 *>(coarse), cw, ch, d_xofs, d_xa, yscale, mul, nullptr};
#ifdef OFARN_STAMPS
#define OFARN_EXP_PART 5
#include "experiments/flow_iter_stamps.inc"
#endif
    const float2 *fin = reinterpret_cast<const float2 *>(flow_in);
    float2 *fout = reinterpret_cast<float2 *>(flow_out);
    if (mode == 0)
        hipLaunchKernelGGL((k_flow_iter<M_, 0>), grid, dim3(FI_THREADS), 0, s, R, fstep, fin, fout, w, h, strip_h, scale, up);
    else if (mode == 1)
        hipLaunchKernelGGL((k_flow_iter<M_, 1>), grid, dim3(FI_THREADS), 0, s, R, fstep, fin, fout, w, h, strip_h, scale, up);
    else
        hipLaunchKernelGGL((k_flow_iter<M_, 2>), grid, dim3(FI_THREADS), 0, s, R, fstep, fin, fout, w, h, strip_h, scale, up);
}

// mode 0: zero input flow; 1: upsample from coarse (up_* valid); 2: read flow_in.
void launch_flow_iter(hipStream_t s, const float *R, int fstep, const float *flow_in, float *flow_out, int w,
                      int h, int npairs, int winsize, int mode, const float *coarse, int cw, int ch,
                      const int *d_xofs, const float *d_xa, const int *d_yofs, const float *d_ya, float mul, int tile_mode)
{
#define OFARN_FI_CASE(M)                                                                                          \
    case M:                                                                                                       \
        launch_flow_iter_m<M>(s, R, fstep, flow_in, flow_out, w, h, npairs, winsize, mode, coarse, cw, ch, d_xofs, \
                              d_xa, d_yofs, d_ya, mul, tile_mode);                                                 \
        break;
    switch (winsize / 2) {
        OFARN_FI_CASE(3) OFARN_FI_CASE(4) OFARN_FI_CASE(5) OFARN_FI_CASE(6) OFARN_FI_CASE(7) OFARN_FI_CASE(8) OFARN_FI_CASE(9)
        OFARN_FI_CASE(10)
        default: break;
    }
#undef OFARN_FI_CASE
}

}  // namespace ofarn

namespace ofarn {

// Strip height in `unit`-row steps for a marching kernel: each block marches strip + warm rows, the
// GPU holds blocks_per_cu * CUs blocks at once; minimise ceil(blocks / resident) * rows-per-block.
// CU count of the CURRENT device, cached per device ordinal (a process may drive several GPUs)
int march_cu_count()
{
    static int ncu_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int ncu = __atomic_load_n(&ncu_of[dev], __ATOMIC_RELAXED);
    if (!ncu) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        ncu = v;
        __atomic_store_n(&ncu_of[dev], v, __ATOMIC_RELAXED);
    }
    return ncu;
}

int best_strip_units(int nunits, int unit, int warm, int blocks_per_strip_row, int blocks_per_cu)
{
    const int ncu = march_cu_count();
    const long resident = (long)ncu * blocks_per_cu;
    long best_cost = -1;
    int best = nunits;
    for (int per = 1; per <= nunits; per++) {
        const int nstrips = (nunits + per - 1) / per;
        const long blocks = (long)nstrips * blocks_per_strip_row;
        const long rounds = (blocks + resident - 1) / resident;
        const long cost = rounds * ((long)per * unit + warm);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && per > best)) { best_cost = cost; best = per; }
    }
    return best;
}

bool polyexp_march_supported(int poly_n) { return poly_n == 5 || poly_n == 7; }   // the two radii OpenCV's documentation names

template <int N>
static void launch_polyexp_march_n(hipStream_t s, const void *src, size_t src_stride, int src_is_u8, float *R, int w, int h,
                                   int nframes, const PolyCoef &c, const float *blur3)
{
    constexpr int OUTW = pe_out_width(N);
    const int strip_h = best_strip_units(h, 1, 2 * N, (int)cdivu(w, OUTW) * nframes, OFARN_PE_WAVES * FI_THREADS / OFARN_PE_THREADS);
    dim3 grid(cdivu(w, OUTW), cdivu(h, strip_h), nframes);
    const int nt = nt_hint((size_t)nframes * w * h * 20);
    if (src_is_u8)
        hipLaunchKernelGGL((k_polyexp_march<N, 1>), grid, dim3(OFARN_PE_THREADS), 0, s, src, src_stride, R, w, h, strip_h, c,
                           blur3[0], blur3[1], blur3[2], nt);
    else
        hipLaunchKernelGGL((k_polyexp_march<N, 0>), grid, dim3(OFARN_PE_THREADS), 0, s, src, src_stride, R, w, h, strip_h, c,
                           0.f, 0.f, 0.f, nt);
}

// src_is_u8 = 1: level 0, src = uint8 frames (stride in bytes = elements); else float level images.
void launch_polyexp_march(hipStream_t s, const void *src, size_t src_stride, int src_is_u8, float *R, int w, int h,
                          int nframes, const PolyCoef &c, const float *blur3)
{
    if (c.n == 7) launch_polyexp_march_n<7>(s, src, src_stride, src_is_u8, R, w, h, nframes, c, blur3);
    else launch_polyexp_march_n<5>(s, src, src_stride, src_is_u8, R, w, h, nframes, c, blur3);
}

void launch_level_hpass_lds(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                            const float *d_kern, int ksize, const int *d_xofs, int dw, float *tmp, int symm)
{
    const int r = ksize >> 1;
    const int ext = W + 2 * r;
    const size_t lds = sizeof(float) * (size_t)(ext + (ext >> 5) + 2 + ksize);
    dim3 grid(1, H, nframes);
    hipLaunchKernelGGL(k_level_hpass_lds, grid, dim3(256), lds, s, frames, frame_stride, W, H, d_kern, ksize, d_xofs,
                       dw, reinterpret_cast<float2 *>(tmp), symm);
}

}  // namespace ofarn

namespace ofarn {

size_t hpass_multi_lds_bytes(int W, int rmax)
{
    const int ext = W + 2 * rmax;
    return sizeof(float) * (size_t)(ext + (ext >> 5) + 2 + 12 * (2 * rmax + 1));   // row + taps of <= 12 levels
}

void launch_level_hpass_multi(hipStream_t s, const uint8_t *frames, size_t frame_stride, int W, int H, int nframes,
                              const HLevels &L)
{
    dim3 grid(1, H, nframes);
#ifndef OFARN_HP_THREADS
#define OFARN_HP_THREADS 256
#endif
    int maxdw = 0;
    for (int i = 0; i < L.n; i++) maxdw = L.lv[i].dw > maxdw ? L.lv[i].dw : maxdw;
    const int nt = maxdw <= 128 ? OFARN_HP_THREADS : 256;   // few columns: smaller blocks leave no idle waves
    hipLaunchKernelGGL(k_level_hpass_multi, grid, dim3(nt), hpass_multi_lds_bytes(W, L.rmax), s, frames, frame_stride,
                       W, H, L);
}

}  // namespace ofarn
