// farneback_device.h -- device functions shared by the generic and the fused kernels.
#pragma once
#include "ofarn_internal.h"

namespace ofarn {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// GaussianBlur row pass of a symmetric float kernel with ksize 3 or 5: filter.simd.hpp SymmRowSmallFilter<float, float>,
//   ksize 3: S[0]*k0 + (S[-1] + S[1])*k1;   ksize 5: ... + (S[-2] + S[2])*k2      (k0 = centre tap; oracle OFO_ROW_SMALL_SYMM)
// px: the ksize source values left to right; k: the ksize taps (k[ksize/2] is the centre).
__device__ __forceinline__ float row_small_symm(const float *px, const float *k, int ksize)
{
    const int r = ksize >> 1;
    float s = px[r] * k[r] + (px[r - 1] + px[r + 1]) * k[r + 1];
    if (ksize == 5) s = s + (px[0] + px[4]) * k[r + 2];
    return s;
}

// R layout ("4+1"): per frame, channels 0..3 of pixel o as one aligned float4 at ((float4*)R)[o],
// channel 4 as a float at R[4*npx + o].  One 16-byte and one 4-byte load per pixel instead of five
// 4-byte loads: the texture addresser works per lane and cycle, so wide per-lane loads are what
// moves bytes (measured: the 26 dword loads of the planar layout took a third of flow_iter's time
// just to issue).
__device__ __forceinline__ void load_r(const float *__restrict__ Rf, size_t npx, unsigned o, float out[5])
{
    // (uniform base) + (32-bit byte offset): scalar-base addressing, one VGPR per address
    const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Rf) + o * 16u);
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
    out[4] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(Rf + 4 * npx) + o * 4u);
}
// Non-temporal variant of the R store.  A level's R (20 B per pixel and frame) is written once and read back by a LATER
// kernel; when a wave's worth of it is far larger than the 256 MB Infinity Cache none of it survives until then, and the `nt`
// hint lets the write stream pass without allocating there: the level-0 polynomial expansion runs 6.3 -> 5.9 ms per 512 frames
// with it (same box, alternating runs).  At the coarse levels, whose R does fit, plain stores stay (the launcher decides by
// size); the 8-B flow stores of the fused iteration measured the same with and without the hint and keep plain stores.
typedef float ofarn_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_r_nt(float *__restrict__ Rf, size_t npx, unsigned o, const float v[5])
{
    const ofarn_f4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<ofarn_f4 *>(reinterpret_cast<char *>(Rf) + o * 16u));
    __builtin_nontemporal_store(v[4], reinterpret_cast<float *>(reinterpret_cast<char *>(Rf + 4 * npx) + o * 4u));
}
__device__ __forceinline__ void store_r(float *__restrict__ Rf, size_t npx, unsigned o, const float v[5])
{
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(Rf) + o * 16u) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float *>(reinterpret_cast<char *>(Rf + 4 * npx) + o * 4u) = v[4];
}

// The two frames of pair p in an R buffer: R0 = frame p * |fstep|, R1 = the frame behind it (fstep > 0: batches; 1 = video
// order, 2 = independent pairs) or in FRONT of it (fstep < 0: the streaming session keeps two slots per level and writes the
// newest frame into the slot the older one does not occupy, so every other turn the later frame sits in the lower slot).
__device__ __forceinline__ void pair_frames(const float *R, int fstep, size_t p, size_t npx, const float *&R0, const float *&R1)
{
    const size_t st = r_frame_stride(npx);
    R0 = R + p * (size_t)(fstep < 0 ? -fstep : fstep) * st;
    R1 = fstep < 0 ? R0 - st : R0 + st;
}

// FarnebackUpdateMatrices for one pixel (optflowgf.cpp), all float32, no FMA contraction.
// R0, R1: the pair's two frames in the 4+1 layout; (dx, dy) the current flow at (x, y).
__device__ __forceinline__ void update_matrices_px(const float *__restrict__ R0, const float *__restrict__ R1,
                                                    size_t npx, int w, int h, int x, int y, float dx,
                                                    float dy, float out[5])
{
    float fx = (float)x + dx, fy = (float)y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= (float)x1; fy -= (float)y1;
    const unsigned o = (unsigned)y * (unsigned)w + (unsigned)x;
    float c0[5];
    load_r(R0, npx, o, c0);
    float r2, r3, r4, r5, r6;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const unsigned q = (unsigned)y1 * (unsigned)w + (unsigned)x1;
        float t00[5], t01[5], t10[5], t11[5];
        load_r(R1, npx, q, t00); load_r(R1, npx, q + 1, t01);
        load_r(R1, npx, q + w, t10); load_r(R1, npx, q + w + 1, t11);
        r2 = a00 * t00[0] + a01 * t01[0] + a10 * t10[0] + a11 * t11[0];
        r3 = a00 * t00[1] + a01 * t01[1] + a10 * t10[1] + a11 * t11[1];
        r4 = a00 * t00[2] + a01 * t01[2] + a10 * t10[2] + a11 * t11[2];
        r5 = a00 * t00[3] + a01 * t01[3] + a10 * t10[3] + a11 * t11[3];
        r6 = a00 * t00[4] + a01 * t01[4] + a10 * t10[4] + a11 * t11[4];
        r4 = (c0[2] + r4) * 0.5f;
        r5 = (c0[3] + r5) * 0.5f;
        r6 = (c0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = c0[2];
        r5 = c0[3];
        r6 = c0[4] * 0.5f;
    }
    r2 = (c0[0] - r2) * 0.5f;
    r3 = (c0[1] - r3) * 0.5f;
    r2 = r2 + (r4 * dy + r6 * dx);
    r3 = r3 + (r6 * dy + r5 * dx);
    if ((unsigned)(x - kBorder) >= (unsigned)(w - kBorder * 2) ||
        (unsigned)(y - kBorder) >= (unsigned)(h - kBorder * 2)) {
        const float tab[kBorder] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const float scale = (x < kBorder ? tab[x] : 1.f) * (x >= w - kBorder ? tab[w - x - 1] : 1.f) *
                            (y < kBorder ? tab[y] : 1.f) * (y >= h - kBorder ? tab[h - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    out[0] = r4 * r4 + r6 * r6;
    out[1] = (r4 + r5) * r6;
    out[2] = r5 * r5 + r6 * r6;
    out[3] = r4 * r2 + r6 * r3;
    out[4] = r6 * r2 + r5 * r3;
}


// Loads/stores at (uniform base pointer + 32-bit byte offset): lets the compiler use the
// scalar-base addressing form (one VGPR per address).  Byte offsets must stay below 4 GiB per plane.
__device__ __forceinline__ float ldg_f32(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float2 ldg_f2(const float2 *base, unsigned byte_off)
{
    return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void stg_f2(float2 *base, unsigned byte_off, float2 v)
{
    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// ---- FarnebackUpdateMatrices split in two for software pipelining -------------------------------
// gather_issue() only computes addresses and issues the loads (flow-dependent bilinear taps of R1
// plus R0 at the pixel); matrices_finish() does the arithmetic.  Together they perform exactly the
// operations of update_matrices_px, in the same order.
struct GatherRaw {
    float dx, dy, fx, fy;
    float r0[5];
    float t00[5], t01[5], t10[5], t11[5];
    int inb;
};

__device__ __forceinline__ void gather_issue(const float *__restrict__ R0, const float *__restrict__ R1, size_t npx,
                                             int w, int h, int x, int y, float dx, float dy, GatherRaw &g)
{
    float fx = (float)x + dx, fy = (float)y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= (float)x1; fy -= (float)y1;
    g.dx = dx; g.dy = dy; g.fx = fx; g.fy = fy;
    const unsigned o = (unsigned)y * (unsigned)w + (unsigned)x;
    load_r(R0, npx, o, g.r0);
    g.inb = ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) ? 1 : 0;
    // out-of-range taps are never used; read pixel 0 instead so the loads stay unconditional
    const unsigned q = g.inb ? (unsigned)y1 * (unsigned)w + (unsigned)x1 : 0u;
    const unsigned q01 = g.inb ? q + 1u : 0u, q10 = g.inb ? q + (unsigned)w : 0u, q11 = g.inb ? q + (unsigned)w + 1u : 0u;
    // float4 planes: one aligned 16-byte load per tap; scalar plane: the two x-neighbours of a row
    // in one 8-byte load (4-byte aligned: legal for global memory on gfx9+)
    {
        const char *b4 = reinterpret_cast<const char *>(R1);
#ifdef OFARN_EXP_HALFTAPS   // wrong-result experiment, kept out of this file: experiments/gather_halftaps.inc
#define OFARN_EXP_PART 1
#include "experiments/gather_halftaps.inc"
#else
        const float4 a00 = *reinterpret_cast<const float4 *>(b4 + q * 16u), a01 = *reinterpret_cast<const float4 *>(b4 + q01 * 16u);
        const float4 a10 = *reinterpret_cast<const float4 *>(b4 + q10 * 16u), a11 = *reinterpret_cast<const float4 *>(b4 + q11 * 16u);
#endif
        g.t00[0] = a00.x; g.t00[1] = a00.y; g.t00[2] = a00.z; g.t00[3] = a00.w;
        g.t01[0] = a01.x; g.t01[1] = a01.y; g.t01[2] = a01.z; g.t01[3] = a01.w;
        g.t10[0] = a10.x; g.t10[1] = a10.y; g.t10[2] = a10.z; g.t10[3] = a10.w;
        g.t11[0] = a11.x; g.t11[1] = a11.y; g.t11[2] = a11.z; g.t11[3] = a11.w;
        struct __attribute__((packed, aligned(4))) F2 { float a, b; };
        const char *b1 = reinterpret_cast<const char *>(R1 + 4 * npx);
#ifdef OFARN_EXP_HALFTAPS
#define OFARN_EXP_PART 2
#include "experiments/gather_halftaps.inc"
#else
        const F2 s0 = *reinterpret_cast<const F2 *>(b1 + q * 4u), s1 = *reinterpret_cast<const F2 *>(b1 + q10 * 4u);
        g.t00[4] = s0.a; g.t01[4] = s0.b; g.t10[4] = s1.a; g.t11[4] = s1.b;
#endif
    }
}

// Branch-free: a divergent branch here makes the compiler drain every outstanding load at the
// join (s_waitcnt vmcnt(0)), which would also wait for the NEXT rows' prefetched gathers.  The
// selects and the multiplication by an interior scale of exactly 1.0f leave every bit unchanged.
__device__ __forceinline__ float border_factor(int i) { return i < 2 ? 0.14f : 0.4472f; }   // BORDER table
// the four factors of FarnebackUpdateMatrices' border scale, in its order: (x lo)*(x hi)*(y lo)*(y hi)
__device__ __forceinline__ float border_x(int x, int w)
{
    return (x < kBorder ? border_factor(x) : 1.f) * (x >= w - kBorder ? border_factor(w - x - 1) : 1.f);
}
__device__ __forceinline__ float border_ylo(int y) { return y < kBorder ? border_factor(y) : 1.f; }
__device__ __forceinline__ float border_yhi(int y, int h) { return y >= h - kBorder ? border_factor(h - y - 1) : 1.f; }

// bx = border_x(x, w), ax = border_applies(x, w) (per-thread constants of a marching kernel); y is
// uniform across the block.  optflowgf.cpp only scales when its unsigned range test fires, which for
// images narrower than 2*BORDER is NOT the same as "within BORDER of an edge" -- kept literally.
__device__ __forceinline__ bool border_applies(int v, int n)
{
    return (unsigned)(v - kBorder) >= (unsigned)(n - kBorder * 2);
}

__device__ __forceinline__ void matrices_finish(const GatherRaw &g, float bx, bool ax, int h, int y, float out[5])
{
    const float dx = g.dx, dy = g.dy, fx = g.fx, fy = g.fy;
    const bool inb = g.inb != 0;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float r2 = a00 * g.t00[0] + a01 * g.t01[0] + a10 * g.t10[0] + a11 * g.t11[0];
    float r3 = a00 * g.t00[1] + a01 * g.t01[1] + a10 * g.t10[1] + a11 * g.t11[1];
    float r4 = a00 * g.t00[2] + a01 * g.t01[2] + a10 * g.t10[2] + a11 * g.t11[2];
    float r5 = a00 * g.t00[3] + a01 * g.t01[3] + a10 * g.t10[3] + a11 * g.t11[3];
    float r6 = a00 * g.t00[4] + a01 * g.t01[4] + a10 * g.t10[4] + a11 * g.t11[4];
    r4 = (g.r0[2] + r4) * 0.5f;
    r5 = (g.r0[3] + r5) * 0.5f;
    r6 = (g.r0[4] + r6) * 0.25f;
    r2 = inb ? r2 : 0.f;
    r3 = inb ? r3 : 0.f;
    r4 = inb ? r4 : g.r0[2];
    r5 = inb ? r5 : g.r0[3];
    r6 = inb ? r6 : g.r0[4] * 0.5f;
    r2 = (g.r0[0] - r2) * 0.5f;
    r3 = (g.r0[1] - r3) * 0.5f;
    r2 = r2 + (r4 * dy + r6 * dx);
    r3 = r3 + (r6 * dy + r5 * dx);
    const float scale = (ax || border_applies(y, h)) ? (bx * border_ylo(y)) * border_yhi(y, h) : 1.f;
    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    out[0] = r4 * r4 + r6 * r6;
    out[1] = (r4 + r5) * r6;
    out[2] = r5 * r5 + r6 * r6;
    out[3] = r4 * r2 + r6 * r3;
    out[4] = r6 * r2 + r5 * r3;
}

// Workgroup barrier that waits for LDS traffic only: __syncthreads() also drains vmcnt, which would
// stall on the global loads deliberately left in flight for the next row.
__device__ __forceinline__ void barrier_lds_only()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace ofarn
