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

// FarnebackUpdateMatrices for one pixel (optflowgf.cpp), all float32, no FMA contraction.
// R0, R1: channel-planar [5][h][w] of the pair's two frames; (dx, dy) the current flow at (x, y).
__device__ __forceinline__ void update_matrices_px(const float *__restrict__ R0, const float *__restrict__ R1,
                                                    size_t npx, int w, int h, int x, int y, float dx,
                                                    float dy, float out[5])
{
    float fx = (float)x + dx, fy = (float)y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= (float)x1; fy -= (float)y1;
    const size_t o = (size_t)y * w + x;
    float r2, r3, r4, r5, r6;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const size_t q = (size_t)y1 * w + x1;
        const float *p = R1 + q;
        r2 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r3 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r4 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r5 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r6 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1];
        r4 = (R0[2 * npx + o] + r4) * 0.5f;
        r5 = (R0[3 * npx + o] + r5) * 0.5f;
        r6 = (R0[4 * npx + o] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = R0[2 * npx + o];
        r5 = R0[3 * npx + o];
        r6 = R0[4 * npx + o] * 0.5f;
    }
    r2 = (R0[o] - r2) * 0.5f;
    r3 = (R0[npx + o] - r3) * 0.5f;
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

}  // namespace ofarn
