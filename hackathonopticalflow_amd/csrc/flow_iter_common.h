// flow_iter_common.h -- pieces shared by the fused iteration kernels (kernels_fast.hip: box window, kernels_gauss.hip:
// Gaussian window): block geometry, the XCD-aware block remap, the on-the-fly upsample arguments.
#pragma once
#include "farneback_device.h"

#include <cstdlib>

namespace ofarn {

constexpr int FI_THREADS = 256;

// Output columns of a marching block with `halo` halo lanes per side: the largest multiple of 16 that the 256 threads
// leave room for.  A block's row segment is then a whole number of 128-byte lines in every plane it writes (16 columns x
// 8-byte flow, 8 columns x 16-byte R): with the natural width 256 - 2 * halo (242, 246) every block edge is a partially
// written line that a neighbouring block on another CU completes at another time, and the marching stores of the
// polynomial expansion reach 3.4 instead of 4.7 TB/s (tools/microbench/hbm_rw.hip).
constexpr int march_out_width(int halo) { return (FI_THREADS - 2 * halo) & ~15; }

// XCD-aware block remap (speed only): the dispatcher deals consecutive workgroup ids round-robin over
// the 8 XCDs, each with its own L2.  Regrouping ids so that ids congruent mod 8 become a contiguous
// range puts neighbouring column strips -- which read each other's halo columns -- behind the same L2.
__device__ __forceinline__ void xcd_remap(unsigned &bx, unsigned &by, unsigned &bz)
{
    const unsigned nx = gridDim.x, ny = gridDim.y, nb = nx * ny * gridDim.z;
    unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    if ((nb & 7u) == 0) lin = (lin & 7u) * (nb >> 3) + (lin >> 3);
    bx = lin % nx;
    const unsigned q = lin / nx;
    by = q % ny;
    bz = q / ny;
}

struct UpsampleArgs {
    const float2 *coarse;   // [P][ch][cw]
    int cw, ch;
    const int *xofs;
    const float *xa;
    double yscale;          // 1 / ((double)h / ch): resize.cpp's `scale` of the vertical axis
    float mul;
    unsigned long long *dbg;   // OFARN_STAMPS diagnostic build only: per-segment cycle sums
};

// resize(INTER_LINEAR) source coordinate of destination index d (resize.cpp, the table the host builds in resize_tables()):
// the same double and float operations in the same order, so the result equals the table entry bit for bit.  Used for the
// ROW tables of the on-the-fly upsample: a table lookup per row is a scalar load whose latency sits in the row's critical
// path (address -> four tap loads), the arithmetic is six VALU instructions on a uniform value.
__device__ __forceinline__ void resize_coord(int d, double scale, int ssize, int &s, float &f)
{
    f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
}


// Bytes written by one launch beyond which its stores carry the non-temporal hint (see store_r_nt): 4 x the 256 MB Infinity Cache.
constexpr size_t kNtStoreBytes = (size_t)1 << 30;
// OFARN_NT=0 switches the hint off, OFARN_NT=1 forces it on for every launch (A/B measurements on one box)
inline int nt_hint(size_t bytes)
{
    static const int mode = [] { const char *e = getenv("OFARN_NT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    return mode < 0 ? (bytes > kNtStoreBytes) : mode;
}

// Strip height in `unit`-row steps for a marching kernel (kernels_fast.hip).
int best_strip_units(int nunits, int unit, int warm, int blocks_per_strip_row, int blocks_per_cu);
int march_cu_count();   // CUs of the current device

}  // namespace ofarn
