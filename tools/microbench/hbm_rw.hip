// What HBM3E on MI355X sustains for the traffic mixes of the two dominant kernels, measured with plain streaming kernels
// on 16 GiB buffers (far beyond the 256 MB Infinity Cache): read only, write only, copy (1 : 1), and the 6 : 1 read : write
// mix of the fused iteration kernel.  Each thread moves 16 B per access, grid-stride, fully coalesced.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_rw hbm_rw.hip && ./hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_write(float4 *dst, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_read(const float4 *src, size_t n, float4 *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 t = src[i]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    if (a.x == 1234.5f) sink[0] = a;
}
// R reads of 16 B per 1 write of 16 B (R = 1: copy)
template <int R>
__global__ __launch_bounds__(256) void k_mix(const float4 *src, float4 *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; r++) {   // R source planes of n elements each: every index is < R * n
            const float4 t = src[(size_t)r * n + i];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        dst[i] = a;
    }
}
// the polynomial expansion's write pattern: one thread per column marching down a strip, 16 B + 4 B per pixel into two planes.
// MODE 0: row-major planes (what the kernels use).  1: only the 16-B plane.  2: column-tiled planes [W/246][H][246] (every
// block writes one contiguous slab).  3: row-major, non-temporal stores.  4: one 20-B record per pixel, five dword stores
// (interleaved layout).  5: tiled [strip][W/246][rows][246] so a block's whole output is one contiguous range.
// OUTW: output columns per block (246 = 256 threads minus the 2 x 5 halo lanes of the real kernel; 240 and 224 make a block's
// row segment a whole number of 128-byte lines in the 16-B plane, resp. in both planes)
template <int MODE, int OUTW = 246>
__global__ __launch_bounds__(256) void k_march_write(float4 *p4, float *p1, int W, int H, int strip, float v)
{
    const int x = blockIdx.x * OUTW + threadIdx.x - 5;
    const int y0 = blockIdx.y * strip, y1 = min(y0 + strip, H);
    const size_t f = (size_t)blockIdx.z * W * H;
    if (threadIdx.x < 5 || threadIdx.x >= 5 + OUTW || x >= W) return;
    const int lx = threadIdx.x - 5;
    for (int y = y0; y < y1; y++) {
        size_t o = f + (size_t)y * W + x;
        if (MODE == 2) o = f + ((size_t)blockIdx.x * H + y) * 246 + lx;
        const float4 val = make_float4(v, v, v, v);
        if (MODE == 3) {
            __builtin_nontemporal_store(val.x, &p4[o].x); __builtin_nontemporal_store(val.y, &p4[o].y);
            __builtin_nontemporal_store(val.z, &p4[o].z); __builtin_nontemporal_store(val.w, &p4[o].w);
            __builtin_nontemporal_store(v, &p1[o]);
        } else if (MODE == 4) {
            float *r = reinterpret_cast<float *>(p4) + o * 5;
            r[0] = v; r[1] = v; r[2] = v; r[3] = v; r[4] = v;
        } else {
            p4[o] = val;
            if (MODE != 1) p1[o] = v;
        }
    }
}
int main()
{
    const size_t bytes = 16ull << 30, n = bytes / 16;
    float4 *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes + (4ull << 30)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, double gb, auto launch) {
        launch();
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-52s %8.3f ms  %7.0f GB/s\n", name, ms, gb / (ms * 1e-3));
    };
    const int nb = 256 * 8;
    time("write only, 16 B per lane, grid-stride", bytes / 1e9, [&] { hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, 0, a, n, 1.f); });
    time("read only", bytes / 1e9, [&] { hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, 0, a, n, b); });
    time("copy (1 read : 1 write), bytes moved = 2 x", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_mix<1>, dim3(nb), dim3(256), 0, 0, a, b, n); });
    // 6 source planes of n/6 elements each (all inside a's n elements), one destination plane of n/6 elements
    time("6 reads : 1 write (fused iteration mix), 7/6 x", (bytes + bytes / 6) / 1e9, [&] { hipLaunchKernelGGL(k_mix<6>, dim3(nb), dim3(256), 0, 0, a, b, n / 6); });
    for (int strips : {3}) {
        const int W = 1920, H = 1080, F = 400;
        dim3 grid((W + 245) / 246, strips, F);
        const int sh = (H + strips - 1) / strips;
        float *p1 = reinterpret_cast<float *>(b + (size_t)2000 * H * F);     // planes padded to 2000 columns: room for the tiled layouts
        char name[128];
        const double gb20 = (double)W * H * F * 20 / 1e9, gb16 = (double)W * H * F * 16 / 1e9;
        snprintf(name, sizeof name, "march write 16+4 B row-major, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<0>, grid, dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 240 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 240>), dim3((W + 239) / 240, strips, F), dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 224 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 224>), dim3((W + 223) / 224, strips, F), dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16 B plane only, %d strips", strips);
        time(name, gb16, [&] { hipLaunchKernelGGL(k_march_write<1>, grid, dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B column-tiled, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<2>, grid, dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major nt, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<3>, grid, dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 20-B records (5 dwords), %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<4>, grid, dim3(256), 0, 0, b, p1, W, H, sh, 1.f); });
    }
    return 0;
}
