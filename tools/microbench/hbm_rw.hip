// What HBM3E on MI355X sustains for the traffic mixes of the two dominant kernels, measured with plain streaming kernels
// on buffers far beyond the 256 MB Infinity Cache: read only, write only, copy (1 : 1), the 6 : 1 read : write mix of the
// fused iteration kernel, and the marching store pattern of the polynomial expansion.  16 B per lane per access.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_rw hbm_rw.hip && ./hbm_rw [GiB per buffer, default 16]
// Round 3 adds the copy shapes MI355X_MICROARCH.md's 6.29 TB/s figure is quoted for (several loads in flight per lane,
// grids that are a multiple of 256 CUs x 8 waves, block-contiguous ranges, non-temporal variants) next to the plain
// grid-stride loop of round 2, so that the "copy" line can be reconciled with the guide on the same box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_write(f4 *dst, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4 val = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = val;
}
__global__ __launch_bounds__(256) void k_read(const f4 *src, size_t n, f4 *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a += src[i];
    if (a.x == 1234.5f) sink[0] = a;
}
// U independent loads in flight per lane before the first use (the guide's ">= 2 loads in flight per lane")
template <int U>
__global__ __launch_bounds__(256) void k_read_u(const f4 *src, size_t n, f4 *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 a = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) a += t[u];
    }
    for (; i < n; i += stride) a += src[i];
    if (a.x == 1234.5f) sink[0] = a;
}
// R reads of 16 B per 1 write of 16 B (R = 1: copy), one access in flight per lane and plane (round-2 form)
template <int R>
__global__ __launch_bounds__(256) void k_mix(const f4 *src, f4 *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        f4 a = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; r++) a += src[(size_t)r * n + i];   // R source planes of n elements each
        dst[i] = a;
    }
}
// copy with U loads in flight per lane; NT: non-temporal loads and stores
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_u(const f4 *src, f4 *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) __builtin_nontemporal_store(t[u], dst + i + u * stride);
            else dst[i + u * stride] = t[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// copy where each block owns one contiguous range (the shape of a tiled kernel), U loads in flight
template <int U>
__global__ __launch_bounds__(256) void k_copy_blk(const f4 *src, f4 *dst, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    size_t i = b0 + threadIdx.x;
    for (; i + (U - 1) * 256 < b1; i += U * 256) {
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) dst[i + u * 256] = t[u];
    }
    for (; i < b1; i += 256) dst[i] = src[i];
}
// the polynomial expansion's write pattern: one thread per column marching down a strip, 16 B + 4 B per pixel into two planes.
// MODE 0: row-major planes (what the kernels use).  1: only the 16-B plane.  2: column-tiled planes [W/246][H][246] (every
// block writes one contiguous slab).  3: row-major, non-temporal stores.  4: one 20-B record per pixel, five dword stores
// (interleaved layout).
// OUTW: output columns per block (246 = 256 threads minus the 2 x 5 halo lanes of the round-1 kernel; 240 and 224 make a block's
// row segment a whole number of 128-byte lines in the 16-B plane, resp. in both planes)
// Round 4: 192 and 256 columns per block -- the two widths below / above 240 for which a block's row segment
// is a whole number of 128-byte lines in BOTH planes (192: 3072 + 768 B, 256: 4096 + 1024 B; 240 ends the 4-B plane on a half line:
// 960 B = 7.5 lines).  THREADS: block size (272 = 256 writers + 2 x 8 halo lanes); HALO: idle lanes in front of the writers.
template <int MODE, int OUTW = 246, int THREADS = 256, int HALO = 5>
__global__ __launch_bounds__(THREADS) void k_march_write(float4 *p4, float *p1, int W, int H, int strip, float v)
{
    const int x = blockIdx.x * OUTW + threadIdx.x - HALO;
    const int y0 = blockIdx.y * strip, y1 = min(y0 + strip, H);
    const size_t f = (size_t)blockIdx.z * W * H;
    if ((int)threadIdx.x < HALO || (int)threadIdx.x >= HALO + OUTW || x >= W || x < 0) return;
    const int lx = threadIdx.x - HALO;
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
int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 16;
    const size_t bytes = gib << 30, n = bytes / 16;
    f4 *a, *b;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# %s (%s), %d CUs, buffers of %zu GiB\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, gib);
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes + (4ull << 30)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, double gb, auto launch) {
        launch();
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-66s %8.3f ms  %7.0f GB/s\n", name, ms, gb / (ms * 1e-3));
        fflush(stdout);
    };
    const int nb = 256 * 8;
    char name[160];
    time("write only, 16 B per lane, grid-stride, 2048 blocks", bytes / 1e9, [&] { hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, 0, a, n, 1.f); });
    time("read only, 1 load in flight, 2048 blocks", bytes / 1e9, [&] { hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, 0, a, n, b); });
    time("read only, 4 loads in flight, 2048 blocks", bytes / 1e9, [&] { hipLaunchKernelGGL(k_read_u<4>, dim3(nb), dim3(256), 0, 0, a, n, b); });
    time("copy (1 read : 1 write), 1 in flight, 2048 blocks; bytes = 2 x", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_mix<1>, dim3(nb), dim3(256), 0, 0, a, b, n); });
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        snprintf(name, sizeof name, "copy, 2 loads in flight per lane, %d blocks", blocks);
        time(name, 2 * bytes / 1e9, [&] { hipLaunchKernelGGL((k_copy_u<2, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
        snprintf(name, sizeof name, "copy, 4 loads in flight per lane, %d blocks", blocks);
        time(name, 2 * bytes / 1e9, [&] { hipLaunchKernelGGL((k_copy_u<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    time("copy, 8 loads in flight per lane, 2048 blocks", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL((k_copy_u<8, false>), dim3(nb), dim3(256), 0, 0, a, b, n); });
    time("copy, 4 in flight, non-temporal loads + stores, 2048 blocks", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL((k_copy_u<4, true>), dim3(nb), dim3(256), 0, 0, a, b, n); });
    time("copy, block-contiguous ranges, 4 in flight, 2048 blocks", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy_blk<4>, dim3(nb), dim3(256), 0, 0, a, b, n); });
    time("copy, block-contiguous ranges, 4 in flight, 65536 blocks", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy_blk<4>, dim3(65536), dim3(256), 0, 0, a, b, n); });
    {
        // the guide's buffer size: 4 GiB (still 16 x the Infinity Cache)
        const size_t n4 = (4ull << 30) / 16;
        if (n4 <= n) time("copy, 4 in flight, 2048 blocks, 4 GiB buffers", 2 * (4ull << 30) / 1e9, [&] { hipLaunchKernelGGL((k_copy_u<4, false>), dim3(nb), dim3(256), 0, 0, a, b, n4); });
    }
    // 6 source planes of n/6 elements each (all inside a's n elements), one destination plane of n/6 elements
    time("6 reads : 1 write (fused iteration mix), 7/6 x, 2048 blocks", (bytes + bytes / 6) / 1e9, [&] { hipLaunchKernelGGL(k_mix<6>, dim3(nb), dim3(256), 0, 0, a, b, n / 6); });
    time("6 reads : 1 write, 8192 blocks", (bytes + bytes / 6) / 1e9, [&] { hipLaunchKernelGGL(k_mix<6>, dim3(8192), dim3(256), 0, 0, a, b, n / 6); });
    for (int strips : {3}) {
        const int W = 1920, H = 1080, F = 400;
        dim3 grid((W + 245) / 246, strips, F);
        const int sh = (H + strips - 1) / strips;
        float4 *b4 = reinterpret_cast<float4 *>(b);
        float *p1 = reinterpret_cast<float *>(b4 + (size_t)2000 * H * F);     // planes padded to 2000 columns: room for the tiled layouts
        const double gb20 = (double)W * H * F * 20 / 1e9, gb16 = (double)W * H * F * 16 / 1e9;
        snprintf(name, sizeof name, "march write 16+4 B row-major, 246 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<0>, grid, dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 240 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 240>), dim3((W + 239) / 240, strips, F), dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 224 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 224>), dim3((W + 223) / 224, strips, F), dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        // round 4: whole lines in both planes
        snprintf(name, sizeof name, "march write 16+4 B row-major, 192 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 192>), dim3((W + 191) / 192, strips, F), dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 256 columns per 272-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 256, 272, 8>), dim3((W + 255) / 256, strips, F), dim3(272), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 128 columns per 192-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 128, 192, 32>), dim3((W + 127) / 128, strips, F), dim3(192), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 320 columns per 384-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 320, 384, 5>), dim3((W + 319) / 320, strips, F), dim3(384), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 384 columns per 448-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 384, 448, 5>), dim3((W + 383) / 384, strips, F), dim3(448), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 96 columns per 128-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 96, 128, 5>), dim3((W + 95) / 96, strips, F), dim3(128), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 480 columns per 512-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 480, 512, 5>), dim3((W + 479) / 480, strips, F), dim3(512), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major, 960 columns per 1024-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<0, 960, 1024, 5>), dim3((W + 959) / 960, strips, F), dim3(1024), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major nt, 240 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<3, 240>), dim3((W + 239) / 240, strips, F), dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major nt, 192 columns per block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<3, 192>), dim3((W + 191) / 192, strips, F), dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major nt, 256 columns per 272-thread block, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL((k_march_write<3, 256, 272, 8>), dim3((W + 255) / 256, strips, F), dim3(272), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16 B plane only, %d strips", strips);
        time(name, gb16, [&] { hipLaunchKernelGGL(k_march_write<1>, grid, dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B column-tiled, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<2>, grid, dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 16+4 B row-major nt, 246 columns, %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<3>, grid, dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
        snprintf(name, sizeof name, "march write 20-B records (5 dwords), %d strips", strips);
        time(name, gb20, [&] { hipLaunchKernelGGL(k_march_write<4>, grid, dim3(256), 0, 0, b4, p1, W, H, sh, 1.f); });
    }
    return 0;
}
