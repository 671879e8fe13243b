// Microbenchmark behind DESIGN.md 5 ("how close to the memory system"): the marching access pattern of the fused kernels
// (one thread per column, row by row, 16 B per element, 256 frames) in three modes -- read only, read + write, read + a
// four-tap gather of the same plane -- and two layouts, row-major [H][W] and column-tiled [W/256][H][256].
//   hipcc --offload-arch=gfx950 -O3 -o march_layout march_layout.hip && ./march_layout
// Measured on MI355X: read 6.2 TB/s in either layout, read + write 5.0 TB/s, read + gather 4.0 TB/s of unique bytes
// (the four mostly cache-hitting tap loads cost L1 / texture-addresser throughput).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_march(const float4* __restrict__ src, float4* __restrict__ dst, int W, int H, int strip, int tiled, int write)
{
    const int tid = threadIdx.x;
    const int cb = blockIdx.x;              // column block (256 wide)
    const int y0 = blockIdx.y * strip, y1 = min(y0 + strip, H);
    const size_t frame = (size_t)blockIdx.z * W * H;
    float4 acc = make_float4(0,0,0,0);
    for (int y = y0; y < y1; y++) {
        const size_t idx = tiled ? (size_t)cb * H * 256 + (size_t)y * 256 + tid : (size_t)y * W + cb * 256 + tid;
        float4 v = src[frame + idx];
        if (write == 2) {   // gather mode: four bilinear taps around a displaced position (row-major only), as flow_iter's R1 reads
            const size_t q = frame + (size_t)min(y + 2, H - 2) * W + min(cb * 256 + tid + 3, W - 2);
            const float4 t0 = src[q], t1 = src[q + 1], t2 = src[q + W], t3 = src[q + W + 1];
            v.x += t0.x + t1.x + t2.x + t3.x; v.y += t0.y + t1.y + t2.y + t3.y;
            v.z += t0.z + t1.z + t2.z + t3.z; v.w += t0.w + t1.w + t2.w + t3.w;
        }
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        if (write == 1) dst[frame + idx] = acc;
    }
    if (!write && acc.x == 123.456f) dst[0] = acc;
}
int main(int argc, char** argv)
{
    const int W = 1792, H = 1080, F = 256;  // 7 column blocks
    const size_t n = (size_t)W * H * F;
    float4 *a, *b;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int write = 0; write < 3; write++)
    for (int tiled = 0; tiled < (write == 2 ? 1 : 2); tiled++)
    for (int strips : {1, 3, 9}) {
        const int strip = (H + strips - 1) / strips;
        dim3 grid(W / 256, strips, F);
        hipLaunchKernelGGL(k_march, grid, dim3(256), 0, 0, a, b, W, H, strip, tiled, write);
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_march, grid, dim3(256), 0, 0, a, b, W, H, strip, tiled, write);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("mode=%d (0 read, 1 read+write, 2 read + 4-tap gather of the same plane) tiled=%d strips=%d: %.3f ms  %.0f GB/s of unique bytes\n", write, tiled, strips, ms, n * 16.0 * (write == 1 ? 2 : 1) / ms / 1e6);
    }
    return 0;
}
