// pitch.hip -- does the ROW PITCH of the R planes matter?  The marching kernels write (polynomial expansion) and read (fused iteration)
// rows of 1920 pixels: 30720 B apart in the 16-byte plane, 7680 B in the 4-byte plane.  This runs the two patterns -- one thread per
// column marching down a strip, 240 columns per block, 3 strips per frame, 400 frames -- for a list of pitches (in pixels).
//   hipcc --offload-arch=gfx950 -O3 -o pitch pitch.hip && ./pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_march_write(float4 *p4, float *p1, int W, int H, int pitch, int strip, float v)
{
    const int x = blockIdx.x * 240 + threadIdx.x - 8;
    const int y0 = blockIdx.y * strip, y1 = min(y0 + strip, H);
    const size_t f = (size_t)blockIdx.z * pitch * H;
    if (threadIdx.x < 8 || threadIdx.x >= 248 || x >= W) return;
    for (int y = y0; y < y1; y++) {
        const size_t o = f + (size_t)y * pitch + x;
        p4[o] = make_float4(v, v, v, v);
        p1[o] = v;
    }
}

// the fused iteration's R0 stream: 16 + 4 B per pixel read, 8 B written (to a third plane)
__global__ __launch_bounds__(256) void k_march_read(const float4 *p4, const float *p1, float2 *out, int W, int H, int pitch, int strip)
{
    const int x = blockIdx.x * 240 + threadIdx.x - 8;
    const int y0 = blockIdx.y * strip, y1 = min(y0 + strip, H);
    const size_t f = (size_t)blockIdx.z * pitch * H;
    if (threadIdx.x < 8 || threadIdx.x >= 248 || x >= W) return;
    float acc = 0;
    for (int y = y0; y < y1; y++) {
        const size_t o = f + (size_t)y * pitch + x;
        const float4 a = p4[o];
        const float b = p1[o];
        acc += a.x + a.y + a.z + a.w + b;
        out[o] = make_float2(acc, b);
    }
}

int main()
{
    const int W = 1920, H = 1080, F = 400, strips = 3;
    const size_t maxpitch = 2304;
    float4 *p4; float *p1; float2 *o2;
    if (hipMalloc(&p4, maxpitch * H * F * 16) != hipSuccess || hipMalloc(&p1, maxpitch * H * F * 4) != hipSuccess ||
        hipMalloc(&o2, maxpitch * H * F * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(p4, 0, maxpitch * H * F * 16); hipMemset(p1, 0, maxpitch * H * F * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sh = (H + strips - 1) / strips;
    const dim3 grid((W + 239) / 240, strips, F);
    const int pitches[] = {1920, 1928, 1936, 1952, 1984, 2016, 2048, 2064, 2080, 2112, 2176, 2304, 1920};
    for (int pitch : pitches) {
        float msw, msr;
        hipLaunchKernelGGL(k_march_write, grid, dim3(256), 0, 0, p4, p1, W, H, pitch, sh, 1.f);
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_march_write, grid, dim3(256), 0, 0, p4, p1, W, H, pitch, sh, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msw, e0, e1); msw /= 3;
        hipLaunchKernelGGL(k_march_read, grid, dim3(256), 0, 0, p4, p1, o2, W, H, pitch, sh);
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_march_read, grid, dim3(256), 0, 0, p4, p1, o2, W, H, pitch, sh);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msr, e0, e1); msr /= 3;
        const double px = (double)W * H * F;
        printf("pitch %4d px (%6d B rows in the 16-B plane): march write 20 B/px %6.3f ms = %5.2f TB/s;  march read 20 + write 8 B/px %6.3f ms = %5.2f TB/s\n",
               pitch, pitch * 16, msw, px * 20 / msw / 1e9, msr, px * 28 / msr / 1e9);
    }
    return 0;
}
