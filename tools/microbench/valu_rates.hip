// Issue cost of the VALU instructions the polynomial expansion is made of, on gfx950: cycles per wave-instruction per SIMD
// with W waves per SIMD and independent chains (s_memtime around an unrolled loop, one number per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define LOOPS 200
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed)
{
    float f[8];
    double d[8];
    for (int i = 0; i < 8; i++) { f[i] = seed + i + threadIdx.x * 1e-3f; d[i] = f[i]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < LOOPS; l++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (OP == 6) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += f[i] + (float)d[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345.678f) out[0] = 0;
}
template <int OP> void run(const char *name, int blocks_per_cu)
{
    unsigned long long *d;
    const int nb = 256 * blocks_per_cu;
    hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(256), 0, 0, d, 1.5f);
    hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(256), 0, 0, d, 1.5f);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, d, nb * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb; i++) s += h[i];
    // a block = 4 waves = one wave per SIMD; with blocks_per_cu blocks resident, each SIMD interleaves that many waves
    const double cyc_per_instr_per_simd = s / nb / (double)(REP * LOOPS) / blocks_per_cu;
    printf("%-18s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD\n", name, blocks_per_cu, cyc_per_instr_per_simd);
    hipFree(d);
}
int main()
{
    for (int w : {1, 4}) {
        if (w == 1) { run<0>("v_mul_f32", 1); run<1>("v_add_f64", 1); run<2>("v_fma_f64", 1); run<5>("v_mul_f64", 1); run<3>("v_cvt_f64_f32", 1); run<4>("v_cvt_f32_f64", 1); run<6>("v_cvt_f32_ubyte0", 1); run<7>("v_mov_b32", 1); }
        else { run<0>("v_mul_f32", 4); run<1>("v_add_f64", 4); run<2>("v_fma_f64", 4); run<5>("v_mul_f64", 4); run<3>("v_cvt_f64_f32", 4); run<4>("v_cvt_f32_f64", 4); run<6>("v_cvt_f32_ubyte0", 4); run<7>("v_mov_b32", 4); }
    }
    return 0;
}
