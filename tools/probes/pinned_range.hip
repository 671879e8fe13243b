// Probe (round 4): which runtime calls report the extent of a page-locked host allocation, given a pointer
// into its middle?  Decides how the zero-copy output path validates [h_flow, h_flow + bytes).
//   hipcc --offload-arch=gfx950 -o pinned_range pinned_range.hip && ./pinned_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static void probe(const char *what, char *p, size_t off)
{
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p + off);
    printf("%s +%zu: hipPointerGetAttributes -> %s", what, off, hipGetErrorName(e));
    if (e == hipSuccess) printf(" type=%d hostPointer=%p (query %p) devicePointer=%p", (int)at.type, at.hostPointer, (void *)(p + off), at.devicePointer);
    else (void)hipGetLastError();
    printf("\n");
    void *dp = nullptr;
    e = hipHostGetDevicePointer(&dp, p + off, 0);
    printf("   hipHostGetDevicePointer -> %s dp=%p\n", hipGetErrorName(e), dp);
    if (e != hipSuccess) { (void)hipGetLastError(); return; }
    hipDeviceptr_t base = nullptr; size_t size = 0;
    e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dp);
    printf("   hipMemGetAddressRange(dp) -> %s base=%p size=%zu (dp - base = %td)\n", hipGetErrorName(e), (void *)base, size,
           e == hipSuccess ? (char *)dp - (char *)base : (ptrdiff_t)0);
    if (e != hipSuccess) (void)hipGetLastError();
    e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)(p + off));
    printf("   hipMemGetAddressRange(host ptr) -> %s base=%p size=%zu\n", hipGetErrorName(e), (void *)base, size);
    if (e != hipSuccess) (void)hipGetLastError();
}
int main()
{
    char *a = nullptr;
    const size_t n = 1 << 20;
    if (hipHostMalloc((void **)&a, n, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    probe("hipHostMalloc 1 MiB", a, 0);
    probe("hipHostMalloc 1 MiB", a, n - 64);
    probe("hipHostMalloc 1 MiB", a, n);           // one past the end
    char *r = (char *)aligned_alloc(4096, n);
    if (hipHostRegister(r, n, hipHostRegisterDefault) == hipSuccess) {
        probe("hipHostRegister 1 MiB", r, 0);
        probe("hipHostRegister 1 MiB", r, n - 64);
        hipHostUnregister(r);
    } else printf("hipHostRegister failed\n");
    char *m = (char *)malloc(n);
    probe("malloc", m, 0);
    return 0;
}
