// pmc_calib.hip -- known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950:
// a coalesced 16-byte-per-lane stream, and scattered 4- / 8-byte gathers and scatters over a 1 GiB footprint (far
// beyond the 256 MiB Infinity Cache), the access shapes of the pivot kernels.  tools/pmc_calib.sh runs it under
// rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes) and writes counter bytes / useful bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x)
{ // SplitMix64 finalizer: a different pseudo-random element for every thread
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void calib_stream16(const float4 *a, float4 *b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <class T> __global__ void calib_gather(const T *a, T *out, size_t nelem, size_t nthreads)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nthreads) return;
    T acc = 0;
    for (int r = 0; r < 8; r++) acc += a[mix(i * 8 + r) % nelem];
    if (acc == (T)12345.678) out[i] = acc; // (keeps the loads alive; practically never true)
}
template <class T> __global__ void calib_scatter(T *a, size_t nelem, size_t nthreads)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nthreads) return;
    for (int r = 0; r < 8; r++) a[mix(i * 8 + r) % nelem] = (T)r;
}

int main()
{
    const size_t bytes = (size_t)1 << 30; // 1 GiB footprint
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    const size_t n16 = bytes / 16, nth = (size_t)1 << 24; // 16 Mi threads x 8 accesses each
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib_stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n16);
        hipLaunchKernelGGL(calib_gather<float>, dim3((unsigned)(nth / 256)), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4, nth);
        hipLaunchKernelGGL(calib_gather<double>, dim3((unsigned)(nth / 256)), dim3(256), 0, 0, (const double *)a, (double *)b, bytes / 8, nth);
        hipLaunchKernelGGL(calib_scatter<float>, dim3((unsigned)(nth / 256)), dim3(256), 0, 0, (float *)a, bytes / 4, nth);
        hipLaunchKernelGGL(calib_scatter<double>, dim3((unsigned)(nth / 256)), dim3(256), 0, 0, (double *)a, bytes / 8, nth);
        CK(hipDeviceSynchronize());
    }
    printf("useful bytes per launch: stream16 read %zu write %zu; gather4 %zu; gather8 %zu; scatter4 %zu; scatter8 %zu\n", bytes, bytes,
           nth * 8 * 4, nth * 8 * 8, nth * 8 * 4, nth * 8 * 8);
    return 0;
}
