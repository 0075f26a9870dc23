// pmc_calib.hip -- known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM
// section: "calibrate on a known byte count in your own access pattern").  Every kernel touches a buffer larger than the
// 256 MiB Infinity Cache exactly once:
//   calib_load16   reads  n x 16 B (float4 per lane, coalesced)      -- the NN kernel's candidate / query loads
//   calib_gather16 reads  n x 16 B in groups of 8 consecutive float4 at scattered places -- the NN kernel's row runs
//   calib_store16  writes n x 16 B (float4 per lane, coalesced)      -- matched-point stores
//   calib_store4   writes n x 4 B  (one float per lane, coalesced)   -- d^2 / index stores
//   calib_atomic4  n atomicAdd(u32) on distinct words                 -- histogram updates
//   hipcc -O3 --offload-arch=gfx950 scripts/pmc_calib.hip -o scripts/pmc_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_load16(const float4* __restrict__ in, size_t n, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 v = in[i];
    if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = 1.f;
}
__global__ void calib_gather16(const float4* __restrict__ in, size_t n, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t grp = i >> 3, sub = i & 7;
    const size_t g2 = (grp * 2654435761ull) % (n >> 3); // a permutation-like scatter of the groups of 8
    const float4 v = in[g2 * 8 + sub];
    if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = 1.f;
}
__global__ void calib_store16(float4* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ void calib_store4(float* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)i;
}
__global__ void calib_atomic4(unsigned* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&out[i], 1u);
}

int main()
{
    const size_t n = 64ull << 20; // 64 Mi elements: 1 GiB of float4, 256 MiB of float
    float4* a; float* o; float* f; unsigned* u;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&o, 64)); CK(hipMalloc(&f, n * sizeof(float))); CK(hipMalloc(&u, n * sizeof(unsigned)));
    CK(hipMemset(a, 0, n * sizeof(float4))); CK(hipMemset(u, 0, n * sizeof(unsigned)));
    const unsigned g = (unsigned)((n + 255) / 256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_load16, dim3(g), dim3(256), 0, 0, a, n, o);
        hipLaunchKernelGGL(calib_gather16, dim3(g), dim3(256), 0, 0, a, n, o);
        hipLaunchKernelGGL(calib_store16, dim3(g), dim3(256), 0, 0, a, n);
        hipLaunchKernelGGL(calib_store4, dim3(g), dim3(256), 0, 0, f, n);
        hipLaunchKernelGGL(calib_atomic4, dim3(g), dim3(256), 0, 0, u, n);
        CK(hipDeviceSynchronize());
    }
    printf("n = %zu: load16 / gather16 read %zu bytes, store16 writes %zu, store4 writes %zu, atomic4 touches %zu\n", n, n * 16, n * 16, n * 4, n * 4);
    return 0;
}
