// Micro-benchmark: histogram atomics on a few hundred hot bins -- device scope (memory side on a multi-XCD part) against
// workgroup-scope atomics (executed in the XCD's own L2) on a copy private to the XCC the wave runs on (HW_REG_XCC_ID).
// hipcc --offload-arch=gfx950 -O3 atomic_bench.hip -o atomic_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15u; }
__device__ __forceinline__ unsigned fidx(unsigned b) { return ((b & 255u) << 8) | (b >> 8); }
__device__ __forceinline__ unsigned bin_of(unsigned i, unsigned nbins)
{
    unsigned h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // triangular-ish concentration over nbins around bin 30000
    const unsigned a = h % nbins, b = (h >> 16) % nbins;
    return 30000u + (a + b) / 2u;
}

template <int MODE> // 0: device scope, one copy; 1: device scope, copy = blockIdx % 8; 2: workgroup scope, copy = XCC id; 3: agent scope explicit, copy = XCC id
__global__ __launch_bounds__(256) void hist_kernel(unsigned* __restrict__ h, unsigned n, unsigned nbins, unsigned* __restrict__ xcc_seen)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const unsigned x = xcc_id();
    if (xcc_seen && threadIdx.x == 0) atomicAdd(&xcc_seen[(blockIdx.x & 7) * 16 + x], 1u);
    if (i >= n) return;
    const unsigned b = fidx(bin_of(i, nbins));
    if (MODE == 0) atomicAdd(&h[b], 1u);
    else if (MODE == 1) atomicAdd(&h[(blockIdx.x & 7) * 65536 + b], 1u);
    else if (MODE == 2) __hip_atomic_fetch_add(&h[x * 65536 + b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&h[x * 65536 + b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void sum_kernel(const unsigned* __restrict__ h, unsigned long long* out)
{
    unsigned long long s = 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < 16u * 65536u; i += gridDim.x * 256) s += h[i];
    atomicAdd(out, s);
}

template <int MODE>
int run(const char* name, unsigned n, unsigned nbins, unsigned* d_h, unsigned long long* d_sum, unsigned* d_x)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (n + 255) / 256, reps = 50;
    float best = 1e9f, tot = 0;
    for (int r = 0; r < reps + 3; ++r) {
        CK(hipMemsetAsync(d_h, 0, 16u * 65536u * 4u));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(hist_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_h, n, nbins, r == 0 ? d_x : nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { tot += ms; best = ms < best ? ms : best; }
    }
    CK(hipMemset(d_sum, 0, 8));
    hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, 0, d_h, d_sum);
    unsigned long long s; CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
    printf("%-34s n %7u bins %5u : mean %7.2f us  best %7.2f us  sum %llu %s\n", name, n, nbins, tot / reps * 1e3f, best * 1e3f, s, s == n ? "OK" : "WRONG");
    return 0;
}

int main()
{
    unsigned* d_h; unsigned long long* d_sum; unsigned* d_x;
    CK(hipMalloc(&d_h, 16u * 65536u * 4u)); CK(hipMalloc(&d_sum, 8)); CK(hipMalloc(&d_x, 8 * 16 * 4)); CK(hipMemset(d_x, 0, 8 * 16 * 4));
    for (unsigned n : {100000u, 600000u})
        for (unsigned nbins : {64u, 500u, 4000u}) {
            if (run<0>("device scope, 1 copy", n, nbins, d_h, d_sum, d_x)) return 1;
            if (run<1>("device scope, copy = block % 8", n, nbins, d_h, d_sum, d_x)) return 1;
            if (run<2>("workgroup scope, copy = XCC id", n, nbins, d_h, d_sum, d_x)) return 1;
            if (run<3>("agent scope, copy = XCC id", n, nbins, d_h, d_sum, d_x)) return 1;
        }
    std::vector<unsigned> x(8 * 16); CK(hipMemcpy(x.data(), d_x, x.size() * 4, hipMemcpyDeviceToHost));
    printf("workgroups by (blockIdx %% 8) -> XCC id counts (first launches):\n");
    for (int b = 0; b < 8; ++b) { printf("  b%%8 = %d:", b); for (int i = 0; i < 16; ++i) if (x[b * 16 + i]) printf(" xcc%d x%u", i, x[b * 16 + i]); printf("\n"); }
    return 0;
}
