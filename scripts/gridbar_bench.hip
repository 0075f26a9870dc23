// gridbar_bench.hip -- cost of a software grid barrier (agent-scope release/acquire) in a persistent
// kernel on gfx950, against the cost of a kernel boundary inside a hipGraph.  Decides whether the ICP
// iteration chain (6 dependent kernels, ~6 us fixed cost each) should become one persistent kernel.
//   hipcc -O3 --offload-arch=gfx950 scripts/gridbar_bench.hip -o gpurun_out/gridbar_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar { unsigned count; unsigned gen; unsigned fail; unsigned pad; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nwg, unsigned& gen)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE); // agent scope by default for HIP's __threadfence equivalent
        __threadfence();
        const unsigned old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nwg - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { ok = false; __hip_atomic_store(&b->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __threadfence();
    }
    ++gen;
    __syncthreads();
    return ok;
}

// each phase: every WG writes a slice, after the barrier reads a slice written by another WG (other XCD)
__global__ __launch_bounds__(256) void persistent(Bar* b, float* buf, int per_wg, int phases, unsigned* bad)
{
    unsigned gen = 0;
    const unsigned nwg = gridDim.x;
    unsigned errors = 0;
    for (int ph = 0; ph < phases; ++ph) {
        float* mine = buf + (size_t)blockIdx.x * per_wg;
        for (int i = threadIdx.x; i < per_wg; i += 256) mine[i] = (float)(ph * 7 + i);
        if (!grid_barrier(b, nwg, gen)) return;
        const float* other = buf + (size_t)((blockIdx.x + 1 + 8 * (ph % 3)) % nwg) * per_wg;
        for (int i = threadIdx.x; i < per_wg; i += 256) if (other[i] != (float)(ph * 7 + i)) ++errors;
        if (!grid_barrier(b, nwg, gen)) return;
    }
    if (errors) atomicAdd(bad, errors);
}

__global__ __launch_bounds__(256) void phase_write(float* buf, int per_wg, int ph)
{
    float* mine = buf + (size_t)blockIdx.x * per_wg;
    for (int i = threadIdx.x; i < per_wg; i += 256) mine[i] = (float)(ph * 7 + i);
}
__global__ __launch_bounds__(256) void phase_read(const float* buf, int per_wg, int ph, unsigned* bad)
{
    const float* other = buf + (size_t)((blockIdx.x + 1 + 8 * (ph % 3)) % gridDim.x) * per_wg;
    unsigned errors = 0;
    for (int i = threadIdx.x; i < per_wg; i += 256) if (other[i] != (float)(ph * 7 + i)) ++errors;
    if (errors) atomicAdd(bad, errors);
}

int main(int argc, char** argv)
{
    const int phases = 200;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent, 256, 0));
    printf("CUs %d, occupancy %d WG/CU, cooperativeLaunch %d\n", prop.multiProcessorCount, occ, prop.cooperativeLaunch);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int per_cu : {1, 2, 4}) {
        for (int per_wg : {256, 4096}) {
            const int nwg = prop.multiProcessorCount * per_cu;
            if (per_cu > occ) continue;
            Bar* b; float* buf; unsigned* bad;
            CK(hipMalloc((void**)&b, sizeof(Bar))); CK(hipMemset(b, 0, sizeof(Bar)));
            CK(hipMalloc((void**)&buf, (size_t)nwg * per_wg * sizeof(float)));
            CK(hipMalloc((void**)&bad, 4)); CK(hipMemset(bad, 0, 4));
            int ph = phases;
            void* args[] = {&b, &buf, (void*)&per_wg, &ph, &bad};
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(b, 0, sizeof(Bar), st));
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel((const void*)persistent, dim3(nwg), dim3(256), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            Bar hb; unsigned hbad; CK(hipMemcpy(&hb, b, sizeof hb, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            printf("persistent nwg %4d per_wg %5d: %.2f us per barrier (write+read phases incl.), fail %u, stale reads %u\n", nwg, per_wg,
                   ms * 1e3 / (2 * phases), hb.fail, hbad);
            // same work as a graph of kernels
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipMemset(bad, 0, 4));
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int p = 0; p < phases; ++p) {
                hipLaunchKernelGGL(phase_write, dim3(nwg), dim3(256), 0, st, buf, per_wg, p);
                hipLaunchKernelGGL(phase_read, dim3(nwg), dim3(256), 0, st, buf, per_wg, p, bad);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            printf("graph      nwg %4d per_wg %5d: %.2f us per kernel, stale reads %u\n", nwg, per_wg, ms * 1e3 / (2 * phases), hbad);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            CK(hipFree(b)); CK(hipFree(buf)); CK(hipFree(bad));
        }
    }
    return 0;
}
