// l2_survive.hip -- does an XCD's L2 keep a read-only slice from one kernel launch to the next?  
// A 16 MB buffer (the size of the level-0 map at 1 M points) is cut into 8 slices of 2 MB; workgroup b runs on XCD b % 8 and only
// ever touches slice b % 8 (the XCD-aware mapping of nn1_wg_kernel).  `warm` reads the XCD's slice with wide coalesced loads;
// `probe` has lane 0 of one-wave workgroups chase dependent loads through random lines of a slice and reports cycles per step.
//   same     : warm + chase in ONE launch (a private 512 KB region per workgroup): the L2-hit latency under this access pattern
//   next     : warm in launch A, chase in launch B (eager, back to back on one stream): survives the boundary?
//   graph    : the same two launches as nodes of one hipGraph
//   other    : launch B chases the slice ANOTHER XCD warmed: L2 miss served by the Infinity Cache
//   streamed : between A and B a launch streams 12 MB of other data through every L2 (plain / non-temporal loads)
//   cold     : launch B after 1 GiB of other traffic: HBM
// also: the duration of `warm` itself on a first and a repeated launch (bandwidth view of the same question).
//   hipcc -O3 --offload-arch=gfx950 scripts/r5/l2_survive.hip -o scripts/r5/l2_survive.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
// the shader clock, ordered against the loads around it: the compiler is free to move a plain clock64() across loads it does not depend on
// (the first version of this file read both clocks before the chase and reported 0 cycles)
__device__ __forceinline__ long long tick(unsigned dep)
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return (long long)t;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t SLICE_BYTES = 2u << 20;
constexpr size_t SLICE_U4 = SLICE_BYTES / 16, SLICE_W = SLICE_BYTES / 4;

__global__ __launch_bounds__(256) void warm(const uint4* __restrict__ buf, unsigned* __restrict__ sink)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, G = gridDim.x >> 3;
    const uint4* s = buf + (size_t)xcd * SLICE_U4;
    unsigned acc = 0;
    for (size_t i = (size_t)j * 256 + threadIdx.x; i < SLICE_U4; i += (size_t)G * 256) { const uint4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) sink[0] = 1;
}

// `warm` that also WRITES: 1.5 MB per XCD of stores to another buffer + one atomic per thread (what an NN launch leaves behind: matches, histogram)
__global__ __launch_bounds__(256) void warm_write(const uint4* __restrict__ buf, uint4* __restrict__ other, unsigned* __restrict__ hist, unsigned* __restrict__ sink)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, G = gridDim.x >> 3;
    const uint4* s = buf + (size_t)xcd * SLICE_U4;
    unsigned acc = 0;
    for (size_t i = (size_t)j * 256 + threadIdx.x; i < SLICE_U4; i += (size_t)G * 256) { const uint4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    constexpr size_t WU4 = (3u << 19) / 16; // 1.5 MB per XCD
    uint4* o = other + (size_t)xcd * WU4;
    for (size_t i = (size_t)j * 256 + threadIdx.x; i < WU4; i += (size_t)G * 256) o[i] = make_uint4((unsigned)i, acc, 2u, 3u);
    atomicAdd(&hist[(blockIdx.x * 256 + threadIdx.x) & 65535], 1u);
    if (acc == 0x12345u) sink[0] = 1;
}
// a one-workgroup launch (the solve of an iteration): does it shift the round-robin of the NEXT launch's workgroups over the XCDs?
__global__ void one_wg(unsigned* __restrict__ sink) { if (threadIdx.x == 999) sink[1] = 1; }
// which XCC does a workgroup run on?  (HW_REG_XCC_ID, low 4 bits)
__global__ void xcc_of(unsigned* __restrict__ out)
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v & 15u;
}

template <bool NT>
__global__ __launch_bounds__(256) void stream(const uint4* __restrict__ buf, size_t n_u4, unsigned* __restrict__ sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_u4; i += (size_t)gridDim.x * 256) {
        uint4 v;
        if (NT) { const unsigned* p = reinterpret_cast<const unsigned*>(buf + i);
                  v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1); v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3); }
        else v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = 1;
}

// one wave per workgroup, lane 0 chases; `shift` rotates the slice a workgroup looks at (0: its own XCD's)
__global__ __launch_bounds__(64) void probe(const unsigned* __restrict__ buf, int steps, int shift, unsigned long long* __restrict__ out)
{
    const int xcd = (blockIdx.x + shift) & 7;
    const unsigned* s = buf + (size_t)xcd * SLICE_W;
    if (threadIdx.x != 0) return;
    unsigned idx = (blockIdx.x * 2654435761u) % (unsigned)SLICE_W;
    const long long t0 = tick(idx);
    for (int k = 0; k < steps; ++k) { const unsigned v = s[idx]; idx = (idx * 1664525u + 1013904223u + v) % (unsigned)SLICE_W; }
    const long long t1 = tick(idx);
    out[blockIdx.x] = (unsigned long long)(t1 - t0) / (unsigned)steps + (idx == 0xffffffffu ? 1 : 0);
}

// warm a private 512 KB region with the whole wave, then lane 0 chases inside it: L2 hits (the region is 16 x the L1)
__global__ __launch_bounds__(64) void same(const unsigned* __restrict__ buf, int steps, unsigned long long* __restrict__ out, unsigned* __restrict__ sink)
{
    constexpr unsigned RW = (512u << 10) / 4;
    const int xcd = blockIdx.x & 7, j = (blockIdx.x >> 3) & 3;
    const unsigned* s = buf + (size_t)xcd * SLICE_W + (size_t)j * RW;
    unsigned acc = 0;
    for (unsigned i = threadIdx.x * 4; i < RW; i += 256) { const uint4 v = *reinterpret_cast<const uint4*>(s + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) sink[0] = 1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned idx = (blockIdx.x * 2654435761u) % RW;
    const long long t0 = tick(idx);
    for (int k = 0; k < steps; ++k) { const unsigned v = s[idx]; idx = (idx * 1664525u + 1013904223u + v) % RW; }
    const long long t1 = tick(idx);
    out[blockIdx.x] = (unsigned long long)(t1 - t0) / (unsigned)steps + (idx == 0xffffffffu ? 1 : 0);
}

static double median_of(unsigned long long* d_out, int n)
{
    std::vector<unsigned long long> h(n);
    CK(hipMemcpy(h.data(), d_out, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    return (double)h[n / 2];
}

int main()
{
    const int PW = 64, STEPS = 64; // 64 probing waves (8 per XCD), 64 dependent steps each
    uint4 *map, *other, *big; unsigned *sink, *hist; unsigned long long* out;
    CK(hipMalloc(&map, 8 * SLICE_BYTES)); CK(hipMalloc(&other, 12u << 20)); CK(hipMalloc(&big, 1u << 30)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&hist, 65536 * 4)); CK(hipMemset(hist, 0, 65536 * 4)); CK(hipMalloc(&out, 4096 * 8));
    CK(hipMemset(map, 0, 8 * SLICE_BYTES)); CK(hipMemset(other, 0, 12u << 20)); CK(hipMemset(big, 0, 1u << 30));
    hipStream_t s; CK(hipStreamCreate(&s));
    const unsigned* mw = reinterpret_cast<const unsigned*>(map);
    auto W = [&] { hipLaunchKernelGGL(warm, dim3(8 * 64), dim3(256), 0, s, map, sink); };
    auto P = [&](int shift) { hipLaunchKernelGGL(probe, dim3(PW), dim3(64), 0, s, mw, STEPS, shift, out); };
    auto flush = [&] { hipLaunchKernelGGL(stream<false>, dim3(2048), dim3(256), 0, s, big, (size_t)(1u << 30) / 16, sink); };
    for (int rep = 0; rep < 3; ++rep) {
        printf("--- repetition %d (cycles per dependent load, median of %d waves)\n", rep, PW);
        hipLaunchKernelGGL(same, dim3(PW), dim3(64), 0, s, mw, STEPS, out, sink); CK(hipStreamSynchronize(s));
        printf("same launch (L2 hit)                : %6.0f\n", median_of(out, PW));
        flush(); W(); P(0); CK(hipStreamSynchronize(s));
        printf("next launch, own XCD's slice        : %6.0f\n", median_of(out, PW));
        flush(); W(); P(4); CK(hipStreamSynchronize(s));
        printf("next launch, another XCD's slice    : %6.0f\n", median_of(out, PW));
        flush(); W(); W(); W(); P(0); CK(hipStreamSynchronize(s));
        printf("after three warm launches, own      : %6.0f\n", median_of(out, PW));
        flush(); W(); hipLaunchKernelGGL(stream<false>, dim3(2048), dim3(256), 0, s, other, (size_t)(12u << 20) / 16, sink); P(0); CK(hipStreamSynchronize(s));
        printf("12 MB streamed in between (plain)   : %6.0f\n", median_of(out, PW));
        flush(); W(); hipLaunchKernelGGL(stream<true>, dim3(2048), dim3(256), 0, s, other, (size_t)(12u << 20) / 16, sink); P(0); CK(hipStreamSynchronize(s));
        printf("12 MB streamed in between (nt)      : %6.0f\n", median_of(out, PW));
        W(); CK(hipStreamSynchronize(s)); flush(); P(0); CK(hipStreamSynchronize(s));
        printf("after 1 GiB of other traffic (HBM)  : %6.0f\n", median_of(out, PW));
        flush(); hipLaunchKernelGGL(warm_write, dim3(8 * 64), dim3(256), 0, s, map, other, hist, sink); P(0); CK(hipStreamSynchronize(s));
        printf("warm launch also stores 1.5 MB/XCD + atomics : %6.0f\n", median_of(out, PW));
        flush(); W(); hipLaunchKernelGGL(one_wg, dim3(1), dim3(64), 0, s, sink); P(0); CK(hipStreamSynchronize(s));
        printf("a one-workgroup launch in between   : %6.0f\n", median_of(out, PW));
        flush(); W(); hipLaunchKernelGGL(one_wg, dim3(3), dim3(64), 0, s, sink); P(0); CK(hipStreamSynchronize(s));
        printf("a three-workgroup launch in between : %6.0f\n", median_of(out, PW));
        {
            unsigned h[48];
            hipLaunchKernelGGL(xcc_of, dim3(16), dim3(64), 0, s, hist); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, hist, 16 * 4, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(one_wg, dim3(1), dim3(64), 0, s, sink);
            hipLaunchKernelGGL(xcc_of, dim3(16), dim3(64), 0, s, hist + 16); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h + 16, hist + 16, 16 * 4, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(one_wg, dim3(3), dim3(64), 0, s, sink);
            hipLaunchKernelGGL(xcc_of, dim3(16), dim3(64), 0, s, hist + 32); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h + 32, hist + 32, 16 * 4, hipMemcpyDeviceToHost));
            printf("XCC of workgroups 0..15: plain"); for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
            printf(" | behind a 1-workgroup launch"); for (int i = 0; i < 16; ++i) printf(" %u", h[16 + i]);
            printf(" | behind a 3-workgroup launch"); for (int i = 0; i < 16; ++i) printf(" %u", h[32 + i]);
            printf("\n");
        }
        // the same pair as a graph
        hipGraph_t g; hipGraphExec_t ge;
        flush(); CK(hipStreamSynchronize(s));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); W(); P(0); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        printf("graph {warm, probe}, own slice      : %6.0f\n", median_of(out, PW));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // bandwidth view: `warm` on a cold L2 (after the flush) and repeated
        hipEvent_t e[6]; for (auto& x : e) CK(hipEventCreate(&x));
        flush(); W(); // (first launch pays the MALL fill as well)
        for (int i = 0; i < 5; ++i) { CK(hipEventRecord(e[i], s)); W(); } CK(hipEventRecord(e[5], s)); CK(hipEventSynchronize(e[5]));
        printf("warm launch durations (us), repeated :");
        for (int i = 0; i < 5; ++i) { float ms; CK(hipEventElapsedTime(&ms, e[i], e[i + 1])); printf(" %.2f", ms * 1e3f); }
        printf("\n");
    }
    return 0;
}
