// dispatch_bench.hip -- how long does the chip take just to START (and retire) the waves of a launch?  Kernels with almost no work,
// for the grid shapes of the NN kernels (r3): workgroups x threads, with and without an LDS allocation, with a short dependent
// chain (two global loads) to stand in for a wave's life.  Prints the mean kernel duration by HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDSW, int CHAIN>
__global__ void k(const int* __restrict__ in, int* __restrict__ out, int n)
{
    __shared__ int lds[LDSW > 0 ? LDSW : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int v = i;
    if (LDSW > 0) { lds[threadIdx.x] = i; __syncthreads(); v = lds[(threadIdx.x + 1) % blockDim.x]; }
#pragma unroll
    for (int c = 0; c < CHAIN; ++c) v = in[(unsigned)(v * 97 + c) % (unsigned)n]; // dependent gathers
    if (v == -12345) out[i] = v;                                                    // (never: keeps the chain alive)
}

template <int LDSW, int CHAIN>
float run(int grid, int block, const int* in, int* out, int n, hipStream_t s)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<LDSW, CHAIN>), dim3(grid), dim3(block), 0, s, in, out, n);
    hipEventRecord(e0, s);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<LDSW, CHAIN>), dim3(grid), dim3(block), 0, s, in, out, n);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main()
{
    const int n = 1 << 22;
    int *in, *out; hipMalloc(&in, n * sizeof(int)); hipMalloc(&out, n * sizeof(int));
    std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 2654435761u) >> 10;
    hipMemcpy(in, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    const int shapes[][2] = {{1564, 64}, {1564, 192}, {1564, 256}, {3128, 128}, {6256, 64}, {12512, 64}, {256, 64}, {25024, 64}};
    printf("back-to-back launches, us per launch (includes the ~launch-to-launch gap of an eager stream)\n");
    for (auto& sh : shapes) {
        printf("grid %6d x %3d : no-LDS chain0 %6.2f | no-LDS chain2 %6.2f | no-LDS chain8 %6.2f | LDS 20KB chain2 %6.2f | LDS 20KB chain8 %6.2f\n", sh[0], sh[1],
               run<0, 0>(sh[0], sh[1], in, out, n, s), run<0, 2>(sh[0], sh[1], in, out, n, s), run<0, 8>(sh[0], sh[1], in, out, n, s),
               run<5120, 2>(sh[0], sh[1], in, out, n, s), run<5120, 8>(sh[0], sh[1], in, out, n, s));
    }
    return 0;
}
