// Which runtime calls of ANOTHER thread invalidate a thread-local stream capture?  (r5: tests/test_gpu_runtime.py::test_two_threads_two_handles
// failed once with hipErrorStreamCaptureInvalidated in the capturing thread.)   hipcc --offload-arch=gfx950 -O2 -pthread capture_threads.hip -o capture_threads.bin
#include <hip/hip_runtime.h>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
__global__ void nop_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
static std::atomic<bool> stop{false};
static int capture_loop(hipStreamCaptureMode mode, int rounds)
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        hipGraph_t g = nullptr; hipGraphExec_t e = nullptr;
        hipError_t err = hipStreamBeginCapture(s, mode);
        for (int k = 0; k < 40 && err == hipSuccess; ++k) { hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); err = hipGetLastError(); }
        hipError_t e2 = hipStreamEndCapture(s, &g);
        if (err != hipSuccess || e2 != hipSuccess) { if (bad < 2) fprintf(stderr, "    capture broken: launch %s, end %s\n", hipGetErrorString(err), hipGetErrorString(e2)); ++bad; (void)hipGetLastError(); }
        else if (hipGraphInstantiate(&e, g, nullptr, nullptr, 0) == hipSuccess) { hipGraphLaunch(e, s); hipStreamSynchronize(s); hipGraphExecDestroy(e); }
        if (g) hipGraphDestroy(g);
    }
    hipStreamDestroy(s);
    return bad;
}
int main()
{
    const char* names[] = {"nothing", "hipMalloc + hipFree", "hipDeviceSynchronize", "hipStreamCreate + Destroy", "hipHostMalloc + hipHostFree", "hipStreamSynchronize (own stream)",
                           "hipMemcpy (synchronous)", "hipMemcpyAsync + sync (own stream)", "hipEventCreate + Destroy", "kernel launch (own stream)", "capture of its own (thread local)"};
    for (int mode_i = 0; mode_i < 2; ++mode_i) {
        const hipStreamCaptureMode mode = mode_i == 0 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
        printf("capturing thread in %s mode\n", mode_i == 0 ? "ThreadLocal" : "Relaxed");
        for (int what = 0; what < 11; ++what) {
            stop = false;
            int other_err = 0;
            std::thread b([&] {
                hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                void* d = nullptr; hipMalloc(&d, 1 << 20); void* h = malloc(1 << 20);
                while (!stop) {
                    hipError_t e = hipSuccess;
                    switch (what) {
                    case 1: { void* p = nullptr; e = hipMalloc(&p, 1 << 22); if (e == hipSuccess) e = hipFree(p); break; }
                    case 2: e = hipDeviceSynchronize(); break;
                    case 3: { hipStream_t t; e = hipStreamCreateWithFlags(&t, hipStreamNonBlocking); if (e == hipSuccess) e = hipStreamDestroy(t); break; }
                    case 4: { void* p = nullptr; e = hipHostMalloc(&p, 1 << 16, hipHostMallocMapped); if (e == hipSuccess) e = hipHostFree(p); break; }
                    case 5: e = hipStreamSynchronize(s); break;
                    case 6: e = hipMemcpy(d, h, 1 << 16, hipMemcpyHostToDevice); break;
                    case 7: e = hipMemcpyAsync(d, h, 1 << 16, hipMemcpyHostToDevice, s); if (e == hipSuccess) e = hipStreamSynchronize(s); break;
                    case 8: { hipEvent_t ev; e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e == hipSuccess) e = hipEventDestroy(ev); break; }
                    case 9: hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); e = hipGetLastError(); break;
                    case 10: if (capture_loop(hipStreamCaptureModeThreadLocal, 1)) e = hipErrorUnknown; break;
                    default: std::this_thread::yield();
                    }
                    if (e != hipSuccess) { if (other_err < 2) fprintf(stderr, "    other thread: %s\n", hipGetErrorString(e)); ++other_err; (void)hipGetLastError(); }
                }
                hipFree(d); free(h); hipStreamDestroy(s);
            });
            const int bad = capture_loop(mode, 300);
            stop = true; b.join();
            printf("  other thread does %-36s: %3d of 300 captures broken, %d errors in the other thread\n", names[what], bad, other_err);
            fflush(stdout);
        }
    }
    return 0;
}
