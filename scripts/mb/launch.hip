// Micro-benchmark: cost of one dependent kernel launch on the same stream -- direct launches from a tight C++ loop
// versus the same chain captured into a hipGraph.  Tells whether a C-level plan executor could beat graph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_small(float* x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
}
struct Big { char pad[200]; float* x; int n; };
__global__ __launch_bounds__(256) void k_bigarg(Big b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.n) b.x[i] = b.x[i] * 1.0001f + 1.0f;
}

int main() {
    float* x; CK(hipMalloc(&x, 1 << 22)); CK(hipMemset(x, 0, 1 << 22));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    Big b; b.x = x; b.n = 192 * 768;
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&] {
            if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
            else if (variant == 1) hipLaunchKernelGGL(k_small, dim3(576), dim3(256), 0, s, x, 192 * 768);
            else hipLaunchKernelGGL(k_bigarg, dim3(576), dim3(256), 0, s, b);
        };
        const char* name = variant == 0 ? "empty" : variant == 1 ? "small(576 wg)" : "small, 216-byte args";
        for (int i = 0; i < 100; ++i) launch();
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-24s direct: %6.2f us/launch\n", name, ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-24s graph : %6.2f us/launch\n", name, ms * 1e3 / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
