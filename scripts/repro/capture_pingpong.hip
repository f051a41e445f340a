// Minimal reproducer attempt (no torch, no library of ours) for profiles/r05t_capture_pingpong_crash.txt:
// hipStreamEndCapture segfaulted when a captured side stream ALTERNATED waits with its origin stream
// (origin -> side -> origin -> side ..., one small kernel between consecutive waits), ROCm 7.2, gfx950.
//   hipcc --offload-arch=gfx950 -O2 scripts/repro/capture_pingpong.hip -o /tmp/pingpong && /tmp/pingpong [alternations=64] [mode=global|threadlocal|relaxed]
// Exit 0 + "OK" = the pattern captures, instantiates and replays with the right result (checked on the host).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void bump(float* p, int n, float v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + v;          // order-dependent: a lost edge changes the result
}

int main(int argc, char** argv) {
    int alt = argc > 1 ? atoi(argv[1]) : 64;
    hipStreamCaptureMode mode = hipStreamCaptureModeGlobal;
    if (argc > 2 && !strcmp(argv[2], "threadlocal")) mode = hipStreamCaptureModeThreadLocal;
    if (argc > 2 && !strcmp(argv[2], "relaxed")) mode = hipStreamCaptureModeRelaxed;
    const int n = 1 << 20;
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * alt + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipStreamBeginCapture(a, mode));
    int k = 0;
    CK(hipEventRecord(ev[k], a)); CK(hipStreamWaitEvent(b, ev[k], 0)); ++k;          // fork
    for (int i = 0; i < alt; ++i) {
        hipLaunchKernelGGL(bump, dim3(n / 256), dim3(256), 0, a, d, n, 1.0f);          // origin works ...
        CK(hipEventRecord(ev[k], a)); CK(hipStreamWaitEvent(b, ev[k], 0)); ++k;      // ... side waits for it,
        hipLaunchKernelGGL(bump, dim3(n / 256), dim3(256), 0, b, d, n, 2.0f);          // works,
        CK(hipEventRecord(ev[k], b)); CK(hipStreamWaitEvent(a, ev[k], 0)); ++k;      // and the origin waits for the side
    }
    hipGraph_t g;
    printf("ending capture of %d alternations ...\n", alt); fflush(stdout);
    CK(hipStreamEndCapture(a, &g));
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, a));
    CK(hipStreamSynchronize(a));
    std::vector<float> h(4);
    CK(hipMemcpy(h.data(), d, 4 * sizeof(float), hipMemcpyDeviceToHost));
    float want = 0.f;
    for (int r = 0; r < 3; ++r) for (int i = 0; i < alt; ++i) { want = want * 1.0001f + 1.0f; want = want * 1.0001f + 2.0f; }
    printf("graph nodes %zu, result %.3f, expected %.3f -> %s\n", nn, h[0], want, h[0] == want ? "OK" : "MISMATCH");
    return h[0] == want ? 0 : 1;
}
