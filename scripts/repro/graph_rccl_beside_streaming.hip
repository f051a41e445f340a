// Minimal reproducer attempt (no torch, no library of ours) for profiles/r03g_dp_ride_crash.txt: hipGraphLaunch segfaulted when ONE
// captured graph held (i) a chain of kernels that stream ~4.6 GB of optimizer state (the riding BertAdam update) and (ii) RCCL
// all-reduces on a forked communication stream (world size 1, 7 buckets of ~85 MB), ROCm 7.2, gfx950.  The product works around it by
// capturing the iteration as two graphs.
//   hipcc --offload-arch=gfx950 -O2 scripts/repro/graph_rccl_beside_streaming.hip -o /tmp/grccl -lrccl && /tmp/grccl [kernels=200] [buckets=7]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { printf("FAIL %s:%d %s -> %s\n", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); return 2; } } while (0)

__global__ void stream_update(float* p, const float* g, size_t n) {          // 8 B in, 4 B out per element, grid-stride
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 0.999f + g[i] * 1e-3f;
}
__global__ void small_chain(float* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] += 1.0f; }

int main(int argc, char** argv) {
    int nk = argc > 1 ? atoi(argv[1]) : 200, nb = argc > 2 ? atoi(argv[2]) : 7;
    const size_t NP = 150u << 20, BUCKET = 21u << 20;                          // 150 M parameters; 21 M floats = 84 MB per bucket
    float *p, *g, *x;
    CK(hipMalloc(&p, NP * 4)); CK(hipMalloc(&g, NP * 4)); CK(hipMalloc(&x, 1 << 20));
    CK(hipMemset(p, 0, NP * 4)); CK(hipMemset(g, 0, NP * 4)); CK(hipMemset(x, 0, 1 << 20));
    ncclUniqueId id; ncclComm_t comm;
    NK(ncclGetUniqueId(&id));
    NK(ncclCommInitRank(&comm, 1, id, 0));
    hipStream_t a, c;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    NK(ncclAllReduce(g, g, BUCKET, ncclFloat, ncclAvg, comm, c));              // one eager collective first (RCCL's lazy setup outside the capture)
    CK(hipStreamSynchronize(c));
    std::vector<hipEvent_t> ev(nb + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
    // forward: a chain of small kernels, each third one followed by a streaming slice of the update (what rides in the forward products)
    for (int i = 0; i < nk; ++i) {
        hipLaunchKernelGGL(small_chain, dim3(64), dim3(256), 0, a, x, 1 << 14);
        if (i % 3 == 0) {
            size_t lo = (NP / nk) * i, len = NP / nk * 3;
            if (lo + len > NP) len = NP - lo;
            hipLaunchKernelGGL(stream_update, dim3(1024), dim3(256), 0, a, p + lo, g + lo, len);
        }
    }
    // backward: the chain again, with an exchange point every nk / nb kernels on the communication stream
    for (int i = 0, b = 0; i < nk; ++i) {
        hipLaunchKernelGGL(small_chain, dim3(64), dim3(256), 0, a, x, 1 << 14);
        if (b < nb && i % (nk / nb) == nk / nb - 1) {
            CK(hipEventRecord(ev[b], a)); CK(hipStreamWaitEvent(c, ev[b], 0));
            NK(ncclAllReduce(g + b * BUCKET, g + b * BUCKET, BUCKET, ncclFloat, ncclAvg, comm, c));
            ++b;
        }
    }
    CK(hipEventRecord(ev[nb], c)); CK(hipStreamWaitEvent(a, ev[nb], 0));        // join before the clip
    hipLaunchKernelGGL(small_chain, dim3(64), dim3(256), 0, a, x, 1 << 14);
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamEndCapture(a, &gr));
    size_t nn = 0; CK(hipGraphGetNodes(gr, nullptr, &nn));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    printf("graph with %zu nodes instantiated; launching 20 times ...\n", nn); fflush(stdout);
    for (int r = 0; r < 20; ++r) { CK(hipGraphLaunch(ge, a)); CK(hipStreamSynchronize(a)); }
    float h; CK(hipMemcpy(&h, x, 4, hipMemcpyDeviceToHost));
    printf("x[0] = %.0f (expected %d) -> %s\n", h, 20 * (2 * nk + 1), h == 20.f * (2 * nk + 1) ? "OK" : "MISMATCH");
    ncclCommDestroy(comm);
    return h == 20.f * (2 * nk + 1) ? 0 : 1;
}
