// Micro-benchmark behind the "whole contraction in one burst" GEMM for M <= a few hundred rows (round 2):
// how long does it take ONE workgroup per CU to pull its complete operand set -- A[BM x K] (shared by all column tiles,
// L2 hits) + W[BN x K] (unique per column tile, HBM / Infinity-Cache cold) -- into LDS by LDS-DMA when EVERY piece is
// issued before the first wait (one vmcnt(0) + one barrier), compared with the same bytes in dependent 128-deep steps
// (two stages, a wait + barrier per step: the round-1 kernel's K loop)?
//   hipcc --offload-arch=gfx950 -O3 burst.hip -o burst && ./burst
// Prints microseconds per launch (events around a back-to-back train of launches on one stream, minus the same train of an
// empty kernel) for grids of 72 / 144 / 216 / 256 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// One K-major sub-tile: ROWS rows x 256 B (128 bf16), pieces of 16 B, XOR-swizzled on the source side like gemm.hip.
template <int ROWS>
__device__ __forceinline__ void issue_tile(const char* base, long ld, int k0_bytes, unsigned char* stage, int tid) {
    constexpr int PER = ROWS * 16 / 256;                       // 16-B pieces per thread
    unsigned char* w = stage + (tid & ~63) * 16;
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        const int L = tid + 256 * c, r = L >> 4, q = (L & 15) ^ (r & 15);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long)r * ld + k0_bytes + q * 16),
                                         (__attribute__((address_space(3))) void*)(w + c * 4096), 16, 0, 0);
    }
}

// MODE 0: all NSTEP sub-tiles of A and W in flight at once, one wait.  MODE 1: two stages, wait + barrier per step.
template <int BM, int BN, int NSTEP, int MODE>
__global__ __launch_bounds__(256) void k_load(const char* A, const char* W, long ld, int tiles_m, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const char* a = A + (long)tm * BM * ld;
    const char* w = W + (long)tn * BN * ld;
    constexpr int TA = BM * 256, TB = BN * 256;
    if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            issue_tile<BM>(a, ld, s * 256, smem + s * (TA + TB), tid);
            issue_tile<BN>(w, ld, s * 256, smem + s * (TA + TB) + TA, tid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        issue_tile<BM>(a, ld, 0, smem, tid);
        issue_tile<BN>(w, ld, 0, smem + TA, tid);
        for (int s = 0; s < NSTEP; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (s + 1 < NSTEP) {
                unsigned char* st = smem + ((s + 1) & 1) * (TA + TB);
                issue_tile<BM>(a, ld, (s + 1) * 256, st, tid);
                issue_tile<BN>(w, ld, (s + 1) * 256, st + TA, tid);
            }
        }
    }
    // touch the data so that nothing is optimised away
    float acc = 0.f;
    const float* f = reinterpret_cast<const float*>(smem);
    for (int i = tid; i < (MODE == 0 ? NSTEP : 2) * (TA + TB) / 4; i += 256 * 16) acc += f[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }

template <int BM, int BN, int NSTEP, int MODE>
float run(const char* A, const char* W, long ld, int M, int N, float* out, int reps, hipStream_t s) {
    const int tiles_m = M / BM, tiles_n = N / BN;
    const size_t smem = (size_t)(MODE == 0 ? NSTEP : 2) * (BM + BN) * 256;
    if (smem > 160 * 1024) return -1.f;                          // does not fit the 160 KB LDS
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_load<BM, BN, NSTEP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_load<BM, BN, NSTEP, MODE>), dim3(tiles_m * tiles_n), dim3(256), smem, s, A, W, ld, tiles_m, out);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_load<BM, BN, NSTEP, MODE>), dim3(tiles_m * tiles_n), dim3(256), smem, s, A, W + (size_t)(i % 64) * 3072 * ld, ld, tiles_m, out);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main() {
    const int M = 192, K = 768, NMAX = 3072;
    const long ld = K * 2;
    char *A, *W; float* out;
    CK(hipMalloc(&A, (size_t)M * ld)); CK(hipMalloc(&W, (size_t)64 * NMAX * ld + (1 << 20))); CK(hipMalloc(&out, 1 << 16));
    CK(hipMemset(A, 1, (size_t)M * ld)); CK(hipMemset(W, 1, (size_t)64 * NMAX * ld));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int reps = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, out);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float base; CK(hipEventElapsedTime(&base, e0, e1)); base = base * 1000.f / reps;
    printf("empty-kernel train: %.2f us per launch (subtract from the numbers below for pure load time)\n", base);
    printf("M=192 K=768 (6 steps of 128), W rotates over 64 buffers = 302 MB (HBM-cold); us per launch:\n");
#define ROW(BM, BN, N) \
    printf("  tile %3dx%-3d N=%4d (%3d WGs, %3d KB/WG)  burst %.2f   stepped %.2f\n", BM, BN, N, (M / BM) * (N / BN), 6 * (BM + BN) / 4, \
           run<BM, BN, 6, 0>(A, W, ld, M, N, out, reps, s), run<BM, BN, 6, 1>(A, W, ld, M, N, out, reps, s));
    ROW(64, 64, 768)  ROW(64, 64, 2304) ROW(64, 64, 3072)
    ROW(64, 32, 768)  ROW(64, 32, 2304) ROW(64, 32, 3072)
    ROW(32, 64, 768)  ROW(32, 64, 2304) ROW(32, 64, 3072)
    ROW(32, 32, 768)  ROW(32, 32, 2304)
    return 0;
}
