// Micro-benchmark: how fast can ONE workgroup per CU stream K-major operand tiles (64 rows x 256 B, two of them per
// K step = 32 KB) from L2/HBM, by LDS-DMA with 1/2/3 tiles in flight, or by plain register loads?  Answers what bounds
// the small-M GEMM's K step.  Build: hipcc --offload-arch=gfx950 -O3 stream.hip -o stream; run: ./stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int TILE = 16384;            // bytes per operand tile (64 rows x 256 B)
constexpr int PT = 4;                  // 16-B pieces per thread per operand tile

__device__ __forceinline__ void issue(const char* const (&p)[PT], unsigned char* stage, int tid) {
    unsigned char* w = stage + (tid & ~63) * 16;
#pragma unroll
    for (int c = 0; c < PT; ++c)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p[c],
                                         (__attribute__((address_space(3))) void*)(w + c * 4096), 16, 0, 0);
}

// MODE 0: glds, NS stages, wait for tile t only (vmcnt((NS-2)*8)) -> NS-1 tiles in flight during "compute"
// MODE 1: register loads, 1 tile ahead
template <int NS, bool COMPUTE>
__global__ __launch_bounds__(256) void k_glds(const char* A, const char* B, long ld, int nsteps, float* out, int shareB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const char* pa[PT]; const char* pb[PT];
    const long wg = blockIdx.x;
#pragma unroll
    for (int c = 0; c < PT; ++c) {
        const int L = tid + 256 * c, r = L >> 4, q = (L & 15) ^ (r & 15);
        pa[c] = A + (long)r * ld + q * 16;                                  // shared by all workgroups
        pb[c] = B + ((shareB ? 0 : wg * 64) + r) * ld + q * 16;             // unique per workgroup
    }
    unsigned char* sA = smem; unsigned char* sB = smem + NS * TILE;
    float acc = 0.f;
    // prologue: NS-1 tiles
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        issue(pa, sA + s * TILE, tid); issue(pb, sB + s * TILE, tid);
#pragma unroll
        for (int c = 0; c < PT; ++c) { pa[c] += 256; pb[c] += 256; }
    }
    int st = 0;
    for (int t = 0; t < nsteps; ++t) {
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NS == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (NS == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int sn = (st + NS - 1) % NS;
        issue(pa, sA + sn * TILE, tid); issue(pb, sB + sn * TILE, tid);     // (reads past the end are harmless: buffer is padded)
#pragma unroll
        for (int c = 0; c < PT; ++c) { pa[c] += 256; pb[c] += 256; }
        if (COMPUTE) {
            // 32 ds_read_b64 + 16 MFMA per wave, like the 64x64 bf16 tile
            typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 a4[4] = {};
            const int lane = tid & 63, i = lane & 15, g = lane >> 4, wv = tid >> 6;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf8 fa[2], fb[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int row = (wv >> 1) * 32 + 16 * x + i, q = c * 4 + g;
                    fa[x] = *reinterpret_cast<const bf8*>(sA + st * TILE + row * 256 + ((q ^ (row & 15)) << 4));
                    const int rowb = (wv & 1) * 32 + 16 * x + i;
                    fb[x] = *reinterpret_cast<const bf8*>(sB + st * TILE + rowb * 256 + ((q ^ (rowb & 15)) << 4));
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) a4[x * 2 + y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[x], fb[y], a4[x * 2 + y], 0, 0, 0);
            }
            acc += a4[0][0] + a4[1][1] + a4[2][2] + a4[3][3];
        } else {
            acc += *reinterpret_cast<const float*>(sA + st * TILE + tid * 4);
        }
        st = (st + 1) % NS;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) out[0] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_regs(const char* A, const char* B, long ld, int nsteps, float* out, int shareB) {
    const int tid = threadIdx.x;
    const char* pa[PT]; const char* pb[PT];
    const long wg = blockIdx.x;
#pragma unroll
    for (int c = 0; c < PT; ++c) {
        const int L = tid + 256 * c, r = L >> 4, q = L & 15;
        pa[c] = A + (long)r * ld + q * 16;
        pb[c] = B + ((shareB ? 0 : wg * 64) + r) * ld + q * 16;
    }
    u32x4 ring[DEPTH][2 * PT];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
#pragma unroll
        for (int c = 0; c < PT; ++c) { ring[s][c] = *(const u32x4*)pa[c]; ring[s][PT + c] = *(const u32x4*)pb[c]; pa[c] += 256; pb[c] += 256; }
    }
    unsigned acc = 0;
    for (int t = 0; t < nsteps; t += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            unsigned x = 0;
#pragma unroll
            for (int c = 0; c < 2 * PT; ++c) x ^= ring[s][c][0] ^ ring[s][c][3];
            acc += x;
#pragma unroll
            for (int c = 0; c < PT; ++c) { ring[s][c] = *(const u32x4*)pa[c]; ring[s][PT + c] = *(const u32x4*)pb[c]; pa[c] += 256; pb[c] += 256; }
        }
    }
    if (acc == 0x12345u) out[0] = (float)acc;
}

template <typename F> double timeit(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const long KB = 200 * 256 + 4096;         // bytes per row (>= (nsteps + 4) * 256)
    const int maxwg = 512;
    char* A; char* B; float* out;
    CK(hipMalloc(&A, 64 * KB)); CK(hipMalloc(&B, (long)maxwg * 64 * KB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(A, 1, 64 * KB)); CK(hipMemset(B, 1, (long)maxwg * 64 * KB));
    auto run = [&](const char* name, auto kern, size_t smem, int wgs, int shareB) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        double t[2]; int ns[2] = {48, 192};
        for (int j = 0; j < 2; ++j) t[j] = timeit([&] { hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), smem, 0, A, B, KB, ns[j], out, shareB); }, 200);
        const double step = (t[1] - t[0]) / (ns[1] - ns[0]);
        printf("%-34s wgs=%3d shareB=%d: %7.3f us/step  -> %6.1f GB/s per WG, %6.2f TB/s aggregate (fixed %.1f us)\n", name, wgs, shareB, step,
               32768.0 / step * 1e-3, 32768.0 * wgs / step * 1e-6, t[0] - 48 * step);
    };
    for (int wgs : {36, 144, 256, 512}) {
        for (int sh : {0, 1}) {
            run("glds 2-stage (1 in flight)", k_glds<2, false>, 2 * 2 * TILE, wgs, sh);
            run("glds 3-stage (2 in flight)", k_glds<3, false>, 3 * 2 * TILE, wgs, sh);
            run("glds 4-stage (3 in flight)", k_glds<4, false>, 4 * 2 * TILE, wgs, sh);
            run("glds 2-stage + mfma", k_glds<2, true>, 2 * 2 * TILE, wgs, sh);
            run("glds 3-stage + mfma", k_glds<3, true>, 3 * 2 * TILE, wgs, sh);
            run("glds 4-stage + mfma", k_glds<4, true>, 4 * 2 * TILE, wgs, sh);
            run("regs depth 1", k_regs<1>, 0, wgs, sh);
            run("regs depth 2", k_regs<2>, 0, wgs, sh);
            run("regs depth 4", k_regs<4>, 0, wgs, sh);
        }
    }
    return 0;
}
