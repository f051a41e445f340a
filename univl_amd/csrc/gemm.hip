// MFMA GEMM for every dense contraction on the UniVL hot path:
//
//      C[M,N] = epilogue( alpha * A_op[M,Kc] . B_op[N,Kc]^T )
//
// Each operand may be given K-major (row-major [rows][Kc], the layout of nn.Linear's x and W) or T-major
// (row-major [Kc][rows]): forward uses (K,K); dgrad dX = dY.W uses A K-major, B = W T-major; wgrad
// dW = dY^T.X uses both T-major.  T-major tiles are staged row-major in LDS and consumed with the gfx950
// transpose read (ds_read_b64_tr_b16) for bf16, plain strided b32 reads for f32 -- no transposed copies of
// weights or activations ever exist in HBM.
//
// Workgroup = 4 waves (2x2), tile BM x BN in {64x64, 128x128}, BK = 2 chunks (64 bf16 / 32 f32), global ->
// registers -> LDS staging with one barrier per K step (loads of tile k+1 are in flight while tile k is
// multiplied).  Optional split-K over gridDim.z accumulates with fp32 atomics into a pre-zeroed C.
//
// Epilogue (all optional, in this order): *alpha, +bias[n], +residual[m,n] (fp32), erf-GELU forward (saving the
// pre-activation), *gelu'(saved pre-activation), +C_old (accumulate), store fp32 and/or T.  A wgrad launch can
// also emit the bias gradient (row sums of A_op over the contraction) from the tiles it already staged.
#include "common.h"
#include "univl_hip.h"

namespace {

struct GemmArgs {
    const void* A; const void* B;
    long lda, ldb;
    int M, N, K;
    float* C32; void* C16; long ldc;
    const float* bias; const float* R; long ldr;
    void* aux; long ldaux;
    float* dbias;
    float alpha;
    int flags;
    int ksplit_len;   // contraction length handled by one z-slice (multiple of BK)
};

template <typename T, bool TR, int ROWS, int BK> struct Tile {
    static constexpr int EPC = Mma<T>::EPC;
    static constexpr int PITCH = TR ? (ROWS + Mma<T>::tpad) : (BK + Mma<T>::kpad);
    static constexpr int ELEMS = TR ? BK * PITCH : ROWS * PITCH;
    static constexpr int CHUNKS = ROWS * BK / EPC;
    static constexpr int PER_THREAD = CHUNKS / 256;
    static_assert(CHUNKS % 256 == 0, "tile must split evenly over 256 threads");

    // global -> registers.  `row0` first tile row, `k0` first contraction index, `rows_total`/`k_end` bounds.
    __device__ static __forceinline__ void load(uint4 (&r)[PER_THREAD], const T* base, long ld, int row0, int k0,
                                                int rows_total, int k_end, int tid) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            const int ch = tid + 256 * c;
            int row, kk;
            if (TR) { kk = ch / (ROWS / EPC); row = (ch % (ROWS / EPC)) * EPC; }
            else    { row = ch / (BK / EPC);  kk = (ch % (BK / EPC)) * EPC; }
            const int gr = row0 + row, gk = k0 + kk;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (gr < rows_total && gk < k_end) {
                const T* p = TR ? (base + (long)gk * ld + gr) : (base + (long)gr * ld + gk);
                v = *reinterpret_cast<const uint4*>(p);
                // ragged tail inside a 16-byte vector (only possible when an extent is not a multiple of EPC):
                // zero the out-of-range elements so that padding never contributes.
                const int lim = TR ? (rows_total - gr) : (k_end - gk);
                if (lim < EPC) {
                    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
                    for (int j = 0; j < EPC; ++j) if (j >= lim) e[j] = from_f32<T>(0.0f);
                }
            }
            r[c] = v;
        }
    }
    __device__ static __forceinline__ void store(const uint4 (&r)[PER_THREAD], T* lds, int tid) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            const int ch = tid + 256 * c;
            int off;
            if (TR) { const int kk = ch / (ROWS / EPC), row = (ch % (ROWS / EPC)) * EPC; off = kk * PITCH + row; }
            else    { const int row = ch / (BK / EPC), kk = (ch % (BK / EPC)) * EPC;    off = row * PITCH + kk; }
            *reinterpret_cast<uint4*>(lds + off) = r[c];
        }
    }
    // fragment for the 16 tile rows starting at `r16`, chunk `c` (contraction offset c*CH)
    __device__ static __forceinline__ typename Mma<T>::frag frag(const T* lds, int r16, int c, int lane) {
        if (TR) return Mma<T>::lds_tmajor(lds + (c * Mma<T>::CH) * PITCH + r16, PITCH, lane);
        return Mma<T>::lds_kmajor(lds + (r16 + (lane & 15)) * PITCH + c * Mma<T>::CH, lane >> 4);
    }
};

template <typename T, bool TA, bool TB, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int CH = Mma<T>::CH;
    constexpr int BK = 2 * CH;
    using TileA = Tile<T, TA, BM, BK>;
    using TileB = Tile<T, TB, BN, BK>;
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave sub-tile
    constexpr int MI = WM / 16, NI = WN / 16;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sA = reinterpret_cast<T*>(smem_raw);
    T* sB = sA + 2 * TileA::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.ksplit_len;
    const int kend = min(p.K, kbeg + p.ksplit_len);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 ra[TileA::PER_THREAD], rb[TileB::PER_THREAD];
    const bool want_dbias = TA && (p.dbias != nullptr) && (blockIdx.x == 0);
    float dbias_acc = 0.0f;

    if (nk > 0) {
        TileA::load(ra, A, p.lda, m0, kbeg, p.M, kend, tid);
        TileB::load(rb, B, p.ldb, n0, kbeg, p.N, kend, tid);
        TileA::store(ra, sA, tid);
        TileB::store(rb, sB, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const T* cA = sA + cur * TileA::ELEMS;
        const T* cB = sB + cur * TileB::ELEMS;
        if (kt + 1 < nk) {
            TileA::load(ra, A, p.lda, m0, kbeg + (kt + 1) * BK, p.M, kend, tid);
            TileB::load(rb, B, p.ldb, n0, kbeg + (kt + 1) * BK, p.N, kend, tid);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            typename Mma<T>::frag fa[MI], fb[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) fa[a] = TileA::frag(cA, wm0 + 16 * a, c, lane);
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[b] = TileB::frag(cB, wn0 + 16 * b, c, lane);
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = Mma<T>::mma(fa[a], fb[b], acc[a][b]);
        }
        if (want_dbias) {
            // bias gradient = sum over the contraction (tokens) of A_op rows; A is T-major: [BK][BM]
            if (tid < BM) {
#pragma unroll 8
                for (int kk = 0; kk < BK; ++kk) dbias_acc += to_f32<T>(cA[kk * TileA::PITCH + tid]);
            }
        }
        if (kt + 1 < nk) {
            TileA::store(ra, sA + (cur ^ 1) * TileA::ELEMS, tid);
            TileB::store(rb, sB + (cur ^ 1) * TileB::ELEMS, tid);
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------------------------------ epilogue
    const bool atomic = (p.flags & UNIVL_GEMM_ATOMIC) != 0;
    const bool first_slice = (blockIdx.z == 0);
    T* C16 = reinterpret_cast<T*>(p.C16);
    T* aux = reinterpret_cast<T*>(p.aux);
#pragma unroll
    for (int a = 0; a < MI; ++a) {
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int col = n0 + wn0 + 16 * b + i;
            if (col >= p.N) continue;
            const float bv = (p.bias && first_slice) ? p.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + 16 * a + 4 * g + r;
                if (row >= p.M) continue;
                float v = acc[a][b][r] * p.alpha + bv;
                if (p.R && first_slice) v += p.R[(long)row * p.ldr + col];
                if (p.flags & UNIVL_GEMM_GELU_FWD) {
                    aux[(long)row * p.ldaux + col] = from_f32<T>(v);
                    v = gelu_f(v);
                }
                if (p.flags & UNIVL_GEMM_GELU_BWD) v *= gelu_grad_f(to_f32<T>(aux[(long)row * p.ldaux + col]));
                const long o = (long)row * p.ldc + col;
                if (atomic) {
                    unsafeAtomicAdd(p.C32 + o, v);
                } else {
                    if (p.flags & UNIVL_GEMM_ACCUM) v += p.C32[o];
                    if (p.C32) p.C32[o] = v;
                    if (C16) C16[o] = from_f32<T>(v);
                }
            }
        }
    }
    if (want_dbias && tid < BM) {
        const int row = m0 + tid;
        if (row < p.M) {
            if (gridDim.z > 1 || (p.flags & UNIVL_GEMM_DBIAS_ATOMIC)) unsafeAtomicAdd(p.dbias + row, dbias_acc);
            else if (p.flags & UNIVL_GEMM_ACCUM) p.dbias[row] += dbias_acc;
            else p.dbias[row] = dbias_acc;
        }
    }
}

template <typename T, bool TA, bool TB, int BM, int BN>
int launch(const GemmArgs& a, int ksplit, hipStream_t stream) {
    constexpr int BK = 2 * Mma<T>::CH;
    using TileA = Tile<T, TA, BM, BK>;
    using TileB = Tile<T, TB, BN, BK>;
    const size_t smem = 2 * (TileA::ELEMS + TileB::ELEMS) * sizeof(T);
    static bool attr_done = false;   // per instantiation
    if (!attr_done && smem > 48 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, TA, TB, BM, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, ksplit);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, BM, BN>), grid, dim3(256), smem, stream, a);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

template <typename T, int BM, int BN>
int dispatch_trans(const GemmArgs& a, int ta, int tb, int ksplit, hipStream_t s) {
    if (!ta && !tb) return launch<T, false, false, BM, BN>(a, ksplit, s);
    if (!ta && tb) return launch<T, false, true, BM, BN>(a, ksplit, s);
    if (ta && tb) return launch<T, true, true, BM, BN>(a, ksplit, s);
    return launch<T, true, false, BM, BN>(a, ksplit, s);
}

}  // namespace

extern "C" int univl_gemm(const UnivlGemm* d, hipStream_t stream) {
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "univl_gemm: null descriptor");
    UNIVL_CHECK_ARG(d->dtype == UNIVL_F32 || d->dtype == UNIVL_BF16, UNIVL_EUNSUPPORTED, "univl_gemm: dtype %d", d->dtype);
    UNIVL_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, UNIVL_EINVAL, "univl_gemm: empty problem %dx%dx%d", d->M, d->N, d->K);
    UNIVL_CHECK_ARG(d->A && d->B && (d->C32 || d->C16), UNIVL_EINVAL, "univl_gemm: null operand");
    const int epc = d->dtype == UNIVL_BF16 ? 8 : 4;
    UNIVL_CHECK_ARG(aligned16(d->A) && aligned16(d->B) && d->lda % epc == 0 && d->ldb % epc == 0, UNIVL_EALIGN,
                    "univl_gemm: operands must be 16-byte aligned with leading dims a multiple of %d (lda=%ld ldb=%ld)",
                    epc, d->lda, d->ldb);
    const int flags = d->flags;
    UNIVL_CHECK_ARG(!((flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD)) && !d->aux), UNIVL_EINVAL,
                    "univl_gemm: GELU epilogue needs aux");
    int ksplit = d->ksplit < 1 ? 1 : d->ksplit;
    const int BK = d->dtype == UNIVL_BF16 ? 64 : 32;
    int klen = ((d->K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    ksplit = (d->K + klen - 1) / klen;
    if (ksplit > 1) {
        UNIVL_CHECK_ARG(d->C32 && !d->C16 && !(flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD)), UNIVL_EINVAL,
                        "univl_gemm: split-K needs a pre-zeroed fp32 output and a linear epilogue");
    }
    UNIVL_CHECK_ARG(!(d->dbias && !d->trans_a), UNIVL_EINVAL, "univl_gemm: dbias only with T-major A (wgrad)");
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.lda = d->lda; a.ldb = d->ldb; a.M = d->M; a.N = d->N; a.K = d->K;
    a.C32 = d->C32; a.C16 = d->C16; a.ldc = d->ldc; a.bias = d->bias; a.R = d->R; a.ldr = d->ldr;
    a.aux = d->aux; a.ldaux = d->ldaux; a.dbias = d->dbias; a.alpha = d->alpha;
    a.flags = flags | (ksplit > 1 ? UNIVL_GEMM_ATOMIC : 0);
    a.ksplit_len = klen;
    // tile choice: 128x128 once the grid fills the chip twice over, else 64x64 for parallelism
    const long tiles128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
    const bool big = d->tile == 128 || (d->tile == 0 && tiles128 >= 384);
    if (d->dtype == UNIVL_BF16) {
        return big ? dispatch_trans<__bf16, 128, 128>(a, d->trans_a, d->trans_b, ksplit, stream)
                   : dispatch_trans<__bf16, 64, 64>(a, d->trans_a, d->trans_b, ksplit, stream);
    }
    return big ? dispatch_trans<float, 128, 128>(a, d->trans_a, d->trans_b, ksplit, stream)
               : dispatch_trans<float, 64, 64>(a, d->trans_a, d->trans_b, ksplit, stream);
}
