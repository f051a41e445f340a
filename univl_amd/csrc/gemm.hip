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
// register ring (D K-tiles in flight) -> double-buffered LDS, one barrier per K step.  Optional split-K over gridDim.z accumulates with fp32 atomics into a pre-zeroed C.
//
// Epilogue (all optional, in this order): *alpha, +bias[n], +residual[m,n] (fp32), erf-GELU forward (saving the
// pre-activation), *gelu'(saved pre-activation), +C_old (accumulate), store fp32 and/or T.  A wgrad launch can
// also emit the bias gradient (row sums of A_op over the contraction) from the tiles it already staged.
#include <stdlib.h>
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

    // Per-thread source pointers of its 16-byte pieces.  Everything that does not change along K (row clamp, piece
    // coordinates) is folded into the pointer ONCE; a K step is then `load; pointer += step` -- the inner loop of
    // a small-M GEMM runs one wave per SIMD and is bound by instruction issue, not by MFMA or HBM.

    __device__ static __forceinline__ void coords(int c, int tid, int& row, int& kk) {
        const int ch = tid + 256 * c;
        if (TR) { kk = ch / (ROWS / EPC); row = (ch % (ROWS / EPC)) * EPC; }
        else    { row = ch / (BK / EPC);  kk = (ch % (BK / EPC)) * EPC; }
    }
    // Rows beyond rows_total are CLAMPED, not masked: they only feed accumulator rows/columns the epilogue never
    // stores.  (Only the contraction direction needs zero fill, and only in the last, partial K tile.)
    __device__ static __forceinline__ void init(const T* (&P)[PER_THREAD], const T* base, long ld, int row0, int k0, int rows_total, int tid) {
        const int rmax = TR ? ((rows_total - 1) / EPC) * EPC : rows_total - 1;
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int row, kk;
            coords(c, tid, row, kk);
            const int gr = min(row0 + row, rmax);
            P[c] = TR ? (base + (long)(k0 + kk) * ld + gr) : (base + (long)gr * ld + (k0 + kk));
        }
    }
    __device__ static __forceinline__ long kstep(long ld) { return TR ? (long)BK * ld : (long)BK; }
    __device__ static __forceinline__ void load_fast(u32x4_t (&r)[PER_THREAD], const T* const (&P)[PER_THREAD]) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) r[c] = *reinterpret_cast<const u32x4_t*>(P[c]);
    }
    __device__ static __forceinline__ void advance(const T* (&P)[PER_THREAD], long step) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) P[c] += step;
    }
    __device__ static __forceinline__ void store_fast(const u32x4_t (&r)[PER_THREAD], T* lds, int tid) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int row, kk;
            coords(c, tid, row, kk);
            *reinterpret_cast<u32x4_t*>(lds + (TR ? kk * PITCH + row : row * PITCH + kk)) = r[c];
        }
    }
    // last, partial K tile (krem < BK contraction indices left; P + skip points at the tile start): clamped addresses,
    // then zero fill of everything at or beyond krem when the registers go to LDS
    __device__ static __forceinline__ void load_tail(u32x4_t (&r)[PER_THREAD], const T* const (&P)[PER_THREAD], long skip, long ld, int krem, int tid) {
        const int klast = TR ? krem - 1 : ((krem - 1) / EPC) * EPC;
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int row, kk;
            coords(c, tid, row, kk);
            const long back = (long)(min(kk, klast) - kk) * (TR ? ld : 1);
            r[c] = *reinterpret_cast<const u32x4_t*>(P[c] + skip + back);
        }
    }
    __device__ static __forceinline__ void store_tail(const u32x4_t (&r)[PER_THREAD], T* lds, int krem, int tid) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int row, kk;
            coords(c, tid, row, kk);
            u32x4_t v = r[c];
            const int lim = TR ? (kk < krem ? EPC : 0) : max(0, min(EPC, krem - kk));
            if (lim < EPC) {
                constexpr int EPD = 4 / (int)sizeof(T);      // elements per dword
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned m = 0u;
                    if (EPD == 1) m = (d < lim) ? 0xFFFFFFFFu : 0u;
                    else m = ((2 * d < lim) ? 0x0000FFFFu : 0u) | ((2 * d + 1 < lim) ? 0xFFFF0000u : 0u);
                    v[d] &= m;
                }
            }
            *reinterpret_cast<u32x4_t*>(lds + (TR ? kk * PITCH + row : row * PITCH + kk)) = v;
        }
    }
    // fragment for the 16 tile rows starting at `r16`, chunk `c` (contraction offset c*CH)
    __device__ static __forceinline__ typename Mma<T>::frag frag(const T* lds, int r16, int c, int lane) {
        if (TR) return Mma<T>::lds_tmajor(lds + (c * Mma<T>::CH) * PITCH + r16, PITCH, lane);
        return Mma<T>::lds_kmajor(lds + (r16 + (lane & 15)) * PITCH + c * Mma<T>::CH, lane >> 4);
    }
};

template <typename T, bool TA, bool TB, int BM, int BN, int D, int NC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int CH = Mma<T>::CH;
    constexpr int BK = NC * CH;          // contraction depth of one LDS stage (NC chunks of CH)
    using TileA = Tile<T, TA, BM, BK>;
    using TileB = Tile<T, TB, BN, BK>;
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave sub-tile
    constexpr int MI = WM / 16, NI = WN / 16;
    static_assert(D == 2, "two register stages");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sA = reinterpret_cast<T*>(smem_raw);
    T* sB = sA + 2 * TileA::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.ksplit_len;
    const int kend = min(p.K, kbeg + p.ksplit_len);
    const int nfull = (kend - kbeg) / BK;              // full K tiles: no masking at all
    const int krem = (kend - kbeg) - nfull * BK;       // > 0: one partial tile at the end

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const bool want_dbias = TA && (p.dbias != nullptr) && (blockIdx.x == 0);
    float dbias_acc = 0.0f;

    auto compute = [&](const T* cA, const T* cB) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
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
    };

    const T* pa[TileA::PER_THREAD];
    const T* pb[TileB::PER_THREAD];
    TileA::init(pa, reinterpret_cast<const T*>(p.A), p.lda, m0, kbeg, p.M, tid);
    TileB::init(pb, reinterpret_cast<const T*>(p.B), p.ldb, n0, kbeg, p.N, tid);
    const long stepA = TileA::kstep(p.lda), stepB = TileB::kstep(p.ldb);

    // the partial tile (if any) is fetched first so that its latency hides behind the whole main loop
    u32x4_t ta[TileA::PER_THREAD], tb[TileB::PER_THREAD];
    if (krem > 0) {
        TileA::load_tail(ta, pa, stepA * nfull, p.lda, krem, tid);
        TileB::load_tail(tb, pb, stepB * nfull, p.ldb, krem, tid);
    }
    if (nfull > 0) {
        // Two named register stages (2 x BK contraction indices in flight).  No branch sits between a load and its
        // use: past the last tile the pointers simply stop advancing (re-reading an L2-resident tile that is never
        // consumed).  The compiler's counted s_waitcnt vmcnt(N) keeps the younger stage in flight while the older is
        // written to LDS; __syncthreads() does not drain register-destination loads.  (Named stages, not an array
        // of stages handed to a lambda: that form was demoted to scratch memory by hipcc.)
        u32x4_t ra0[TileA::PER_THREAD], rb0[TileB::PER_THREAD], ra1[TileA::PER_THREAD], rb1[TileB::PER_THREAD];
        int issued = 0;
#define UNIVL_LOAD_STAGE(RA, RB)                                  \
        do {                                                      \
            TileA::load_fast(RA, pa);                             \
            TileB::load_fast(RB, pb);                             \
            ++issued;                                             \
            const long adv__ = issued < nfull ? 1 : 0;            \
            TileA::advance(pa, stepA * adv__);                    \
            TileB::advance(pb, stepB * adv__);                    \
        } while (0)
#define UNIVL_STEP(RA, RB, CUR, RELOAD)                           \
        do {                                                      \
            T* cA__ = sA + (CUR) * TileA::ELEMS;                  \
            T* cB__ = sB + (CUR) * TileB::ELEMS;                  \
            TileA::store_fast(RA, cA__, tid);                     \
            TileB::store_fast(RB, cB__, tid);                     \
            if (RELOAD) UNIVL_LOAD_STAGE(RA, RB);                 \
            __syncthreads();                                      \
            compute(cA__, cB__);                                  \
        } while (0)
        UNIVL_LOAD_STAGE(ra0, rb0);
        UNIVL_LOAD_STAGE(ra1, rb1);
        const int ngroups = nfull / 2;
        for (int grp = 0; grp < ngroups; ++grp) {
            UNIVL_STEP(ra0, rb0, 0, true);
            UNIVL_STEP(ra1, rb1, 1, true);
        }
        if (nfull & 1) UNIVL_STEP(ra0, rb0, 0, false);
#undef UNIVL_STEP
#undef UNIVL_LOAD_STAGE
    }
    if (krem > 0) {
        const int cur = nfull & 1;
        T* cA = sA + cur * TileA::ELEMS;
        T* cB = sB + cur * TileB::ELEMS;
        __syncthreads();          // every wave is done reading whichever buffer the tail overwrites
        TileA::store_tail(ta, cA, krem, tid);
        TileB::store_tail(tb, cB, krem, tid);
        __syncthreads();
        compute(cA, cB);
    }

    // ------------------------------------------------------------------------------------------ epilogue
    // All epilogue INPUTS (bias, residual, saved pre-activation, old C) are fetched first, in flag-uniform groups
    // of back-to-back loads from clamped (always valid) addresses; only the stores are predicated.  A per-element
    // "if (flag) load" would serialise 16 dependent round trips per thread.
    const bool atomic = (p.flags & UNIVL_GEMM_ATOMIC) != 0;
    const bool first_slice = (blockIdx.z == 0);
    T* C16 = reinterpret_cast<T*>(p.C16);
    T* aux = reinterpret_cast<T*>(p.aux);
    long orow[MI][4];
    int ocol[NI];
    bool vrow[MI][4], vcol[NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm0 + 16 * a + 4 * g + r;
            vrow[a][r] = row < p.M;
            orow[a][r] = (long)min(row, p.M - 1);
        }
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int col = n0 + wn0 + 16 * b + i;
        vcol[b] = col < p.N;
        ocol[b] = min(col, p.N - 1);
    }
    float ev[MI][NI][4];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ev[a][b][r] = acc[a][b][r] * p.alpha;
    if (p.bias && first_slice) {
        float bv[NI];
#pragma unroll
        for (int b = 0; b < NI; ++b) bv[b] = p.bias[ocol[b]];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[a][b][r] += bv[b];
    }
    if (p.R && first_slice) {
        float rv[MI][NI][4];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[a][b][r] = p.R[orow[a][r] * p.ldr + ocol[b]];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[a][b][r] += rv[a][b][r];
    }
    if (p.flags & UNIVL_GEMM_GELU_FWD) {
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (vrow[a][r] && vcol[b]) aux[orow[a][r] * p.ldaux + ocol[b]] = from_f32<T>(ev[a][b][r]);
                    ev[a][b][r] = gelu_f(ev[a][b][r]);
                }
    }
    if (p.flags & UNIVL_GEMM_GELU_BWD) {
        T uv[MI][NI][4];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) uv[a][b][r] = aux[orow[a][r] * p.ldaux + ocol[b]];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[a][b][r] *= gelu_grad_f(to_f32<T>(uv[a][b][r]));
    }
    if ((p.flags & UNIVL_GEMM_ACCUM) && !atomic) {
        float cv[MI][NI][4];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[a][b][r] = p.C32[orow[a][r] * p.ldc + ocol[b]];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[a][b][r] += cv[a][b][r];
    }
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!(vrow[a][r] && vcol[b])) continue;
                const long o = orow[a][r] * p.ldc + ocol[b];
                if (atomic) {
                    unsafeAtomicAdd(p.C32 + o, ev[a][b][r]);
                } else {
                    if (p.C32) p.C32[o] = ev[a][b][r];
                    if (C16) C16[o] = from_f32<T>(ev[a][b][r]);
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

template <typename T, bool TA, bool TB, int BM, int BN, int D, int NC>
int launch(const GemmArgs& a, int ksplit, hipStream_t stream) {
    constexpr int BK = NC * Mma<T>::CH;
    using TileA = Tile<T, TA, BM, BK>;
    using TileB = Tile<T, TB, BN, BK>;
    const size_t smem = 2 * (TileA::ELEMS + TileB::ELEMS) * sizeof(T);
    static bool attr_done = false;   // per instantiation
    if (!attr_done && smem > 48 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, TA, TB, BM, BN, D, NC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, ksplit);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, BM, BN, D, NC>), grid, dim3(256), smem, stream, a);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

template <typename T, int BM, int BN, int D, int NC>
int dispatch_trans(const GemmArgs& a, int ta, int tb, int ksplit, hipStream_t s) {
    if (!ta && !tb) return launch<T, false, false, BM, BN, D, NC>(a, ksplit, s);
    if (!ta && tb) return launch<T, false, true, BM, BN, D, NC>(a, ksplit, s);
    if (ta && tb) return launch<T, true, true, BM, BN, D, NC>(a, ksplit, s);
    return launch<T, true, false, BM, BN, D, NC>(a, ksplit, s);
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
    // tile choice: 128x128 once the grid fills the chip, else 64x64 for parallelism.  The small tile stages
    // 4 chunks (128 bf16 / 64 f32) per barrier: at M <= a few hundred the kernel is a latency chain of K steps
    // (ds_write -> barrier -> ds_read -> MFMA), so fewer, deeper steps win; 2 stages of 32 KB stay in flight.
    const long tiles128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
    const bool big = d->tile == 128 || (d->tile == 0 && tiles128 >= 384);
    static const int small_nc = [] { const char* e = getenv("UNIVL_GEMM_SMALL_NC"); return (e && e[0] == '2') ? 2 : 4; }();
    const int nc = big ? 2 : small_nc;
    int ksplit = d->ksplit < 1 ? 1 : d->ksplit;
    const int BK = (d->dtype == UNIVL_BF16 ? 32 : 16) * nc;
    int klen = ((d->K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    ksplit = (d->K + klen - 1) / klen;
    if (ksplit > 1) {
        UNIVL_CHECK_ARG(d->C32 && !d->C16 && !(flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD)), UNIVL_EINVAL,
                        "univl_gemm: split-K needs a pre-zeroed fp32 output and a linear epilogue");
    }
    UNIVL_CHECK_ARG(!(d->dbias && !d->trans_a), UNIVL_EINVAL, "univl_gemm: dbias only with T-major A (wgrad)");
    UNIVL_CHECK_ARG(!((flags & UNIVL_GEMM_ACCUM) && !d->C32), UNIVL_EINVAL, "univl_gemm: ACCUM needs the fp32 output");
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.lda = d->lda; a.ldb = d->ldb; a.M = d->M; a.N = d->N; a.K = d->K;
    a.C32 = d->C32; a.C16 = d->C16; a.ldc = d->ldc; a.bias = d->bias; a.R = d->R; a.ldr = d->ldr;
    a.aux = d->aux; a.ldaux = d->ldaux; a.dbias = d->dbias; a.alpha = d->alpha;
    a.flags = flags | (ksplit > 1 ? UNIVL_GEMM_ATOMIC : 0);
    a.ksplit_len = klen;
    if (d->dtype == UNIVL_BF16) {
        if (big) return dispatch_trans<__bf16, 128, 128, 2, 2>(a, d->trans_a, d->trans_b, ksplit, stream);
        return nc == 4 ? dispatch_trans<__bf16, 64, 64, 2, 4>(a, d->trans_a, d->trans_b, ksplit, stream)
                       : dispatch_trans<__bf16, 64, 64, 2, 2>(a, d->trans_a, d->trans_b, ksplit, stream);
    }
    if (big) return dispatch_trans<float, 128, 128, 2, 2>(a, d->trans_a, d->trans_b, ksplit, stream);
    return nc == 4 ? dispatch_trans<float, 64, 64, 2, 4>(a, d->trans_a, d->trans_b, ksplit, stream)
                   : dispatch_trans<float, 64, 64, 2, 2>(a, d->trans_a, d->trans_b, ksplit, stream);
}
