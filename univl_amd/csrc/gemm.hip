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
    // The load is UNCONDITIONAL (out-of-range pieces read a clamped, valid address): a branch around a load makes
    // hipcc wait vmcnt(0) right behind it and serialises every load of the kernel.  Masking happens in store().
    __device__ static __forceinline__ void load(uint4 (&r)[PER_THREAD], const T* base, long ld, int row0, int k0,
                                                int rows_total, int k_end, int tid) {
        const int rmax = TR ? ((rows_total - 1) / EPC) * EPC : rows_total - 1;
        const int kmax = TR ? k_end - 1 : ((k_end - 1) / EPC) * EPC;
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            const int ch = tid + 256 * c;
            int row, kk;
            if (TR) { kk = ch / (ROWS / EPC); row = (ch % (ROWS / EPC)) * EPC; }
            else    { row = ch / (BK / EPC);  kk = (ch % (BK / EPC)) * EPC; }
            const int gr = min(row0 + row, rmax), gk = min(k0 + kk, kmax);
            const T* p = TR ? (base + (long)gk * ld + gr) : (base + (long)gr * ld + gk);
            r[c] = *reinterpret_cast<const uint4*>(p);
        }
    }
    // registers -> LDS, zeroing whatever lies outside [rows_total) x [k_end) (same tile coordinates as load()).
    __device__ static __forceinline__ void store(const uint4 (&r)[PER_THREAD], T* lds, int row0, int k0,
                                                 int rows_total, int k_end, int tid) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            const int ch = tid + 256 * c;
            int row, kk;
            if (TR) { kk = ch / (ROWS / EPC); row = (ch % (ROWS / EPC)) * EPC; }
            else    { row = ch / (BK / EPC);  kk = (ch % (BK / EPC)) * EPC; }
            const int off = TR ? kk * PITCH + row : row * PITCH + kk;
            const int gr = row0 + row, gk = k0 + kk;
            uint4 v = r[c];
            const int lim = (gr < rows_total && gk < k_end) ? (TR ? (rows_total - gr) : (k_end - gk)) : 0;
            if (lim < EPC) {
                // whole-dword AND masks (no per-element extraction of the loaded data: the compiler would hoist
                // that above the loop and wait for every prologue load)
                constexpr int EPD = 4 / (int)sizeof(T);      // elements per dword
                unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned m = 0u;
                    if (EPD == 1) m = (d < lim) ? 0xFFFFFFFFu : 0u;
                    else m = ((2 * d < lim) ? 0x0000FFFFu : 0u) | ((2 * d + 1 < lim) ? 0xFFFF0000u : 0u);
                    w[d] &= m;
                }
            }
            *reinterpret_cast<uint4*>(lds + off) = v;
        }
    }
    // fragment for the 16 tile rows starting at `r16`, chunk `c` (contraction offset c*CH)
    __device__ static __forceinline__ typename Mma<T>::frag frag(const T* lds, int r16, int c, int lane) {
        if (TR) return Mma<T>::lds_tmajor(lds + (c * Mma<T>::CH) * PITCH + r16, PITCH, lane);
        return Mma<T>::lds_kmajor(lds + (r16 + (lane & 15)) * PITCH + c * Mma<T>::CH, lane >> 4);
    }
};

template <typename T, bool TA, bool TB, int BM, int BN, int D>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int CH = Mma<T>::CH;
    constexpr int BK = 2 * CH;
    using TileA = Tile<T, TA, BM, BK>;
    using TileB = Tile<T, TB, BN, BK>;
    constexpr int WM = BM / 2, WN = BN / 2;      // per-wave sub-tile
    constexpr int MI = WM / 16, NI = WN / 16;
    static_assert(D % 2 == 0, "ring depth must be even (LDS double buffer index is then compile-time)");

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

    // Register ring of D K-tiles in flight: at small M every workgroup streams its own slice of the weights and
    // the kernel is bound by HBM LATENCY, not bandwidth -- the fix is bytes in flight (D x 16 KB per workgroup).
    // The compiler's counted s_waitcnt vmcnt(N) (loads retire in order) keeps D-1 tiles in flight while the
    // oldest is written to LDS; a plain __syncthreads() does not drain register-destination loads.
    uint4 ra[D][TileA::PER_THREAD], rb[D][TileB::PER_THREAD];
    const bool want_dbias = TA && (p.dbias != nullptr) && (blockIdx.x == 0);
    float dbias_acc = 0.0f;

    // prologue: D tiles in flight (tile index clamped: a short contraction re-reads its last tile from L2 rather
    // than branching around the load -- any branch/PHI around these loads makes hipcc copy the ring registers
    // and wait for the data right behind the load)
#pragma unroll
    for (int s = 0; s < D; ++s) {
        const int t = min(s, nk - 1);
        TileA::load(ra[s], A, p.lda, m0, kbeg + t * BK, p.M, kend, tid);
        TileB::load(rb[s], B, p.ldb, n0, kbeg + t * BK, p.N, kend, tid);
    }
    auto step = [&](uint4 (&qa)[TileA::PER_THREAD], uint4 (&qb)[TileB::PER_THREAD], int kt, int cur, bool reload) {
        T* cA = sA + cur * TileA::ELEMS;
        T* cB = sB + cur * TileB::ELEMS;
        TileA::store(qa, cA, m0, kbeg + kt * BK, p.M, kend, tid);
        TileB::store(qb, cB, n0, kbeg + kt * BK, p.N, kend, tid);
        if (reload) {
            const int t = min(kt + D, nk - 1);
            TileA::load(qa, A, p.lda, m0, kbeg + t * BK, p.M, kend, tid);
            TileB::load(qb, B, p.ldb, n0, kbeg + t * BK, p.N, kend, tid);
        }
        __syncthreads();
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
    };
    // steady state: whole groups of D steps, every step refills its ring slot (no conditionals around loads)
    const int ngroups = nk / D;
    for (int grp = 0; grp < ngroups; ++grp) {
#pragma unroll
        for (int s = 0; s < D; ++s) step(ra[s], rb[s], grp * D + s, s & 1, true);
    }
    // tail: fewer than D steps left, nothing more to fetch
    const int rem = nk - ngroups * D;
#pragma unroll
    for (int s = 0; s < D - 1; ++s) {
        if (s < rem) step(ra[s], rb[s], ngroups * D + s, s & 1, false);
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

template <typename T, bool TA, bool TB, int BM, int BN, int D>
int launch(const GemmArgs& a, int ksplit, hipStream_t stream) {
    constexpr int BK = 2 * Mma<T>::CH;
    using TileA = Tile<T, TA, BM, BK>;
    using TileB = Tile<T, TB, BN, BK>;
    const size_t smem = 2 * (TileA::ELEMS + TileB::ELEMS) * sizeof(T);
    static bool attr_done = false;   // per instantiation
    if (!attr_done && smem > 48 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, TA, TB, BM, BN, D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, ksplit);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, BM, BN, D>), grid, dim3(256), smem, stream, a);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

template <typename T, int BM, int BN, int D>
int dispatch_trans(const GemmArgs& a, int ta, int tb, int ksplit, hipStream_t s) {
    if (!ta && !tb) return launch<T, false, false, BM, BN, D>(a, ksplit, s);
    if (!ta && tb) return launch<T, false, true, BM, BN, D>(a, ksplit, s);
    if (ta && tb) return launch<T, true, true, BM, BN, D>(a, ksplit, s);
    return launch<T, true, false, BM, BN, D>(a, ksplit, s);
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
    UNIVL_CHECK_ARG(!((flags & UNIVL_GEMM_ACCUM) && !d->C32), UNIVL_EINVAL, "univl_gemm: ACCUM needs the fp32 output");
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
        return big ? dispatch_trans<__bf16, 128, 128, 2>(a, d->trans_a, d->trans_b, ksplit, stream)
                   : dispatch_trans<__bf16, 64, 64, 4>(a, d->trans_a, d->trans_b, ksplit, stream);
    }
    return big ? dispatch_trans<float, 128, 128, 2>(a, d->trans_a, d->trans_b, ksplit, stream)
               : dispatch_trans<float, 64, 64, 4>(a, d->trans_a, d->trans_b, ksplit, stream);
}
