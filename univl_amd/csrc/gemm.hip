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
// Workgroup = 4 waves (2x2), tile BM x BN in {64x64, 128x128}, BK = 2 or 4 chunks; global -> LDS by DMA
// (global_load_lds_dwordx4) into two XOR-swizzled, unpadded stages, one barrier per K step.  Optional split-K over gridDim.z accumulates with fp32 atomics into a pre-zeroed C.
//
// Besides the single and the grouped launch there are two "carrier" launches in which a latency-bound product shares its launch
// with throughput work that nothing downstream waits for, as extra workgroups behind its own tiles: gemm_pair_kernel (a dgrad
// product + the weight-gradient product of the same nn.Linear, default) and gemm_adam_kernel (a forward product + BertAdam chunks
// of the next layer, experimental).
//
// Epilogue (all optional, in this order): *alpha, +bias[n], +residual[m,n] (fp32), erf-GELU forward (saving the
// pre-activation), *gelu'(saved pre-activation), +C_old (accumulate), store fp32 and/or T.  A wgrad launch can
// also emit the bias gradient (row sums of A_op over the contraction) from the tiles it already staged.
#include <stdlib.h>
#include "common.h"
#include "univl_hip.h"
#include "adam_body.h"
#include "ln_body.h"

// -DUNIVL_TRACE (a second, measurement-only build: univl_amd/build.py trace=True -> lib/libunivl_hip_trace.so, never the product library): thread 0
// of every workgroup of a GEMM-family kernel writes the device wall clock (100 MHz) at kernel entry, when its first staged tile has
// landed, after its K loop and after its epilogue into a buffer registered with univl_trace_set -- where the ~9 us of a 192-row
// product go (scripts/mb_trace_gemm.py), without a profiler attached.
#ifdef UNIVL_TRACE
__device__ unsigned long long* g_univl_trace = nullptr;
__device__ int g_univl_trace_cap = 0;
#define UNIVL_TRACE_AT(phase)                                                                              \
    do {                                                                                                   \
        if (threadIdx.x == 0 && g_univl_trace != nullptr) {                                                \
            const int w__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                \
            if (w__ < g_univl_trace_cap) g_univl_trace[w__ * 4 + (phase)] = wall_clock64();                \
        }                                                                                                  \
    } while (0)
extern "C" int univl_trace_set(unsigned long long* buf, int cap_workgroups) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_univl_trace), &buf, sizeof(buf)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_univl_trace_cap), &cap_workgroups, sizeof(int)) != hipSuccess) return 1;
    return 0;
}
#else
#define UNIVL_TRACE_AT(phase)
#endif

// UNIVL_GELU_FAST=1 (A/B build: univl_amd/build.py --variant fastgelu -DUNIVL_GELU_FAST=1): the branch-free erf-GELU of the 256 body in
// gemm_tile's bf16 GELU / GELU' epilogues too.  Measured -0.8 % per step at 4 / 16 pairs (profiles/r06q_ab_gelu.txt) -- and NOT taken: any
// change of the FFN arithmetic re-draws the bf16 rounding noise of the whole backward, and the one parity statistic that sits at
// north_star's 1e-2 (the global gradient error, 72 % one tensor: DESIGN.md section 2) moved from 0.977e-2 to 1.124e-2 in the deterministic
// mode with medians unchanged (profiles/r06_final3_parity_errors.json vs r05_final2): 0.8 % is not worth a different draw of a validated result.
#ifndef UNIVL_GELU_FAST
#define UNIVL_GELU_FAST 0
#endif

namespace {

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

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
    float* sumsq; int sumsq_rows, sumsq_stride;
    int gm;           // row tiles per L2 block of the tile order (0: plain order)
    const void* A_lo; const void* B_lo; void* C16_lo;      // operand pairs (UnivlGemm): the contraction is walked once per term
};

// One operand tile in LDS: UNPADDED rows of RB = 128 or 256 bytes, filled by direct global->LDS DMA
// (global_load_lds_dwordx4: each wave instruction drops 64 x 16 B at wave-uniform base + lane*16, no VGPR round trip,
// no ds_write -- at one workgroup per CU the ds_write pass of a register-staged tile was ~1/3 of every K step).
// Since the DMA destination is lane-linear, bank conflicts are removed by an XOR swizzle of the 16-byte piece index
// applied to the per-lane SOURCE address and, identically, to every fragment read (an involution):
//      piece (row r, q)  lives at byte  r*RB + ((q ^ swz(r)) << 4)
// K-major tile ([ROWS][BK], fragment = two 8-byte reads per lane over 16 consecutive rows): swz = r & 15 (RB 256) or
// (r >> 1) & 7 (RB 128).  T-major tile ([BK][ROWS], transpose-read of 4 k-rows x 32 B per 16 lanes): swz = (r & 7) << 1
// (RB >= 256) or r & 6 (RB 128).
template <typename T, bool TR, int ROWS, int BK, int NT = 256> struct Tile {
    static constexpr int EPC = Mma<T>::EPC;
    static constexpr int RB = (TR ? ROWS : BK) * (int)sizeof(T);     // bytes per LDS row
    static constexpr int NR = TR ? BK : ROWS;                        // LDS rows
    static constexpr int CPR = RB / 16;                              // 16-byte pieces per row
    static constexpr int BYTES = NR * RB;
    static constexpr int CHUNKS = BYTES / 16;
    static constexpr int PER_THREAD = CHUNKS / NT;                   // NT = threads of the workgroup
    static_assert(CHUNKS % NT == 0 && (CPR == 8 || CPR == 16 || CPR == 32), "unsupported tile geometry");

    __device__ static __forceinline__ int swz(int r) {
        if (TR) return CPR == 8 ? (r & 6) : ((r & 7) << 1);
        return CPR == 8 ? ((r >> 1) & 7) : (r & 15);
    }
    __device__ static __forceinline__ int byte_of(int r, int q) { return r * RB + ((q ^ swz(r)) << 4); }
    // LDS piece L (lane-linear) holds tile piece (r, q):  r = L / CPR,  q = (L % CPR) ^ swz(r)
    __device__ static __forceinline__ void coords(int c, int tid, int& r, int& q) {
        const int L = tid + NT * c;
        r = L / CPR;
        q = (L % CPR) ^ swz(r);
    }
    // Per-thread source pointers.  Everything invariant along K (row clamp, piece coordinates) is folded in once;
    // rows beyond rows_total are CLAMPED, not masked: they only feed accumulator rows/columns the epilogue never
    // stores.  Only the contraction direction needs zero fill, and only in the last, partial K tile.
    __device__ static __forceinline__ void init(const T* (&P)[PER_THREAD], const T* base, long ld, int row0, int k0,
                                                int rows_total, int tid) {
        const int rmax = TR ? ((rows_total - 1) / EPC) * EPC : rows_total - 1;
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int r, q;
            coords(c, tid, r, q);
            if (TR) P[c] = base + (long)(k0 + r) * ld + min(row0 + q * EPC, rmax);
            else    P[c] = base + (long)min(row0 + r, rmax) * ld + (k0 + q * EPC);
        }
    }
    __device__ static __forceinline__ long kstep(long ld) { return TR ? (long)BK * ld : (long)BK; }
    __device__ static __forceinline__ void advance(const T* (&P)[PER_THREAD], long step) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) P[c] += step;
    }
    // asynchronous fill of one LDS stage.  AUX = 2: non-temporal hint (measurement build only: UNIVL_GEMM_NT_B, scripts/mb_trace_gemm.py)
    template <int AUX = 0>
    __device__ static __forceinline__ void issue(const T* const (&P)[PER_THREAD], unsigned char* stage, int tid) {
        unsigned char* wbase = stage + (tid & ~63) * 16;          // wave-uniform
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)P[c],
                                             (__attribute__((address_space(3))) void*)(wbase + c * (NT * 16)), 16, 0, AUX);
    }
    // last, partial K tile (krem < BK contraction indices left; P + skip points at the tile start): through
    // registers, clamped addresses, zero fill of everything at or beyond krem
    __device__ static __forceinline__ void load_tail(u32x4_t (&r)[PER_THREAD], const T* const (&P)[PER_THREAD], long skip,
                                                     long ld, int krem, int tid) {
        const int klast = TR ? krem - 1 : ((krem - 1) / EPC) * EPC;
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int rr, q;
            coords(c, tid, rr, q);
            const int kk = TR ? rr : q * EPC;
            const long back = (long)(min(kk, klast) - kk) * (TR ? ld : 1);
            r[c] = *reinterpret_cast<const u32x4_t*>(P[c] + skip + back);
        }
    }
    __device__ static __forceinline__ void store_tail(const u32x4_t (&r)[PER_THREAD], unsigned char* stage, int krem, int tid) {
#pragma unroll
        for (int c = 0; c < PER_THREAD; ++c) {
            int rr, q;
            coords(c, tid, rr, q);
            const int kk = TR ? rr : q * EPC;
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
            *reinterpret_cast<u32x4_t*>(stage + (tid + NT * c) * 16) = v;
        }
    }
    // element (row r, column e) of the tile, for scalar access
    __device__ static __forceinline__ const T* elem(const unsigned char* stage, int r, int e) {
        const int b = e * (int)sizeof(T);
        return reinterpret_cast<const T*>(stage + byte_of(r, b >> 4) + (b & 15));
    }
    // T-major bf16 fragment as two INLINE-ASM transpose reads (lo: k-slots 4g .. 4g+3, hi: 16+4g .. 16+4g+3 of the chunk).  Round 5: for
    // the builtin (Mma::tr_pair) hipcc cannot prove that the read does not alias the LDS-DMA of the NEXT tile, which gemm_tile has
    // just issued, and emits `s_waitcnt vmcnt(0)` in front of the first transpose read of every K step -- the double buffer of every
    // dgrad / weight-gradient product was a single buffer (DMA round trip + multiply per step, never overlapped).  hipcc does not
    // count an asm load: the caller waits (lgkmcnt(0) naming every destination) before it uses lo / hi (guide 5.7 item 1, form ii).
    __device__ static __forceinline__ void frag_tr(const unsigned char* stage, int r16, int c, int lane, u32x2_t& lo, u32x2_t& hi) {
        static_assert(TR && sizeof(T) == 2, "transpose reads: T-major bf16 tiles");
        typedef __attribute__((address_space(3))) unsigned char lds_u8;
        const int g = lane >> 4, i = lane & 15;
        const int krow = c * 32 + 4 * g + (i >> 2);
        const int cb = (r16 + 4 * (i & 3)) * 2;
        const unsigned a0 = (unsigned)(size_t)(lds_u8*)(stage + byte_of(krow, cb >> 4) + (cb & 15));
        const unsigned a1 = (unsigned)(size_t)(lds_u8*)(stage + byte_of(krow + 16, cb >> 4) + (cb & 15));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    }
    // fragment for the 16 tile rows starting at `r16`, chunk `c` (contraction offset c*CH)
    __device__ static __forceinline__ typename Mma<T>::frag frag(const unsigned char* stage, int r16, int c, int lane) {
        const int g = lane >> 4, i = lane & 15;
        typename Mma<T>::frag f;
        if (!TR) {
            const int row = r16 + i;
            if (sizeof(T) == 2) {
                const int q = c * 4 + (g >> 1), w = (g & 1) * 8;
                // The two row tiles of a wave sit a constant 4096 B apart and hipcc fuses their reads into
                // ds_read2st64_b64 -- which the LDS serves at 1/4 of the ds_read_b64 rate (32-bank mode, 16-lane
                // groups: SQ_LDS_BANK_CONFLICT showed 8 extra cycles per instruction).  Opaque offsets keep them apart.
                int o0 = byte_of(row, q) + w, o1 = byte_of(row, q + 2) + w;
                asm volatile("" : "+v"(o0), "+v"(o1));
                const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(stage + o0);
                const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(stage + o1);
                return Mma<T>::pack(lo, hi);
            } else {
                return Mma<T>::pack16(stage + byte_of(row, c * 4 + g));
            }
        } else {
            if (sizeof(T) == 2) {
                const int krow = c * 32 + 4 * g + (i >> 2);
                const int cb = (r16 + 4 * (i & 3)) * 2;
                return Mma<T>::tr_pair(stage + byte_of(krow, cb >> 4) + (cb & 15), stage + byte_of(krow + 16, cb >> 4) + (cb & 15));
            } else {
                const int cb = (r16 + i) * 4;
                const int k0 = c * 16 + 4 * g;
                return Mma<T>::gather4(stage + byte_of(k0, cb >> 4) + (cb & 15), stage + byte_of(k0 + 1, cb >> 4) + (cb & 15),
                                       stage + byte_of(k0 + 2, cb >> 4) + (cb & 15), stage + byte_of(k0 + 3, cb >> 4) + (cb & 15));
            }
        }
        return f;
    }
};

// The multiply of one staged K step: NC chunks, MI x NI accumulator tiles of this wave (sub-tile origin wm0, wn0).
template <typename T, bool TA, bool TB, int MI, int NI, int NC, typename TileA, typename TileB>
__device__ __forceinline__ void tile_mma(const unsigned char* cA, const unsigned char* cB, const int wm0, const int wn0, const int lane,
                                         f32x4_t (&acc)[MI][NI]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        typename Mma<T>::frag fa[MI], fb[NI];
        if constexpr (sizeof(T) == 2 && (TA || TB)) {
            // bf16 with a T-major operand: its transpose reads go out as inline asm (Tile::frag_tr), all of the chunk first, then
            // ONE wait that names every destination, then the fragments are put together and multiplied
            u32x2_t al[MI], ah[MI], bl[NI], bh[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) {
                if constexpr (TA) TileA::frag_tr(cA, wm0 + 16 * a, c, lane, al[a], ah[a]);
                else fa[a] = TileA::frag(cA, wm0 + 16 * a, c, lane);
            }
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                if constexpr (TB) TileB::frag_tr(cB, wn0 + 16 * b, c, lane, bl[b], bh[b]);
                else fb[b] = TileB::frag(cB, wn0 + 16 * b, c, lane);
            }
#pragma unroll
            for (int a = 0; a < MI; ++a) { if constexpr (TA) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[a]), "+v"(ah[a])); }
#pragma unroll
            for (int b = 0; b < NI; ++b) { if constexpr (TB) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bl[b]), "+v"(bh[b])); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MI; ++a) { if constexpr (TA) fa[a] = __builtin_bit_cast(typename Mma<T>::frag, u32x4_t{al[a][0], al[a][1], ah[a][0], ah[a][1]}); }
#pragma unroll
            for (int b = 0; b < NI; ++b) { if constexpr (TB) fb[b] = __builtin_bit_cast(typename Mma<T>::frag, u32x4_t{bl[b][0], bl[b][1], bh[b][0], bh[b][1]}); }
        } else {
#pragma unroll
            for (int a = 0; a < MI; ++a) fa[a] = TileA::frag(cA, wm0 + 16 * a, c, lane);
#pragma unroll
            for (int b = 0; b < NI; ++b) fb[b] = TileB::frag(cB, wn0 + 16 * b, c, lane);
        }
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = Mma<T>::mma(fa[a], fb[b], acc[a][b]);
    }
}

// Two LDS stages: tile t+1 streams in while tile t is multiplied (one wait + barrier per K step).  WGM x WGN waves; every wave owns a
// (BM / WGM) x (BN / WGN) sub-tile.  (Measured and removed in round 4, numbers in DESIGN.md section 8: a three-stage ring with counted
// waits, a 256 x 128 tile on that ring, a "burst" form with the whole contraction slice in one DMA burst.)
// PAIRS: the instantiation carries operand pairs (UnivlGemm.A_lo / B_lo / C16_lo).  Only the bf16 64 x 64 tile with 128-deep K steps does
// (the tile of every product up to 768 tokens, where the plans pair operands; `prepare` forces it for a paired descriptor): every other
// instantiation compiles to exactly the code it had before pairs existed (the 64 x 128 rider kernel spilled two more VGPRs with the
// term walk compiled in: +0.6 .. +1.4 % per step at 32 - 128 pairs, profiles/r06_like_for_like_r05_vs_r06.txt).
template <typename T, bool TA, bool TB, int BM, int BN, int NC, int WGM = 2, int WGN = 2, bool PAIRS = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, const int bx, const int by, const int bz, const int nz) {
    constexpr int CH = Mma<T>::CH;
    constexpr int BK = NC * CH;          // contraction depth of one LDS stage (NC chunks of CH)
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    using TileA = Tile<T, TA, BM, BK, NT>;
    using TileB = Tile<T, TB, BN, BK, NT>;
    constexpr int WM = BM / WGM, WN = BN / WGN;      // per-wave sub-tile
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int D = NC == 6 ? 1 : 2;      // NC == 6: the single 192-deep K step of a weight gradient at 192 tokens -- one stage
    static_assert(BM <= NT, "the bias-gradient pass uses one thread per tile row");

    // ONE dynamic LDS object, declared once at namespace scope (a second __shared__ object makes hipcc drain vmcnt(0) before every ds_read)
    unsigned char* sA = smem_raw;
    unsigned char* sB = sA + D * TileA::BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int m0 = by * BM, n0 = bx * BN;
    // Operand pairs (UnivlGemm.A_lo / B_lo): the slices divide the CONCATENATED contraction  A.B | A.B_lo | A_lo.B  (nseg terms of K
    // each; K and the slice length are multiples of BK then, host).  A workgroup whose slice crosses a term boundary re-points its
    // DMA source pointers there (`wrap`, in K tiles from the slice start) and keeps its two-stage pipeline running across it.
    const int nseg = PAIRS ? 1 + (p.B_lo != nullptr ? 1 : 0) + (p.A_lo != nullptr ? 1 : 0) : 1;
    int kbeg = bz * p.ksplit_len;
    int kspan, seg = 0, wrap = 0x7fffffff;
    if (nseg > 1) {
        kspan = min(nseg * p.K, kbeg + p.ksplit_len) - kbeg;
        seg = kbeg / p.K;
        kbeg -= seg * p.K;
        wrap = (p.K - kbeg) / BK;
    } else {
        kspan = min(p.K, kbeg + p.ksplit_len) - kbeg;
    }
    const int nfull = kspan / BK;                      // full K tiles: no masking at all
    const int krem = kspan - nfull * BK;               // > 0: one partial tile at the end (never with operand pairs)
    auto seg_a = [&](int s) { return reinterpret_cast<const T*>((p.A_lo != nullptr && s > 0 && s == nseg - 1) ? p.A_lo : p.A); };
    auto seg_b = [&](int s) { return reinterpret_cast<const T*>((p.B_lo != nullptr && s == 1) ? p.B_lo : p.B); };

    UNIVL_TRACE_AT(0);
    f32x4_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Epilogue INPUTS that exist before the launch (bias; for sub-tiles of at most two accumulator tiles also the fp32 residual and the
    // saved pre-activation of the GELU' epilogue) are requested HERE, in front of the first staging round trip, from clamped (always
    // valid) addresses: they land while the first tile is in flight.  Fetched after the K loop they cost the epilogue one more memory
    // round trip (phase trace, profiles/r04b_trace_gemm_phases.txt: 1.4 us of epilogue behind a 4 us K loop at 192 rows).
    const bool first_slice = (bz == 0);
    constexpr bool PRE = (MI * NI <= 2);              // the 64 x 64 tile on 8 waves: 12 more live registers; larger sub-tiles would spill
    constexpr int PM = PRE ? MI : 1;
    float bv[NI];
    float rpre[PM][NI][4];
    float upre[PM][NI][4];
    {
        int pcol[NI];
#pragma unroll
        for (int b = 0; b < NI; ++b) pcol[b] = min(n0 + wn0 + 16 * b + i, p.N - 1);
        if constexpr (PRE) {
            if (p.bias && first_slice) {
#pragma unroll
                for (int b = 0; b < NI; ++b) bv[b] = p.bias[pcol[b]];
            } else {
#pragma unroll
                for (int b = 0; b < NI; ++b) bv[b] = 0.0f;
            }
            const T* auxp = reinterpret_cast<const T*>(p.aux);
            const float* auxf = reinterpret_cast<const float*>(p.aux);
            const bool aux32 = (p.flags & UNIVL_GEMM_AUX_F32) != 0;
            if (p.R && first_slice) {
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int b = 0; b < NI; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            rpre[a][b][r] = p.R[(long)min(m0 + wm0 + 16 * a + 4 * g + r, p.M - 1) * p.ldr + pcol[b]];
            }
            if (p.flags & UNIVL_GEMM_GELU_BWD) {
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int b = 0; b < NI; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                            const long o = (long)min(m0 + wm0 + 16 * a + 4 * g + r, p.M - 1) * p.ldaux + pcol[b];
                            upre[a][b][r] = aux32 ? auxf[o] : to_f32<T>(auxp[o]);
                        }
            }
        }
    }

    const bool want_dbias = TA && (p.dbias != nullptr) && (bx == 0);
    float dbias_acc = 0.0f;

    auto compute = [&](const unsigned char* cA, const unsigned char* cB) {
        tile_mma<T, TA, TB, MI, NI, NC, TileA, TileB>(cA, cB, wm0, wn0, lane, acc);
        if (want_dbias) {
            // bias gradient = sum over the contraction (tokens) of A_op rows; A is T-major: [BK][BM]
            if (tid < BM) {
#pragma unroll 8
                for (int kk = 0; kk < BK; ++kk) dbias_acc += to_f32<T>(*TileA::elem(cA, kk, tid));
            }
        }
    };

    const T* pa[TileA::PER_THREAD];
    const T* pb[TileB::PER_THREAD];
    TileA::init(pa, seg_a(seg), p.lda, m0, kbeg, p.M, tid);
    TileB::init(pb, seg_b(seg), p.ldb, n0, kbeg, p.N, tid);
    const long stepA = TileA::kstep(p.lda), stepB = TileB::kstep(p.ldb);
    auto next_term = [&]() {        // the pointers stand at contraction index K of term `seg`: index 0 of term seg + 1
        TileA::advance(pa, (seg_a(seg + 1) - seg_a(seg)) - (long)p.K * (TA ? p.lda : 1));
        TileB::advance(pb, (seg_b(seg + 1) - seg_b(seg)) - (long)p.K * (TB ? p.ldb : 1));
        ++seg;
        wrap += p.K / BK;
    };

    // the partial tile (if any) goes through registers; fetched first so its latency hides behind the main loop
    u32x4_t ta[TileA::PER_THREAD], tb[TileB::PER_THREAD];
    if (krem > 0) {
        TileA::load_tail(ta, pa, stepA * nfull, p.lda, krem, tid);
        TileB::load_tail(tb, pb, stepB * nfull, p.ldb, krem, tid);
    }
#ifdef UNIVL_TRACE
    const bool nt_b = (p.flags & UNIVL_GEMM_PROBE_NT_B) != 0;
    auto issue_b = [&](unsigned char* st) { if (nt_b) TileB::template issue<2>(pb, st, tid); else TileB::template issue<0>(pb, st, tid); };
#else
    auto issue_b = [&](unsigned char* st) { TileB::issue(pb, st, tid); };
#endif
    if (nfull > 0) {
        // double-buffered LDS-DMA pipeline: tile t+1 streams into the other stage while tile t is multiplied; the
        // __syncthreads() at the end of a step carries the vmcnt(0) that makes tile t+1 visible and releases stage t.
        TileA::issue(pa, sA, tid);
        issue_b(sB);
        TileA::advance(pa, stepA);
        TileB::advance(pb, stepB);
        __syncthreads();
        UNIVL_TRACE_AT(1);
        for (int t = 0; t < nfull; ++t) {
            const int cur = t & 1;
            if (t + 1 < nfull) {
                if constexpr (PAIRS) { if (t + 1 == wrap) next_term(); }
                TileA::issue(pa, sA + (cur ^ 1) * TileA::BYTES, tid);
                issue_b(sB + (cur ^ 1) * TileB::BYTES);
                TileA::advance(pa, stepA);
                TileB::advance(pb, stepB);
            }
            compute(sA + cur * TileA::BYTES, sB + cur * TileB::BYTES);
            __syncthreads();
        }
    }
    if (krem > 0) {
        unsigned char* cA = sA + (nfull & 1) * TileA::BYTES;
        unsigned char* cB = sB + (nfull & 1) * TileB::BYTES;
        TileA::store_tail(ta, cA, krem, tid);
        TileB::store_tail(tb, cB, krem, tid);
        __syncthreads();
        compute(cA, cB);
    }

    UNIVL_TRACE_AT(2);
    // ------------------------------------------------------------------------------------------ epilogue
    // The remaining epilogue inputs (old C; residual / pre-activation of the large sub-tiles) are fetched in flag-uniform groups of
    // back-to-back loads from clamped (always valid) addresses; only the stores are predicated.  A per-element "if (flag) load"
    // would serialise 16 dependent round trips per thread.  One 16-row slab of the wave's sub-tile at a time: keeps the live epilogue
    // values to NI x 4 per input instead of MI x NI x 4 (the 128 x 128 tile would otherwise need > 256 VGPRs).
    const bool atomic = (p.flags & UNIVL_GEMM_ATOMIC) != 0;
    const bool nt_out = (p.flags & UNIVL_GEMM_NT_OUT) != 0;
    T* C16 = reinterpret_cast<T*>(p.C16);
    T* C16lo = PAIRS ? reinterpret_cast<T*>(p.C16_lo) : nullptr;
    T* aux = reinterpret_cast<T*>(p.aux);
    float* auxf32 = reinterpret_cast<float*>(p.aux);
    const bool aux_f32 = (p.flags & UNIVL_GEMM_AUX_F32) != 0;
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
    if constexpr (!PRE) {
        if (p.bias && first_slice) {
#pragma unroll
            for (int b = 0; b < NI; ++b) bv[b] = p.bias[ocol[b]];
        } else {
#pragma unroll
            for (int b = 0; b < NI; ++b) bv[b] = 0.0f;
        }
    }
    float ssq = 0.0f;
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        float ev[NI][4];
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ev[b][r] = acc[a][b][r] * p.alpha + bv[b];
        if (p.R && first_slice) {
            float rv[NI][4];
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (PRE) rv[b][r] = rpre[a][b][r];
                    else rv[b][r] = p.R[orow[a][r] * p.ldr + ocol[b]];
                }
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[b][r] += rv[b][r];
        }
        if (p.flags & UNIVL_GEMM_GELU_FWD) {
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (vrow[a][r] && vcol[b]) {
                        if (aux_f32) auxf32[orow[a][r] * p.ldaux + ocol[b]] = ev[b][r];
                        else aux[orow[a][r] * p.ldaux + ocol[b]] = from_f32<T>(ev[b][r]);
                    }
                    // (UNIVL_GELU_FAST: the branch-free form of common.h -- libm's erff is two divergent branches of ~35 VALU instructions,
                    // 1.2 us of the FFN1 epilogue at 192 rows; see the macro's comment for why the default keeps erff)
                    ev[b][r] = (sizeof(T) == 2 && UNIVL_GELU_FAST) ? g256_gelu(ev[b][r]) : gelu_f(ev[b][r]);
                }
        }
        if (p.flags & UNIVL_GEMM_GELU_BWD) {
            float uv[NI][4];
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (PRE) uv[b][r] = upre[a][b][r];
                    else uv[b][r] = aux_f32 ? auxf32[orow[a][r] * p.ldaux + ocol[b]] : to_f32<T>(aux[orow[a][r] * p.ldaux + ocol[b]]);
                }
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[b][r] *= (sizeof(T) == 2 && UNIVL_GELU_FAST) ? g256_gelu_grad(uv[b][r]) : gelu_grad_f(uv[b][r]);
        }
        if ((p.flags & UNIVL_GEMM_ACCUM) && !atomic) {
            float cv[NI][4];
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[b][r] = p.C32[orow[a][r] * p.ldc + ocol[b]];
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ev[b][r] += cv[b][r];
        }
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!(vrow[a][r] && vcol[b])) continue;
#ifdef UNIVL_TRACE
                if ((p.flags & UNIVL_GEMM_PROBE_NOSTORE) && ev[b][r] != 12345.678f) continue;      // measurement build only: no stores
#endif
                const long o = orow[a][r] * p.ldc + ocol[b];
                if (atomic) {
                    unsafeAtomicAdd(p.C32 + o, ev[b][r]);
                } else {
                    if (p.C32) { if (nt_out) __builtin_nontemporal_store(ev[b][r], p.C32 + o); else p.C32[o] = ev[b][r]; }
                    if (C16) C16[o] = from_f32<T>(ev[b][r]);
                    if constexpr (PAIRS) { if (C16lo) C16lo[o] = from_f32<T>(ev[b][r] - to_f32<T>(from_f32<T>(ev[b][r]))); }
                    ssq += ev[b][r] * ev[b][r];
                }
            }
    }
    if (p.sumsq) {                               // per-wave partial sums, plain stores (see univl_hip.h)
        ssq = wave_sum(ssq);
        const int tensor = p.sumsq_rows > 0 ? m0 / p.sumsq_rows : 0;
        const int mloc = p.sumsq_rows > 0 ? m0 % p.sumsq_rows : m0;
        const int nx = (p.N + BN - 1) / BN;
        if constexpr (NW == 8 && BM * BN == 64 * 64) {
            // the host reserves 4 slots per 64 x 64 tile (rows x N / 1024): waves w and w + 4 share one (p.sumsq is uniform)
            __syncthreads();                     // every wave is past its last fragment read: the stages are free
            float* red = reinterpret_cast<float*>(smem_raw);
            if (lane == 0) red[wave] = ssq;
            __syncthreads();
            if (lane == 0 && wave < 4) p.sumsq[(long)tensor * p.sumsq_stride + ((mloc / BM) * nx + bx) * 4 + wave] = red[wave] + red[wave + 4];
        } else {
            if (lane == 0) p.sumsq[(long)tensor * p.sumsq_stride + ((mloc / BM) * nx + bx) * NW + wave] = ssq;
        }
    }
    if (want_dbias && tid < BM) {
        const int row = m0 + tid;
        if (row < p.M) {
            if (nz > 1 || (p.flags & UNIVL_GEMM_DBIAS_ATOMIC)) unsafeAtomicAdd(p.dbias + row, dbias_acc);
            else if (p.flags & UNIVL_GEMM_ACCUM) p.dbias[row] += dbias_acc;
            else p.dbias[row] = dbias_acc;
        }
    }
    UNIVL_TRACE_AT(3);
}

// XCD-aware, L2-blocked workgroup -> tile map.  The dispatcher deals workgroups round-robin over the 8 XCDs (linear id % 8,
// x fastest), and every XCD has its own 4 MB L2: with the plain (x = column tile, y = row tile) grid the row tiles that share
// one WEIGHT tile land on different XCDs and each of them pulls that tile from HBM again (round-2 PMC pass: 2.9 GB of HBM
// traffic per step for 1.4 GB of algorithmic operand + output bytes).  Here every XCD takes a contiguous run of a tile LIST
// (the first total % 8 XCDs take one tile more), and the list is ordered so that the ~64 workgroups an XCD runs at a time
// form a compact block of the output:
//   * at most `gm` row tiles (small M): (column tile, k slice, row tile) with the row tile fastest -- the row tiles of one
//     weight tile sit behind one L2;
//   * more (M in the thousands): bands of `gm` row tiles, inside a band column by column -- a window of 64 consecutive tiles
//     is a gm x (64 / gm) block that reads gm + 64 / gm operand panels instead of 64 + 1 (at M = 6144 the activation operand
//     alone is 9-38 MB: with the row tile fastest over all 48 row tiles every panel is evicted before its next use).
// A bijection for any grid; only the placement changes, never the result.
__host__ __device__ __forceinline__ void tile_of(int t, int nx, int ny, int nz, int gm, int& bx, int& by, int& bz) {
    if (gm <= 0 || ny <= gm) {
        by = t % ny;
        const int u = t / ny;
        bz = u % nz;
        bx = u / nz;
    } else {
        const int per = nx * ny;
        bz = t / per;
        const int r = t - bz * per;
        const int band = r / (gm * nx), first = band * gm;
        const int rows = (ny - first) < gm ? (ny - first) : gm;
        const int rr = r - band * gm * nx;
        by = first + rr % rows;
        bx = rr / rows;
    }
}

__host__ __device__ __forceinline__ int xcd_run(int L, int total) {      // linear workgroup id -> position in the tile list
    const int c = L & 7, k = L >> 3;
    const int q = total >> 3, r = total & 7;
    return c * q + (c < r ? c : r) + k;
}

// the map of a plain (x, y, z) grid; grids below 16 workgroups and single-row-tile grids keep the plain order
__host__ __device__ __forceinline__ void xcd_tile_grid(int nx, int ny, int nz, int gm, int& bx, int& by, int& bz) {
    const int total = nx * ny * nz;
    if (total < 16 || ny == 1) return;
    tile_of(xcd_run(bx + nx * (by + ny * bz), total), nx, ny, nz, gm, bx, by, bz);
}

__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz, int gm) {
    xcd_tile_grid(gridDim.x, gridDim.y, gridDim.z, gm, bx, by, bz);
}

// (hipcc: the second launch bound is WAVES PER SIMD.)  The half-width tiles (128 x 64 / 64 x 128, 48 KB of LDS) are built for THREE
// resident workgroups per compute unit: 6 waves per SIMD (<= 80 VGPRs) on 8 waves, 3 on 4 waves.
template <typename T, bool TA, bool TB, int BM, int BN, int NC, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, (BM != BN && BM * BN == 128 * 64) ? (WGM * WGN == 8 ? 6 : 3) : 2) void gemm_kernel(GemmArgs p) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.flags & UNIVL_GEMM_XCD_MAP) xcd_tile(bx, by, bz, p.gm);
    constexpr bool PAIRS = sizeof(T) == 2 && !TA && BM == 64 && BN == 64 && NC == 4;      // forward / dgrad products on the small tile
    gemm_tile<T, TA, TB, BM, BN, NC, WGM, WGN, PAIRS>(p, bx, by, bz, gridDim.z);
}

#include "gemm256.h"

struct ColsumArgs {                  // out[c] (+)= sum over rows of x[r, c]: the bias gradient of a weight-gradient product whose A is x
    const __bf16* x; long ld; int rows, n; float* out; int n_tiles, n_pad;
};

// Bias-gradient role of the carrier launches (round 4): one workgroup sums 64 columns of the upstream gradient over ALL its rows, in a
// fixed order (8 column groups x NT / 8 row lanes, four 16-byte loads in flight per lane, the row lanes meet in LDS).  Before, the
// column-0 workgroups of the weight-gradient product walked their staged A tile element by element (64 ds_read_u16 per thread per K
// step): in the phase trace of the FFN1 pair at 192 tokens (profiles/r04b_trace_gemm_phases.txt) those 48 workgroups run 9.2 us
// against 3.8 us for the others and, dispatched behind the dgrad tiles, end 4.6 us after the FFN2 pair that carries no bias gradient.
template <int NT>
__device__ __forceinline__ void colsum_tile(const ColsumArgs& c, int tile, unsigned char* smem) {
    constexpr int RL = NT / 8;
    const int tid = threadIdx.x, cg = tid & 7, r0 = tid >> 3;
    const int c0 = tile * 64, col = c0 + cg * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    auto add = [&](const u32x4_t q) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc[2 * d] += __uint_as_float(q[d] << 16);
            acc[2 * d + 1] += __uint_as_float(q[d] & 0xFFFF0000u);
        }
    };
    if (col < c.n) {                                   // n % 8 == 0 (host)
        const __bf16* p = c.x + col;
        int r = r0;
        for (; r + 3 * RL < c.rows; r += 4 * RL) {
            const u32x4_t q0 = *reinterpret_cast<const u32x4_t*>(p + (long)r * c.ld);
            const u32x4_t q1 = *reinterpret_cast<const u32x4_t*>(p + (long)(r + RL) * c.ld);
            const u32x4_t q2 = *reinterpret_cast<const u32x4_t*>(p + (long)(r + 2 * RL) * c.ld);
            const u32x4_t q3 = *reinterpret_cast<const u32x4_t*>(p + (long)(r + 3 * RL) * c.ld);
            add(q0); add(q1); add(q2); add(q3);
        }
        for (; r < c.rows; r += RL) add(*reinterpret_cast<const u32x4_t*>(p + (long)r * c.ld));
    }
    float* red = reinterpret_cast<float*>(smem);       // [RL][64]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r0 * 64 + cg * 8 + j] = acc[j];
    __syncthreads();
    if (tid < 64 && c0 + tid < c.n) {
        float sum = 0.f;
        for (int r = 0; r < RL; ++r) sum += red[r * 64 + tid];
        unsafeAtomicAdd(c.out + c0 + tid, sum);        // one owner per column in this launch (a single add: deterministic); atomic, so that launches on other streams may add to the same bias gradient
    }
}

// Up to UNIVL_GEMM_GROUP_MAX independent problems of the same operand layout in ONE launch (the four weight-gradient
// GEMMs of an encoder layer: each alone covers a fraction of the 256 CUs for one or two K steps, and four dependent
// launches cost four launch latencies).  Workgroups are numbered problem by problem; x fastest, then y, then z.
struct GroupArgs {
    GemmArgs p[UNIVL_GEMM_GROUP_MAX];
    int first[UNIVL_GEMM_GROUP_MAX + 1];      // first workgroup of problem i; first[n..] = total
    int nx[UNIVL_GEMM_GROUP_MAX], nxy[UNIVL_GEMM_GROUP_MAX], nz[UNIVL_GEMM_GROUP_MAX];
    // bias-gradient roles in FRONT of the tiles (colsum_tile: they stream their 64 columns over every token and are the longest
    // workgroups of the launch at thousands of tokens): member i's column tiles are workgroups [cs_first[i], cs_first[i + 1])
    ColsumArgs cs[UNIVL_GEMM_GROUP_MAX];
    int cs_first[UNIVL_GEMM_GROUP_MAX + 1];   // cs_first[n..] = number of role workgroups, a multiple of 8 (XCD relation of the tiles)
};

// A grid smaller than the number of tiles walks them with stride gridDim.x: the "background" form of the layer's
// weight-gradient launch (engine.EncoderStack, UNIVL_WGRAD_BLOCKS) occupies only that many workgroups while the next
// layer's latency-bound dgrad chain runs beside it on another stream.
template <typename T, bool TA, bool TB, int BM, int BN, int NC, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_group_kernel(GroupArgs by_value) {
  // Read the argument struct THROUGH THE KERNARG POINTER: a member's arguments are then scalar loads with a uniform dynamic index.
  // (By-value selects over all four members cost 175 - 245 spilled SGPRs per instantiation -- VERDICT r4.)
  (void)by_value;
  const GroupArgs& g = *(const GroupArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int total = g.first[UNIVL_GEMM_GROUP_MAX];
  const int ncs = g.cs_first[UNIVL_GEMM_GROUP_MAX];
  const bool remap = (g.p[0].flags & UNIVL_GEMM_XCD_MAP) && (int)gridDim.x >= total + ncs && total >= 16;   // one workgroup per tile
  for (int v0 = blockIdx.x; v0 < total + ncs; v0 += gridDim.x) {
    if (sizeof(T) == 2 && v0 < ncs) {                   // roles exist for bf16 groups only (host: cs_roles)
        int m = 0;
#pragma unroll
        for (int i = 1; i < UNIVL_GEMM_GROUP_MAX; ++i) m += (v0 >= g.cs_first[i]) ? 1 : 0;
        const ColsumArgs c = g.cs[m];
        const int cf = g.cs_first[m];
        if constexpr (sizeof(T) == 2) { if (v0 - cf < c.n_tiles) colsum_tile<64 * WGM * WGN>(c, v0 - cf, smem_raw); }
        if ((int)gridDim.x < total + ncs) __syncthreads();
        continue;
    }
    const int w0 = v0 - ncs;
    const int w = remap ? xcd_run(w0, total) : w0;
    int idx = 0;
#pragma unroll
    for (int i = 1; i < UNIVL_GEMM_GROUP_MAX; ++i) idx += (w >= g.first[i]) ? 1 : 0;
    const GemmArgs p = g.p[idx];
    const int first = g.first[idx], nx = g.nx[idx], nxy = g.nxy[idx], nz = g.nz[idx];
    const int local = w - first;
    int bx, by, bz;
    if (remap) {
        tile_of(local, nx, nxy / nx, nz, p.gm, bx, by, bz);
    } else {
        bz = local / nxy;
        const int rem = local - bz * nxy;
        by = rem / nx;
        bx = rem - by * nx;
    }
    gemm_tile<T, TA, TB, BM, BN, NC, WGM, WGN>(p, bx, by, bz, nz);
    if ((int)gridDim.x < total + ncs) __syncthreads();           // the next tile's DMA reuses the LDS stages
  }
}

template <typename T, bool TA, bool TB, int BM, int BN, int NC, int WGM = 2, int WGN = 2>
int launch_group(const GroupArgs& g, int max_blocks, hipStream_t stream) {
    constexpr int BK = NC * Mma<T>::CH, NT = 64 * WGM * WGN;
    using TileA = Tile<T, TA, BM, BK, NT>;
    using TileB = Tile<T, TB, BN, BK, NT>;
    // NC == 6 is the one-K-step variant: one stage of each operand
    const size_t smem = NC == 6 ? TileA::BYTES + TileB::BYTES : 2 * (TileA::BYTES + TileB::BYTES);
    static bool attr_done[UNIVL_MAX_DEVICES] = {};   // per instantiation, per device
    if (smem > 48 * 1024) univl_allow_lds(gemm_group_kernel<T, TA, TB, BM, BN, NC, WGM, WGN>, smem, attr_done);
    const int total = g.first[UNIVL_GEMM_GROUP_MAX] + g.cs_first[UNIVL_GEMM_GROUP_MAX];
    const int grid = (max_blocks > 0 && max_blocks < total) ? max_blocks : total;
    hipLaunchKernelGGL((gemm_group_kernel<T, TA, TB, BM, BN, NC, WGM, WGN>), dim3(grid), dim3(NT), smem, stream, g);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// The grouped launch on the 256 x 256 body (gemm256.h): a layer's four weight-gradient products over thousands of tokens, one workgroup
// per tile, bias-gradient roles in front as above.  The kernel-argument struct is read THROUGH THE KERNARG POINTER with a uniform
// dynamic index (scalar loads of one member's arguments) instead of by-value selects over all four members -- those cost the older
// grouped kernel 175 - 245 spilled SGPRs.
template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm256_group_kernel(GroupArgs by_value) {
    (void)by_value;
    const GroupArgs& g = *(const GroupArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int total = g.first[UNIVL_GEMM_GROUP_MAX];
    const int ncs = g.cs_first[UNIVL_GEMM_GROUP_MAX];
    const int v0 = blockIdx.x;
    if (v0 < ncs) {
        int m = 0;
#pragma unroll
        for (int i = 1; i < UNIVL_GEMM_GROUP_MAX; ++i) m += (v0 >= g.cs_first[i]) ? 1 : 0;
        const ColsumArgs c = g.cs[m];
        if (v0 - g.cs_first[m] < c.n_tiles) colsum_tile<512>(c, v0 - g.cs_first[m], smem_raw);
        return;
    }
    const int w0 = v0 - ncs;
    const int w = total >= 16 ? xcd_run(w0, total) : w0;
    int idx = 0;
#pragma unroll
    for (int i = 1; i < UNIVL_GEMM_GROUP_MAX; ++i) idx += (w >= g.first[i]) ? 1 : 0;
    const GemmArgs p = g.p[idx];
    const int nx = g.nx[idx], ny = g.nxy[idx] / nx, nz = g.nz[idx];
    int bx, by, bz;
    tile_of(w - g.first[idx], nx, ny, nz, p.gm, bx, by, bz);
    gemm256_tile<TA, TB>(p, bx, by, bz);
}

template <bool TA, bool TB>
int launch_group256(const GroupArgs& g, hipStream_t stream) {
    static bool attr_done[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(gemm256_group_kernel<TA, TB>, G256_SMEM, attr_done);
    const int total = g.first[UNIVL_GEMM_GROUP_MAX] + g.cs_first[UNIVL_GEMM_GROUP_MAX];
    hipLaunchKernelGGL((gemm256_group_kernel<TA, TB>), dim3(total), dim3(512), G256_SMEM, stream, g);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// (engine.EncoderStack, UNIVL_WGRAD_RIDE, default on): a dgrad product and the weight-gradient product that consumes the SAME
// upstream gradient in ONE launch.  At a few hundred tokens a dgrad kernel is a latency chain on 36-144 workgroups that leaves most
// of the 256 compute units idle, while the layer's grouped weight-gradient launch (1728 tiles at 4 pairs x 48 tokens) is the longest
// node of the backward chain (~29 us of ~100 us per layer) although nothing downstream waits for it.  Here the weight-gradient
// tiles fill the compute units the dgrad leaves idle: the dgrad tiles take the first workgroup ids (dispatched first, spread over
// the 8 XCDs; padded to a multiple of 8 so that the weight-gradient ids keep their own id % 8 = XCD relation), the weight-gradient
// tiles follow.  No extra stream, no extra graph branch (every cross-branch edge of a hipGraph costs ~30 us here).
// Both tile bodies are the 64 x 64 tile on 8 waves; NCW = K-step depth of the weight-gradient body (6: the 192-deep single step).
// Arrival counters of the LayerNorm folds (ln_fold / ln_fold_bwd below): the last LN_SHARE workgroups to contribute to a 64-row block of a
// product's output finish the LayerNorm work on that block.
constexpr int LN_SHARE = 8;
// Forward progress of the folds' spin (fold_arrive): the LN_SHARE - 1 earlier ones of a row block's last arrivals wait for the block's
// LATER arrivals -- workgroups that may not have been dispatched yet (arrival order is not dispatch order).  That cannot deadlock as long
// as the spinners can never occupy every resident-workgroup slot: at most 7 per row block spin, so a launch holds at most 7 x (row
// blocks) spinning workgroups.  The entry points refuse more than 16 row blocks (1024 rows: 112 spinners against >= 512 slots of two
// 512-thread workgroups per compute unit, also with a second fold launch running beside it on another stream); the plans fold up to
// 512 rows (8 blocks).
constexpr int LN_FOLD_MAX_BLOCKS = 16;

struct LnFold {
    UnivlLayerNorm ln;
    int* counters;
};

struct PairArgs {
    GemmArgs d, w;
    int nd, nd_pad, nw;
    int dnx, dny, dnz, wnx, wny, wnz;
    ColsumArgs cs;
};

// The arrival protocol shared by both folds: returns the slab (0 .. LN_SHARE - 1) of the 64-row block `by` this workgroup finishes, or
// -1 (not one of the last LN_SHARE arrivals).  No agent-scope FENCE anywhere: __threadfence() is an L2 write-back plus an L2
// invalidate of the whole XCD, per wave, in a launch whose other workgroups stream optimizer state and weight tiles through that L2
// (measured: +70 us per launch, profiles/r04o_ab_ln_fold_with_agent_fences.txt).  The contributions are fp32 atomics (performed at
// the device's coherence point; the host sets UNIVL_GEMM_ATOMIC for every fold launch), so "this thread's contributions are done" is
// s_waitcnt vmcnt(0), and the rows are read back with agent-scope loads (ln_body.h: XC).
__device__ __forceinline__ int fold_arrive(int* cnt, const int n) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // every thread's atomics are performed; the LDS stages are free from here on
    int* slot = reinterpret_cast<int*>(smem_raw);
    if (threadIdx.x == 0) slot[0] = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrival = slot[0];
    if (arrival < n - LN_SHARE) return -1;
    if (arrival != n - 1) {                            // the earlier ones of the last arrivals wait for the rest (all of them running)
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    asm volatile("" ::: "memory");
    return arrival - (n - LN_SHARE);
}

__device__ __forceinline__ void fold_leave(int* cnt) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == LN_SHARE - 1) {
            // every sharer is past its spin: ready for the next launch / graph replay
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Backward twin (the dgrad half of a pair launch whose fp32 output is the upstream gradient `dout` of a LayerNorm backward,
// module_bert.py:207-211 / 246-250 differentiated): the last arrivals of a row block run the LayerNorm backward of 8 rows each.
__device__ __forceinline__ void ln_fold_bwd(const LnFold& f, const int by, const int n) {
    int* cnt = f.counters + 2 * by;
    const int slab = fold_arrive(cnt, n);
    if (slab < 0) return;
    ln_bwd_rows<768, __bf16, 8, true>(f.ln, by * 64 + slab * 8, reinterpret_cast<float*>(smem_raw));
    fold_leave(cnt);
}

__host__ __device__ __forceinline__ void pair_tile(int t, int total, int nx, int ny, int nz, int flags, int gm, int& bx, int& by, int& bz) {
    if ((flags & UNIVL_GEMM_XCD_MAP) && total >= 16 && ny > 1) {
        tile_of(xcd_run(t, total), nx, ny, nz, gm, bx, by, bz);
    } else {
        const int nxy = nx * ny;
        bz = t / nxy;
        const int rem = t - bz * nxy;
        by = rem / nx;
        bx = rem - by * nx;
    }
}

// Forms of the launch (both bodies on 8 waves; DR / WF select the dgrad / weight-gradient body):
//   dgrad   DR = false: 64 x 64 tiles, 128-deep K steps (a few hundred tokens: the product is a latency chain and wants as many
//                       workgroups as there are compute units)
//           DR = true : 64 x 128 tiles, 64-deep K steps (from 384 tokens on, dgrad slices up to 1536 deep)
//   wgrad   WF = 0: 64 x 64 tiles, 128-deep K steps      WF = 1: 64 x 64, the single 192-deep step (round 3's form at 192 tokens)
//           WF = 2: 128 x 64 tiles, 64-deep K steps       WF = 3: 128 x 64, the single 192-deep step
//   (DR, WF) = (false, 0): round 2's form, now only below 384 tokens where neither operand is 8-aligned enough for the others;
//   (true, 2): THREE workgroups per compute unit (48 KB of LDS, <= 80 VGPRs).  At 768 tokens both products are bound by what a
//     compute unit can pull through LDS-DMA (~45 GB/s: the phase trace, profiles/r04c_trace_gemm_768_variants.txt), i.e. by the
//     bytes the tiles stage: an FFN pair stages 226 MB as 1152 square tiles (2.25 rounds of 512 slots, 33 us) but 170 MB as 576
//     rectangular ones -- one round of 768 slots, 25 us;
//   (false, 3) / (false, 2): below 384 tokens the dgrad stays square, the weight gradient takes the 128 x 64 tile -- half as many
//     workgroups behind the dgrad tiles (FFN: 288 instead of 576) staging 3/4 of the bytes: in the phase trace at 192 tokens
//     (profiles/r04g_trace_gemm_phases.txt) the square weight-gradient tiles of the FFN1 / QKV pairs end 3 - 4 us after the dgrad
//     chain they ride with (17.6 / 15.9 us per launch against 14.8 for the dgrad-bound FFN2 pair).
template <bool DR, int WF, bool FOLD = false>
__global__ __launch_bounds__(512, (DR && WF == 2 && !FOLD) ? 6 : 4) void gemm_pair_kernel(PairArgs a, LnFold f) {      // 4 waves per SIMD = two workgroups per compute unit (the fold's LayerNorm rows need the registers)
    const int w0 = blockIdx.x;
    int bx, by, bz;
    if (w0 < a.nd_pad) {
        if (w0 >= a.nd) return;                                  // padding workgroup (whole block: no barrier is skipped)
        pair_tile(w0, a.nd, a.dnx, a.dny, a.dnz, a.d.flags, a.d.gm, bx, by, bz);
        if constexpr (DR) gemm_tile<__bf16, false, true, 64, 128, 2, 2, 4>(a.d, bx, by, bz, a.dnz);
        else gemm_tile<__bf16, false, true, 64, 64, 4, 2, 4>(a.d, bx, by, bz, a.dnz);
        if constexpr (FOLD) ln_fold_bwd(f, by, a.dnx * a.dnz);
    } else if (w0 < a.nd_pad + a.cs.n_pad) {
        const int t = w0 - a.nd_pad;
        if (t >= a.cs.n_tiles) return;
        colsum_tile<512>(a.cs, t, smem_raw);
    } else {
        pair_tile(w0 - a.nd_pad - a.cs.n_pad, a.nw, a.wnx, a.wny, a.wnz, a.w.flags, a.w.gm, bx, by, bz);
        if constexpr (WF == 0) gemm_tile<__bf16, true, true, 64, 64, 4, 2, 4>(a.w, bx, by, bz, a.wnz);
        else if constexpr (WF == 1) gemm_tile<__bf16, true, true, 64, 64, 6, 2, 4>(a.w, bx, by, bz, a.wnz);
        else if constexpr (WF == 2) gemm_tile<__bf16, true, true, 128, 64, 2, 4, 2>(a.w, bx, by, bz, a.wnz);
        else gemm_tile<__bf16, true, true, 128, 64, 6, 4, 2>(a.w, bx, by, bz, a.wnz);
    }
}

template <bool DR, int WF, bool FOLD = false>
int launch_pair(const PairArgs& a, hipStream_t stream, const LnFold* f = nullptr) {
    constexpr size_t smem_d = DR ? 2 * (size_t)(64 + 128) * 64 * sizeof(__bf16) : 2 * (size_t)(64 + 64) * 128 * sizeof(__bf16);
    constexpr size_t smem_w = WF == 0 ? 2 * (size_t)(64 + 64) * 128 * sizeof(__bf16)
                              : WF == 1 ? (size_t)(64 + 64) * 192 * sizeof(__bf16)
                              : WF == 2 ? 2 * (size_t)(128 + 64) * 64 * sizeof(__bf16) : (size_t)(128 + 64) * 192 * sizeof(__bf16);
    constexpr size_t smem = smem_d > smem_w ? smem_d : smem_w;        // >= 8 x 768 floats: the fold's column-sum pass fits in it
    static_assert(!FOLD || smem >= 8 * 768 * sizeof(float), "LayerNorm-backward fold needs 24 KB of LDS");
    static bool attr_done[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(gemm_pair_kernel<DR, WF, FOLD>, smem, attr_done);
    LnFold none = {};
    hipLaunchKernelGGL((gemm_pair_kernel<DR, WF, FOLD>), dim3(a.nd_pad + a.cs.n_pad + a.nw), dim3(512), smem, stream, a, f ? *f : none);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the attention-output dgrad (dctx = dY . W_o, module_bert.py:207 differentiated) INSIDE the attention backward.  At a few
// hundred tokens the backward of a layer was five dependent launches; two of them -- the pair launch of this dgrad (36 tiles, 13 us)
// and the attention backward that consumes its output (9.5 us) -- exchange nothing but a [tokens, 768] bf16 matrix whose (batch row,
// head) block is EXACTLY one 64 x 64 product tile (head width 64 = tile width; sequences of at most 64 positions).  So every workgroup of
// the attention backward computes the dO block of its own (batch row, head) first -- the K loop of gemm_tile on the sequence's rows
// and the head's columns of W_o, rounded to bf16 exactly as the dgrad's epilogue did -- writes it into the LDS images the attention
// body reads (attn_body.h: FUSED), and carries on.  No cross-workgroup hand-off, one launch and one [tokens, 768] round trip through
// HBM less per layer; dQ / dK / dV are BIT-IDENTICAL to the two launches (same chunk order per output element).  The weight gradient
// that rode with the dgrad (dW_o = dY^T . ctx) rides here: its tiles take the workgroups behind the attention roles.
// K loop of gemm_tile alone (no epilogue): acc <- A_op[m0.., K] . B_op[n0.., K]^T, K a multiple of the step depth.
template <typename T, bool TA, bool TB, int BM, int BN, int NC, int WGM, int WGN>
__device__ __forceinline__ void gemm_acc_only(const T* A, long lda, const T* B, long ldb, int M, int N, int K, int m0, int n0,
                                              f32x4_t (&acc)[BM / WGM / 16][BN / WGN / 16], unsigned char* smem) {
    constexpr int BK = NC * Mma<T>::CH, NT = 64 * WGM * WGN;
    using TileA = Tile<T, TA, BM, BK, NT>;
    using TileB = Tile<T, TB, BN, BK, NT>;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    unsigned char* sA = smem;
    unsigned char* sB = sA + 2 * TileA::BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const T* pa[TileA::PER_THREAD];
    const T* pb[TileB::PER_THREAD];
    TileA::init(pa, A, lda, m0, 0, M, tid);
    TileB::init(pb, B, ldb, n0, 0, N, tid);
    const long stepA = TileA::kstep(lda), stepB = TileB::kstep(ldb);
    const int nfull = K / BK;
    TileA::issue(pa, sA, tid);
    TileB::issue(pb, sB, tid);
    TileA::advance(pa, stepA);
    TileB::advance(pb, stepB);
    __syncthreads();
    for (int t = 0; t < nfull; ++t) {
        const int cur = t & 1;
        if (t + 1 < nfull) {
            TileA::issue(pa, sA + (cur ^ 1) * TileA::BYTES, tid);
            TileB::issue(pb, sB + (cur ^ 1) * TileB::BYTES, tid);
            TileA::advance(pa, stepA);
            TileB::advance(pb, stepB);
        }
        tile_mma<T, TA, TB, MI, NI, NC, TileA, TileB>(sA + cur * TileA::BYTES, sB + cur * TileB::BYTES, wm0, wn0, lane, acc);
        __syncthreads();
    }
}

#include "vocab_ce.h"
#include "attn_body.h"

struct AttnFusedArgs {
    UnivlAttention at;
    int Sk_pad, Sq_pad;
    const __bf16* dY; long lddy;        // upstream gradient of the attention-output projection [tokens, K]
    const __bf16* W; long ldw;          // its weight, [K][H * 64] as the dgrad reads it (T-major B)
    int K;
    int n_attn, n_attn_pad;             // attention-role workgroups (2 per (batch row, head)), padded to a multiple of 8
    GemmArgs w;                         // the riding weight gradient (nw = 0: none)
    int nw, wnx, wny, wnz;
};

template <int TRIPS, bool DUAL, int NCW>
__global__ __launch_bounds__(256) void attn_bwd_odgrad_kernel(AttnFusedArgs a) {
    const int w0 = blockIdx.x;
    if (w0 < a.n_attn_pad) {
        if (w0 >= a.n_attn) return;
        const int bh = w0 >> 1, role = w0 & 1;              // one query block (role 0: dQ) and one key block (role 1: dK, dV)
        const int b = bh / a.at.H, h = bh % a.at.H;
        f32x4_t acc[2][2];
        gemm_acc_only<__bf16, false, true, 64, 64, 4, 2, 2>(a.dY + (long)b * a.at.Sq * a.lddy, a.lddy, a.W, a.ldw, a.at.Sq, a.at.H * 64, a.K,
                                                            0, h * 64, acc, smem_raw);
        // (the K loop's last barrier is behind us: every wave is done with the stages, the attention images may overwrite them)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int Sq = a.at.Sq, Sq_pad = a.Sq_pad;
        auto dow = [&](void* img_, int pitch) {
            __bf16* img = reinterpret_cast<__bf16*>(img_);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm0 + 16 * ta + 4 * g + r, col = wn0 + 16 * tb + i;
                        // the images hold Sq_pad rows (32 or 64); rows beyond the sequence are zero, as stage_pair fills them
                        if (row < Sq_pad) img[row * pitch + col] = row < Sq ? (__bf16)acc[ta][tb][r] : (__bf16)0.0f;
                    }
        };
        attn_bwd_body<__bf16, TRIPS, DUAL, true>(a.at, a.Sk_pad, a.Sq_pad, 1, 0.125f, bh, role, smem_raw, dow);
    } else {
        int bx, by, bz;
        pair_tile(w0 - a.n_attn_pad, a.nw, a.wnx, a.wny, a.wnz, a.w.flags, a.w.gm, bx, by, bz);
        gemm_tile<__bf16, true, true, 64, 64, NCW, 2, 2>(a.w, bx, by, bz, a.wnz);
    }
}

// The forward twin: the query / key / value projection (module_bert.py:172-174: three nn.Linear on the layer input) INSIDE the attention
// forward.  A workgroup owns one (batch row, head): it multiplies the 64 x 192 block  x[rows of the sequence] . [W_q | W_k | W_v][head]^T
// (+ bias), stores it to the qkv buffer the backward reads (same bf16 bits as the product's own epilogue), keeps it in the LDS images of
// attn_fwd_body (FUSED) and runs the attention on it: the attention launch and its [tokens, 2304] read leave the forward chain.  The
// launch carries BertAdam chunks like gemm_adam_kernel (workgroups behind the attention ones): the forward launches of a layer are
// bound by the optimizer bytes that ride in them (16 us against 10 without), so the longer per-workgroup product (384 KB through
// one compute unit instead of 192) hides behind them.  Eight waves (2 x 4: 32 x 48 outputs each) multiply, the first four attend.
struct AttnFwdFusedArgs {
    UnivlAttention at;
    int Sk_pad;
    const __bf16* X; long ldx;          // layer input [tokens, K]
    const __bf16* W; long ldw;          // [3 * H * 64, K]: q | k | v weights, K-major
    const float* bias;                  // [3 * H * 64] or null
    __bf16* qkv; long ldqkv;            // [tokens, 3 * H * 64]
    int K;
    int n_attn, n_attn_pad;
    const __bf16* X_lo; const __bf16* W_lo;     // operand pairs of the projection (UnivlGemm.A_lo / B_lo), or null
};

// BIG (round 6): sequences of 65 .. 128 positions (the caption configuration's 128 words / 96 frames, the cross encoder's 96 / 112).  Two
// workgroups per (batch row, head), one per block of 64 queries; each multiplies the 128 x 192 block of q | k | v of the WHOLE sequence (the
// attention of its query block needs every key and value; the other block's q rows are recomputed -- at these sizes the product is a
// latency chain, not MFMA time), stores the q | k | v rows of ITS OWN block to the qkv buffer (every row written once) and attends.
template <bool NT, bool BIG>
__global__ __launch_bounds__(512, 2) void attn_fwd_qkv_kernel(AttnFwdFusedArgs a, UnivlAdam ad, int c0, int c1) {
    const int w0 = blockIdx.x;
    if (w0 >= a.n_attn_pad) {
        const int nb = (int)gridDim.x - a.n_attn_pad;
        for (int c = c0 + (w0 - a.n_attn_pad); c < c1; c += nb) adam_chunk<NT, 512>(ad, c);
        return;
    }
    if (w0 >= a.n_attn) return;
    constexpr int ROWS = BIG ? 128 : 64, MI = ROWS / 32;
    using TileA = Tile<__bf16, false, ROWS, 64, 512>;
    using TileB = Tile<__bf16, false, 192, 64, 512>;
    const int bh = BIG ? (w0 >> 1) : w0, yblk = BIG ? (w0 & 1) : 0, b = bh / a.at.H, h = bh % a.at.H;
    const int HD3 = a.at.H * 64, Sq = a.at.Sq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int wm0 = (wave >> 2) * (ROWS / 2), wn0 = (wave & 3) * 48;
    unsigned char* sA = smem_raw;
    unsigned char* sB = sA + 2 * TileA::BYTES;
    f32x4_t acc[MI][3];
#pragma unroll
    for (int ta = 0; ta < MI; ++ta)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) acc[ta][tb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const __bf16* pa[TileA::PER_THREAD];
    const __bf16* pb[TileB::PER_THREAD];
    TileA::init(pa, a.X + (long)b * Sq * a.ldx, a.ldx, 0, 0, Sq, tid);
#pragma unroll
    for (int c = 0; c < TileB::PER_THREAD; ++c) {           // tile row r -> weight row (r / 64) * H * 64 + h * 64 + r % 64: the head's rows of W_q, W_k, W_v
        int r, q;
        TileB::coords(c, tid, r, q);
        pb[c] = a.W + (long)((r >> 6) * HD3 + h * 64 + (r & 63)) * a.ldw + q * 8;
    }
    // operand pairs: x.W | x.W_lo | x_lo.W, one walk of the contraction per term (gemm_tile)
    const int nseg = 1 + (a.W_lo != nullptr ? 1 : 0) + (a.X_lo != nullptr ? 1 : 0);
    const int per = a.K / 64, nfull = nseg * per;
    int seg = 0, wrap = per;
    auto seg_x = [&](int s_) { return (a.X_lo != nullptr && s_ > 0 && s_ == nseg - 1) ? a.X_lo : a.X; };
    auto seg_w = [&](int s_) { return (a.W_lo != nullptr && s_ == 1) ? a.W_lo : a.W; };
    TileA::issue(pa, sA, tid);
    TileB::issue(pb, sB, tid);
    TileA::advance(pa, 64);
    TileB::advance(pb, 64);
    __syncthreads();
    for (int t = 0; t < nfull; ++t) {
        const int cur = t & 1;
        if (t + 1 < nfull) {
            if (t + 1 == wrap) {
                TileA::advance(pa, (seg_x(seg + 1) - seg_x(seg)) - (long)a.K);
                TileB::advance(pb, (seg_w(seg + 1) - seg_w(seg)) - (long)a.K);
                ++seg;
                wrap += per;
            }
            TileA::issue(pa, sA + (cur ^ 1) * TileA::BYTES, tid);
            TileB::issue(pb, sB + (cur ^ 1) * TileB::BYTES, tid);
            TileA::advance(pa, 64);
            TileB::advance(pb, 64);
        }
        tile_mma<__bf16, false, false, MI, 3, 2, TileA, TileB>(sA + cur * TileA::BYTES, sB + cur * TileB::BYTES, wm0, wn0, lane, acc);
        __syncthreads();
    }
    // epilogue: + bias -> bf16 -> the qkv buffer (rows of the sequence) and the LDS images of the attention body (rows beyond it: zero)
    const AttnFwdImages<__bf16> im(smem_raw, a.Sk_pad);
    using C = AttnCfg<__bf16>;
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
        const int col = wn0 + 16 * tb + i, panel = col >> 6, c64 = col & 63;
        const int gcol = panel * HD3 + h * 64 + c64;
        const float bv = a.bias ? a.bias[gcol] : 0.0f;
        __bf16* img = panel == 0 ? im.sQ : (panel == 1 ? im.sK : im.sV);
        const int pitch = panel == 2 ? C::PT : C::PK;
#pragma unroll
        for (int ta = 0; ta < MI; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + 16 * ta + 4 * g + r;
                const __bf16 v = (__bf16)(acc[ta][tb][r] * 1.0f + bv);
                if (row < Sq && (!BIG || (row >> 6) == yblk)) a.qkv[((long)b * Sq + row) * a.ldqkv + gcol] = v;
                // images: K and V hold Sk_pad rows; the Q image 64 rows (BIG: Sk_pad rows, indexed by the position in the sequence)
                if (row < a.Sk_pad || (!BIG && panel == 0)) img[row * pitch + c64] = row < Sq ? v : (__bf16)0.0f;
            }
    }
    attn_fwd_body<__bf16, BIG ? 8 : 4, true>(a.at, a.Sk_pad, 0.125f, bh, yblk, smem_raw);
}

// EXPERIMENTAL (UNIVL_ADAM_RIDE=1 with graphed.GraphedTrainStep(pipeline_optimizer=True)): a forward product and a range of BertAdam
// chunks in ONE launch.  The fused update is one 0.8 ms HBM stream (30 B per parameter) at the end of a step whose forward is a chain
// of latency-bound kernels on a third of the compute units; its only ordering constraints are "after the clip of its own backward"
// and "layer l's parameters before layer l's first kernel of the NEXT forward".  So the update of step t goes out with the forward of
// step t + 1: the chunks of layer l + 1 as extra workgroups of the forward products of layer l (same stream: the kernel boundary is
// the dependency), with no second stream and no graph branch -- the earlier side-stream form of this overlap lost ~0.6 ms to 19
// cross-branch graph edges.  The GEMM tiles take the first workgroup ids (dispatched first).
template <bool NT>
__global__ __launch_bounds__(512, 2) void gemm_adam_kernel(GemmArgs g, int nd, int nd_pad, int nx, int ny, int nz, UnivlAdam a, int c0, int c1) {
    const int w0 = blockIdx.x;
    if (w0 < nd_pad) {
        if (w0 >= nd) return;                                  // padding workgroup
        int bx, by, bz;
        pair_tile(w0, nd, nx, ny, nz, g.flags, g.gm, bx, by, bz);
        gemm_tile<__bf16, false, false, 64, 64, 4, 2, 4, true>(g, bx, by, bz, nz);
    } else {
        const int nb = (int)gridDim.x - nd_pad;
        for (int c = c0 + (w0 - nd_pad); c < c1; c += nb) adam_chunk<NT, 512>(a, c);
    }
}

// Round 5: the same for forward products on the 64 x 128 tile (what `choose` picks from 1536 tokens on).  There univl_gemm_rider used to
// enqueue product and update one after the other -- 57 update launches, 0.7 - 1.1 ms of serial HBM time in an MFMA-bound step at 128
// pairs (profiles/r05_final_bench_b128_kernel_stats.csv).  Three workgroups per compute unit like gemm_kernel's instantiation of the tile.
// Here the product has several rounds of tiles (2304 at 6144 tokens against 768 resident slots), so update workgroups BEHIND the tiles
// would only meet the last round: they are spread through the grid instead, in groups of 8 consecutive workgroup ids (one per XCD, so
// that a tile's id modulo 8 -- what the tile order is built on -- stays what it was): group gi is an update group iff gi % pg == pg - 1
// and gi / pg < nrg, with nrg = update groups, pg = groups per update group.
// true: workgroup w0 of `total` is update workgroup idx of total - nd_pad; false: it is tile idx of nd_pad (total, nd_pad multiples of 8)
__host__ __device__ __forceinline__ bool rect_rider_role(int w0, int total, int nd_pad, int& idx) {
    const int gi = w0 >> 3, nrg = (total - nd_pad) >> 3, pg = (total >> 3) / nrg;
    const int k = gi / pg;
    if (gi - k * pg == pg - 1 && k < nrg) { idx = k * 8 + (w0 & 7); return true; }
    idx = w0 - 8 * (k < nrg ? k : nrg);
    return false;
}

template <bool NT>
__global__ __launch_bounds__(512, 6) void gemm_adam_rect_kernel(GemmArgs g, int nd, int nd_pad, int nx, int ny, int nz, UnivlAdam a, int c0, int c1) {
    int t;
    if (rect_rider_role((int)blockIdx.x, (int)gridDim.x, nd_pad, t)) {
        const int nb = (int)gridDim.x - nd_pad;
        for (int c = c0 + t; c < c1; c += nb) adam_chunk<NT, 512, false>(a, c);
    } else {
        if (t >= nd) return;                                   // padding workgroup
        int bx, by, bz;
        pair_tile(t, nd, nx, ny, nz, g.flags, g.gm, bx, by, bz);
        gemm_tile<__bf16, false, false, 64, 128, 2, 2, 4>(g, bx, by, bz, nz);
    }
}

// K8 / K10 (module_bert.py:207-211, 246-250: dense -> dropout -> + input -> LayerNorm): the LayerNorm that consumes a product's fp32
// output, finished INSIDE the product's launch.  The output rows of a 64-row block are complete when all nx * nz tiles of the block
// (column tiles x split-K slices, meeting in fp32 atomics) have added to them; every tile workgroup announces itself
// on the block's counter once its atomics are performed, and the LAST LN_SHARE arrivals normalise 8 rows each (one row per wave, the
// body of ln_fwd_kernel) once the counter is full -- the earlier ones of them spin on it, which cannot deadlock: they only wait for
// workgroups that are already running.  What it buys: one kernel boundary (~3 us here) and one launch's fixed cost per LayerNorm on
// a chain of ~190 dependent launches per step; what it costs: a counter round trip on the product's tail.
// counters: two ints per row block (arrivals, finished sharers), zero before the first launch; the last sharer re-zeroes them.
__device__ __forceinline__ void ln_fold(const LnFold& f, const int by, const int n, const int M) {
    int* cnt = f.counters + 2 * by;
    const int slab = fold_arrive(cnt, n);
    if (slab < 0) return;
    const int row = by * 64 + slab * 8 + (int)(threadIdx.x >> 6);
    if (row < M) ln_fwd_row<768, __bf16, false, true>(f.ln, row, (int)(threadIdx.x & 63));
    fold_leave(cnt);
}

template <bool NT>
__global__ __launch_bounds__(512, 2) void gemm_ln_kernel(GemmArgs g, int nd, int nd_pad, int nx, int ny, int nz, UnivlAdam a, int c0, int c1, LnFold f) {
    const int w0 = blockIdx.x;
    if (w0 < nd_pad) {
        if (w0 >= nd) return;                                  // padding workgroup
        int bx, by, bz;
        pair_tile(w0, nd, nx, ny, nz, g.flags, g.gm, bx, by, bz);
        gemm_tile<__bf16, false, false, 64, 64, 4, 2, 4, true>(g, bx, by, bz, nz);
        ln_fold(f, by, nx * nz, g.M);
    } else {
        const int nb = (int)gridDim.x - nd_pad;
        for (int c = c0 + (w0 - nd_pad); c < c1; c += nb) adam_chunk<NT, 512>(a, c);
    }
}

template <typename T, bool TA, bool TB, int BM, int BN, int NC, int WGM = 2, int WGN = 2>
int launch(const GemmArgs& a, int ksplit, hipStream_t stream) {
    constexpr int BK = NC * Mma<T>::CH, NT = 64 * WGM * WGN;
    using TileA = Tile<T, TA, BM, BK, NT>;
    using TileB = Tile<T, TB, BN, BK, NT>;
    // NC == 6 is the one-K-step variant: one stage of each operand
    const size_t smem = NC == 6 ? TileA::BYTES + TileB::BYTES : 2 * (TileA::BYTES + TileB::BYTES);
    static bool attr_done[UNIVL_MAX_DEVICES] = {};   // per instantiation, per device
    if (smem > 48 * 1024) univl_allow_lds(gemm_kernel<T, TA, TB, BM, BN, NC, WGM, WGN>, smem, attr_done);
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, ksplit);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, BM, BN, NC, WGM, WGN>), grid, dim3(NT), smem, stream, a);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

template <typename T, int BM, int BN, int NC, int WGM = 2, int WGN = 2>
int dispatch_trans(const GemmArgs& a, int ta, int tb, int ksplit, hipStream_t s) {
    if (!ta && !tb) return launch<T, false, false, BM, BN, NC, WGM, WGN>(a, ksplit, s);
    if (!ta && tb) return launch<T, false, true, BM, BN, NC, WGM, WGN>(a, ksplit, s);
    if (ta && tb) return launch<T, true, true, BM, BN, NC, WGM, WGN>(a, ksplit, s);
    return launch<T, true, false, BM, BN, NC, WGM, WGN>(a, ksplit, s);
}

}  // namespace

// fp32 products with at most 32 x 32 outputs (the B x B similarity matrix of modeling.py:389 at small batch): one
// workgroup per output element, a 256-lane dot product -- the tiled kernel would run K/64 dependent steps for one tile.
__global__ __launch_bounds__(256) void dot_kernel(GemmArgs p) {
    __shared__ float red[4];
    const int n = blockIdx.x, m = blockIdx.y, t = threadIdx.x;
    const float* a = reinterpret_cast<const float*>(p.A) + (long)m * p.lda;
    const float* b = reinterpret_cast<const float*>(p.B) + (long)n * p.ldb;
    float acc = 0.f;
    const int k4 = p.K & ~3;
    for (int k = 4 * t; k < k4; k += 1024) {
        const float4 x = *reinterpret_cast<const float4*>(a + k);
        const float4 y = *reinterpret_cast<const float4*>(b + k);
        acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    if (t < p.K - k4) acc += a[k4 + t] * b[k4 + t];
    acc = wave_sum(acc);
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        float v = ((red[0] + red[1]) + (red[2] + red[3])) * p.alpha;
        const long o = (long)m * p.ldc + n;
        if (p.flags & UNIVL_GEMM_ACCUM) v += p.C32[o];
        if (p.C32) p.C32[o] = v;
        if (p.C16) reinterpret_cast<float*>(p.C16)[o] = v;
    }
}

// validation + kernel arguments shared by the single and the grouped entry point.
// Tile geometry (UnivlGemm.tile, 0 = chosen here); the two-stage pipeline is the only one (UnivlGemm.stages: 0 or 2):
//   64   64 x 64,  BK = 128 bf16 (64 for the deep grouped weight gradients, 192 for the one-step weight gradients at 192 tokens)
//        -- everything below 256 big tiles: parallelism over tile efficiency
//   128  128 x 128, BK = 64 -- two workgroups per compute unit
//   12864 / 64128  128 x 64 / 64 x 128, 8 waves, BK = 64, bf16, K-major A (forward / dgrad), no sumsq / dbias, univl_gemm only
//        (else 128): 48 KB of LDS and <= 80 VGPRs, i.e. THREE workgroups per compute unit (768 slots) instead of two (512).
//        Picked instead of the 128 tile where it fills its slots better: a 6144 x 768 output is 288 tiles of 128 x 128 in 512 slots
//        -- 32 units run two workgroups, 224 run one -- but 576 half tiles in 768 slots; 6144 x 3072 is 2.25 rounds of 512 slots
//        against exactly 3 rounds of 768; 6144 x 2304 (1.69 rounds against 2.25) stays on the 128 tile.  Per-shape table at 6144
//        rows: profiles/r03u_gemm_variants_b128.txt (+3..13 % per product where picked; bit-identical results: same 64-deep K
//        steps, same chunk order per output element).
// waves = 4 | 8 (tiles 64 / 128, bf16 only; 0 = 8 where the kernel form allows it).
// Every choice is a function of the descriptor alone: no environment switches (round 4; the A/B history is in DESIGN.md section 8).
struct Choice { int tile, nc, waves; };

constexpr long GEMM_BIG_MIN = 256;      // 128 x 128 tiles from this many of them on (measured: +6 % at 128 pairs over 384, same at 16)
constexpr int GEMM_GM = 8;              // row tiles per L2 band of the tile order (xcd map)
constexpr long GEMM_NC64_MIN = 1024;    // grouped weight gradients contracting over at least this many tokens: 64-deep K steps, 4 waves

// The 256 x 256 body (gemm256.h): bf16, M and N multiples of 256, equal K slices that are multiples of 128 (two K tiles per loop trip),
// K-major x K-major / K-major x T-major / T-major x T-major, gradient-norm tensors that are whole tiles, no in-tile bias gradient
// (in_group: the grouped launch takes a member's bias gradient as column-sum roles before the member gets here).
static bool fits256(const UnivlGemm* d, bool in_group) {
    if (d->dtype != UNIVL_BF16 || (d->trans_a && !d->trans_b) || d->M % 256 != 0 || d->N % 256 != 0) return false;
    if (d->dbias && !in_group) return false;
    if (d->flags & UNIVL_GEMM_AUX_F32) return false;          // (A/B measurement form of the GELU epilogues: the older tiles only)
    if (d->sumsq && d->sumsq_rows % 256 != 0) return false;
    // the epilogue moves 4 consecutive columns per lane: 16-byte fp32 / 8-byte bf16 accesses
    if (d->ldc % 4 != 0 || (d->R && d->ldr % 4 != 0) || (d->aux && d->ldaux % 4 != 0) || !aligned16(d->C32) || !aligned16(d->R) ||
        !aligned16(d->bias) || (((uintptr_t)d->C16 | (uintptr_t)d->aux) & 7) != 0) return false;
    const int ks = (d->ksplit < 1 || univl_deterministic()) ? 1 : d->ksplit;
    return d->K % (128 * ks) == 0;
}

// From where on the 256 body is picked without being asked for (tile = 0).  One workgroup per compute unit, every workgroup of a launch
// in the same phase: a launch costs its K loop (~1.1 us per 64-deep K tile with a third of the units busy, ~1.8 with all of them:
// the units' LDS-DMA streams share the fabric) PLUS a prologue (1.6 us), an epilogue in which all units write at once (5 - 7 us of
// HBM-bound stores with the matrix pipe idle) and a 5 us write-back drain behind it (profiles/r05c_trace_gemm256.txt).  So:
//   * a layer's grouped weight gradients (contraction over >= 1536 tokens: 24+ K tiles per workgroup) always: 121 vs 147 us per layer at
//     6144 tokens;
//   * a single product only when it is ONE round that fills most of the chip (192 .. 256 workgroups): QKV forward at 6144 rows, 35 vs
//     39.5 us; two-round shapes (FFN1 forward, FFN2 dgrad: 288 tiles) and the 72-tile N = 768 products lose to the 64 x 128 / 128 x 128
//     tiles' two or three unsynchronised workgroups per unit (profiles/r05d_mb_rot.txt).
constexpr int G256_MIN_ROWS = 1536;
constexpr long G256_MIN_WG = 192, G256_MAX_WG = 256;
static bool auto256(const UnivlGemm* d, bool in_group) {
    if (!fits256(d, in_group)) return false;
    const int ks = (d->ksplit < 1 || univl_deterministic()) ? 1 : d->ksplit;
    const long wg = (long)(d->M / 256) * (d->N / 256) * ks;
    const int rows = d->trans_a ? d->K : d->M;            // tokens: the contraction of a weight gradient
    return rows >= G256_MIN_ROWS && (in_group || (ks == 1 && wg >= G256_MIN_WG && wg <= G256_MAX_WG));
}

static Choice choose(const UnivlGemm* d, int forced_tile, int forced_nc = 0, bool allow_rect = false, bool in_group = false) {
    const bool bf16 = d->dtype == UNIVL_BF16;
    int want = forced_tile ? forced_tile : d->tile;
    const bool pairs = d->A_lo || d->B_lo || d->C16_lo;       // operand pairs: never the 256 body
    if (!pairs && ((want == 256 && fits256(d, in_group)) || (want == 0 && auto256(d, in_group)))) return Choice{256, 2, 8};
    if (want == 256) want = 128;                   // asked for, but not a product the body carries
    const long tiles128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
    const bool rect_ok = allow_rect && bf16 && !d->trans_a && !d->sumsq && !d->dbias;      // univl_gemm only (single launch)
    Choice c;
    if (want == 12864 || want == 64128) {
        c.tile = rect_ok ? want : 128;
        c.nc = 2;
        c.waves = !rect_ok ? 4 : (d->waves ? d->waves : 8);
        return c;
    }
    c.tile = (want >= 128 || (want == 0 && tiles128 >= GEMM_BIG_MIN)) ? 128 : 64;
    if (want == 0 && c.tile == 64 && rect_ok && d->waves == 0) {
        // Round 5: between the regimes -- fewer than 256 tiles of 128 x 128, but at least 256 of 64 x 128 (N = 768 outputs at 2731+ rows,
        // N = 2304 at 911+): the half tile beats the 64 x 64 tile by 15 - 25 % (3072 rows: QKV dgrad 31.6 vs 39.6 us, FFN1 dgrad 39.5 vs
        // 50.6, FFN2 forward 33.8 vs 39.9; 1536 rows: QKV forward 14.4 vs 18.4; profiles/r05a_mb_gemm256_first_version.txt), while
        // below that the 64 tile's extra parallelism wins (1536 rows, N = 768: 22 - 28 vs 29 - 37 us).
        const long tiles_r = (long)((d->M + 63) / 64) * ((d->N + 127) / 128);
        if (tiles_r >= GEMM_BIG_MIN) return Choice{64128, 2, 8};
    }
    if (want == 0 && c.tile == 128 && rect_ok && d->waves == 0) {
        // fill of the resident-workgroup slots in the last round: tiles / (rounds x slots)
        const long tiles_r = (long)((d->M + 63) / 64) * ((d->N + 127) / 128);
        const double fill128 = (double)tiles128 / (double)(((tiles128 + 511) / 512) * 512);
        const double fill_r = (double)tiles_r / (double)(((tiles_r + 767) / 768) * 768);
        // measured (6144 rows): with T-major B (dgrad) the half tile wins at equal fill and above; with K-major B (forward) it
        // needs a clear fill advantage (0.75 vs 0.84: -7 %; 1.0 vs 0.75: +1 %; 0.75 vs 0.56: +3..9 %)
        if (fill_r >= fill128 * (d->trans_b ? 1.0 : 1.2)) {
            c.tile = 64128;
            c.nc = 2;
            c.waves = 8;
            return c;
        }
    }
    c.nc = c.tile == 64 ? 4 : 2;
    // forced_nc = 2 (the grouped launch of deep weight-gradient products, univl_gemm_group_limited): 64-deep K steps instead of
    // 128-deep ones -- half the LDS per workgroup (32 KB), i.e. four workgroups per compute unit instead of two to hide the DMA
    // waits the stall counters show (profiles/r03z_pmc_stall_b128.txt: 63 % of that kernel's wave cycles are spent parked at the
    // wait / barrier)
    if (forced_nc == 2 && c.tile == 64 && bf16) c.nc = 2;
    const int ksplit = (d->ksplit < 1 || univl_deterministic()) ? 1 : d->ksplit;
    // weight gradients at 4 pairs x 48 tokens contract over exactly 192 rows: one 192-deep stage = ONE DMA round trip
    // per workgroup, no partial tile through registers (the 128-deep stage needs a 128-step plus a masked 64-tail)
    if (c.tile == 64 && bf16 && d->trans_a && d->trans_b && d->K == 192 && ksplit == 1) c.nc = 6;
    // 8 waves on the 64 / 128 tiles: the same tile cut into twice as many wave sub-tiles.  Less MFMA work per fragment read, but
    // twice as many waves issuing LDS-DMA (measured, round 2: -2 % per step at 4 pairs, -5 % at 128)
    c.waves = d->waves ? d->waves : 8;
    if (c.waves != 8 || !bf16 || c.nc == 6 || (d->sumsq && c.tile == 64)) c.waves = 4;   // sumsq: rows x N / 1024 slots
    return c;
}

static int prepare(const UnivlGemm* d, GemmArgs& a, int& ksplit, Choice& c, int forced_tile = 0, int forced_nc = 0, bool allow_rect = false,
                   bool in_group = false) {
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "univl_gemm: null descriptor");
    UNIVL_CHECK_ARG(d->dtype == UNIVL_F32 || d->dtype == UNIVL_BF16, UNIVL_EUNSUPPORTED, "univl_gemm: dtype %d", d->dtype);
    UNIVL_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, UNIVL_EINVAL, "univl_gemm: empty problem %dx%dx%d", d->M, d->N, d->K);
    UNIVL_CHECK_ARG(d->A && d->B && (d->C32 || d->C16), UNIVL_EINVAL, "univl_gemm: null operand");
    UNIVL_CHECK_ARG(d->tile == 0 || d->tile == 64 || d->tile == 128 || d->tile == 256 || d->tile == 12864 || d->tile == 64128, UNIVL_EINVAL,
                    "univl_gemm: tile %d (0, 64, 128, 256, 12864, 64128)", d->tile);
    UNIVL_CHECK_ARG(d->stages == 0 || d->stages == 2, UNIVL_EINVAL, "univl_gemm: stages %d (0, 2)", d->stages);
    UNIVL_CHECK_ARG(d->waves == 0 || d->waves == 4 || d->waves == 8, UNIVL_EINVAL, "univl_gemm: waves %d (0, 4, 8)", d->waves);
    const int epc = d->dtype == UNIVL_BF16 ? 8 : 4;
    UNIVL_CHECK_ARG(aligned16(d->A) && aligned16(d->B) && d->lda % epc == 0 && d->ldb % epc == 0, UNIVL_EALIGN,
                    "univl_gemm: operands must be 16-byte aligned with leading dims a multiple of %d (lda=%ld ldb=%ld)",
                    epc, d->lda, d->ldb);
    const int flags = d->flags & ~(UNIVL_GEMM_ATOMIC | UNIVL_GEMM_XCD_MAP | UNIVL_GEMM_PROBE_NOSTORE | UNIVL_GEMM_PROBE_NT_B);     // internal bits are the library's
    UNIVL_CHECK_ARG(!((flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD)) && !d->aux), UNIVL_EINVAL,
                    "univl_gemm: GELU epilogue needs aux");
    // tile choice: 128x128 once the grid fills the chip (>= 256 tiles, measured: +6 % at bs 128 over 384, same at bs 16), else
    // 64x64 for parallelism.  The small tile stages 4 chunks (128 bf16 / 64 f32) per barrier: at M <= a few hundred the kernel
    // is a latency chain of K steps (DMA -> barrier -> ds_read -> MFMA), so fewer, deeper steps win.
    const bool paired = d->A_lo || d->B_lo || d->C16_lo;
    // operand pairs: only the bf16 64 x 64 tile with 128-deep steps carries them (gemm_tile<..., PAIRS>), in the single / rider / fold launches
    UNIVL_CHECK_ARG(!(paired && (in_group || d->trans_a || (forced_tile != 0 && forced_tile != 64) || forced_nc == 2)), UNIVL_EUNSUPPORTED,
                    "univl_gemm: operand pairs only on K-major-A products of the single, rider and LayerNorm-fold launches");
    c = paired ? choose(d, 64, 0, false, false) : choose(d, forced_tile, forced_nc, allow_rect, in_group);
    // deterministic mode (common.h): no split-K -- the slices of a split product meet in fp32 atomics whose order is the hardware's;
    // one workgroup per output tile walks the whole contraction in order (the pre-zeroed arena is simply overwritten)
    ksplit = (d->ksplit < 1 || univl_deterministic()) ? 1 : d->ksplit;
    const int BK = c.tile == 256 ? 128 : (d->dtype == UNIVL_BF16 ? 32 : 16) * c.nc;      // 256 body: K % (128 ksplit) == 0 (fits256)
    // operand pairs: the slices divide the concatenated contraction of nseg terms (gemm_tile)
    const int nseg = 1 + (d->A_lo ? 1 : 0) + (d->B_lo ? 1 : 0);
    if (nseg > 1 || d->C16_lo) {
        UNIVL_CHECK_ARG(d->dtype == UNIVL_BF16 && c.tile != 256 && d->K % BK == 0 && aligned16(d->A_lo) && aligned16(d->B_lo) &&
                            !(d->C16_lo && !d->C16) && (long)nseg * d->K < (1L << 30),
                        UNIVL_EUNSUPPORTED, "univl_gemm: operand pairs need bf16, K (%d) a multiple of the K step (%d), 16-byte aligned lo halves", d->K, BK);
    }
    const long Kv = (long)nseg * d->K;
    int klen = (int)(((Kv + ksplit - 1) / ksplit + BK - 1) / BK * BK);
    ksplit = (int)((Kv + klen - 1) / klen);
    if (ksplit > 1) {
        UNIVL_CHECK_ARG(d->C32 && !d->C16 && !(flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD)), UNIVL_EINVAL,
                        "univl_gemm: split-K needs a pre-zeroed fp32 output and a linear epilogue");
    }
    UNIVL_CHECK_ARG(!(d->dbias && !d->trans_a), UNIVL_EINVAL, "univl_gemm: dbias only with T-major A (wgrad)");
    UNIVL_CHECK_ARG(!((flags & UNIVL_GEMM_ACCUM) && !d->C32), UNIVL_EINVAL, "univl_gemm: ACCUM needs the fp32 output");
    UNIVL_CHECK_ARG(!(d->sumsq && (ksplit > 1 || d->sumsq_rows < 0 || d->sumsq_rows % 128 != 0)), UNIVL_EINVAL,
                    "univl_gemm: sumsq needs ksplit = 1 and sumsq_rows a multiple of 128");
    a.A = d->A; a.B = d->B; a.lda = d->lda; a.ldb = d->ldb; a.M = d->M; a.N = d->N; a.K = d->K;
    a.C32 = d->C32; a.C16 = d->C16; a.ldc = d->ldc; a.bias = d->bias; a.R = d->R; a.ldr = d->ldr;
    a.aux = d->aux; a.ldaux = d->ldaux; a.dbias = d->dbias; a.alpha = d->alpha;
    a.flags = flags | (ksplit > 1 ? UNIVL_GEMM_ATOMIC : 0) | UNIVL_GEMM_XCD_MAP;
#ifdef UNIVL_TRACE
    static const bool probe = getenv("UNIVL_GEMM_PROBE") && atoi(getenv("UNIVL_GEMM_PROBE")) != 0;      // measurement build only
    if (probe) a.flags |= UNIVL_GEMM_PROBE_NOSTORE;
    if (const char* e = getenv("UNIVL_GEMM_NT_B")) { if (atoi(e) != 0) a.flags |= UNIVL_GEMM_PROBE_NT_B; }   // read per call: A/B inside one process
#endif
    a.ksplit_len = klen;
    a.sumsq = d->sumsq; a.sumsq_rows = d->sumsq_rows; a.sumsq_stride = d->sumsq_stride;
    a.gm = GEMM_GM;
    a.A_lo = d->A_lo; a.B_lo = d->B_lo; a.C16_lo = d->C16_lo;
    return UNIVL_OK;
}

// Host-side evaluation of the workgroup -> tile maps above (no device work): what = 0, the plain-grid map of gemm_kernel
// (in: the hardware block index in out[0..2]; out: the tile it computes); what = 1, xcd_run(out[0], nx) -> out[0] (the grouped
// launch's list position); what = 2, tile_of(out[0], ...) (a group member's local tile); what = 3, the local workgroup -> tile map of
// one half of gemm_pair_kernel / gemm_adam_kernel.  tests/test_host_cpu.py proves the
// maps are bijections for every grid shape the plans produce -- a map that is not one would silently skip tiles.
extern "C" int univl_gemm_tile_map(int32_t what, int32_t nx, int32_t ny, int32_t nz, int32_t gm, int32_t* out) {
    UNIVL_CHECK_ARG(out != nullptr && nx > 0 && ny > 0 && nz > 0 && what >= 0 && what <= 4, UNIVL_EINVAL,
                    "univl_gemm_tile_map: what=%d grid=%dx%dx%d", what, nx, ny, nz);
    int bx = out[0], by = out[1], bz = out[2];
    if (what == 0) xcd_tile_grid(nx, ny, nz, gm, bx, by, bz);
    else if (what == 1) bx = xcd_run(out[0], nx);
    else if (what == 2) tile_of(out[0], nx, ny, nz, gm, bx, by, bz);
    else if (what == 4) {                  // gemm_adam_rect_kernel: workgroup out[0] of nx + ny (nx tile slots, ny update workgroups)
        UNIVL_CHECK_ARG(nx % 8 == 0 && ny % 8 == 0 && ny >= 8 && out[0] < nx + ny, UNIVL_EINVAL, "univl_gemm_tile_map: what=4 needs multiples of 8");
        int idx;
        out[1] = rect_rider_role(out[0], nx + ny, nx, idx) ? 1 : 0;
        out[0] = idx;
        return UNIVL_OK;
    }
    else pair_tile(out[0], nx * ny * nz, nx, ny, nz, UNIVL_GEMM_XCD_MAP, gm, bx, by, bz);       // a half of gemm_pair_kernel / gemm_adam_kernel
    out[0] = bx; out[1] = by; out[2] = bz;
    return UNIVL_OK;
}

// Host-side evaluation of the 256 x 256 body's LDS maps (gemm256.h; no device work): tests/test_host_cpu.py rebuilds a half-tile image
// from the DMA map and checks that every fragment read returns the operand elements the MFMA expects, and that the reads are
// bank-conflict free under the LDS's per-instruction lane groups.
//   what 0: out[0..1] <- (row, k) of the FIRST of the 8 elements the DMA drops at lane-linear LDS piece idx (0..1023) of a half-tile
//           image (K-major: 8 consecutive k of one row; T-major: 8 consecutive rows of one k);
//   what 1: out[0..7] <- LDS byte offsets of the fragment reads of lane idx for the 32-row block at `arg` (a multiple of 32):
//           K-major out[ks] (one 16-byte read per k step), T-major out[2 ks], out[2 ks + 1] (two 8-byte transpose reads per k step).
extern "C" int univl_gemm256_layout(int32_t what, int32_t trans, int32_t idx, int32_t arg, int32_t* out) {
    UNIVL_CHECK_ARG(out != nullptr && (what == 0 || what == 1) && idx >= 0 && idx < (what == 0 ? 1024 : 64) && arg >= 0 && arg < 128 &&
                        arg % 32 == 0, UNIVL_EINVAL, "univl_gemm256_layout: what=%d idx=%d arg=%d", what, idx, arg);
    constexpr long LD = 1 << 20;
    if (what == 0) {
        const long o = trans ? Op256<true>::src(idx, LD) : Op256<false>::src(idx, LD);
        if (trans) { out[0] = (int)(o % LD); out[1] = (int)(o / LD); }
        else { out[0] = (int)(o / LD); out[1] = (int)(o % LD); }
        return UNIVL_OK;
    }
    if (trans) {
        const Op256<true>::Lane l = Op256<true>::lane_offsets(idx, arg);
        for (int ks = 0; ks < 4; ++ks) { out[2 * ks] = l.off[0] + 4096 * ks; out[2 * ks + 1] = l.off[0] + 4096 * ks + 1024; }
    } else {
        const Op256<false>::Lane l = Op256<false>::lane_offsets(idx, arg);
        for (int ks = 0; ks < 4; ++ks) { out[ks] = l.off[ks]; out[4 + ks] = -1; }
    }
    return UNIVL_OK;
}

extern "C" int univl_gemm(const UnivlGemm* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    GemmArgs a;
    int ksplit;
    Choice c;
    const int rc = prepare(d, a, ksplit, c, 0, 0, true);
    if (rc != UNIVL_OK) return rc;
    if (d->dtype == UNIVL_F32 && !d->trans_a && !d->trans_b && d->M <= 32 && d->N <= 32 && ksplit == 1 && d->tile == 0 &&
        !d->bias && !d->R && !d->dbias && !d->sumsq && !(d->flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD))) {
        hipLaunchKernelGGL(dot_kernel, dim3(d->N, d->M), dim3(256), 0, stream, a);
        UNIVL_LAUNCH_CHECK();
        return UNIVL_OK;
    }
    const int ta = d->trans_a, tb = d->trans_b;
    if (d->dtype == UNIVL_BF16) {
        if (c.tile == 256) {
            if (!ta && !tb) return launch256<false, false>(a, ksplit, stream);
            if (!ta && tb) return launch256<false, true>(a, ksplit, stream);
            return launch256<true, true>(a, ksplit, stream);
        }
        if (c.tile == 12864) {                       // K-major A only (choose): forward and dgrad products
            if (c.waves == 8) return tb ? launch<__bf16, false, true, 128, 64, 2, 4, 2>(a, ksplit, stream)
                                        : launch<__bf16, false, false, 128, 64, 2, 4, 2>(a, ksplit, stream);
            return tb ? launch<__bf16, false, true, 128, 64, 2, 2, 2>(a, ksplit, stream) : launch<__bf16, false, false, 128, 64, 2, 2, 2>(a, ksplit, stream);
        }
        if (c.tile == 64128) {
            if (c.waves == 8) return tb ? launch<__bf16, false, true, 64, 128, 2, 2, 4>(a, ksplit, stream)
                                        : launch<__bf16, false, false, 64, 128, 2, 2, 4>(a, ksplit, stream);
            return tb ? launch<__bf16, false, true, 64, 128, 2, 2, 2>(a, ksplit, stream) : launch<__bf16, false, false, 64, 128, 2, 2, 2>(a, ksplit, stream);
        }
        if (c.tile == 128) {
            if (c.waves == 8) return dispatch_trans<__bf16, 128, 128, 2, 2, 4>(a, ta, tb, ksplit, stream);
            return dispatch_trans<__bf16, 128, 128, 2>(a, ta, tb, ksplit, stream);
        }
        if (c.nc == 6) return launch<__bf16, true, true, 64, 64, 6>(a, ksplit, stream);
        if (c.nc == 2) {
            if (c.waves == 8) return dispatch_trans<__bf16, 64, 64, 2, 2, 4>(a, ta, tb, ksplit, stream);
            return dispatch_trans<__bf16, 64, 64, 2>(a, ta, tb, ksplit, stream);
        }
        if (c.waves == 8) return dispatch_trans<__bf16, 64, 64, 4, 2, 4>(a, ta, tb, ksplit, stream);
        return dispatch_trans<__bf16, 64, 64, 4>(a, ta, tb, ksplit, stream);
    }
    if (c.tile == 128) return dispatch_trans<float, 128, 128, 2>(a, ta, tb, ksplit, stream);
    return dispatch_trans<float, 64, 64, 4>(a, ta, tb, ksplit, stream);
}

static bool has_pairs(const UnivlGemm* d) { return d->A_lo != nullptr || d->B_lo != nullptr || d->C16_lo != nullptr; }

static int pair_impl(const UnivlGemm* dgrad, const UnivlGemm* wgrad, const UnivlLayerNorm* ln, int32_t* counters, int32_t dry_run,
                     hipStream_t stream) {
    UNIVL_CHECK_ARG(dgrad != nullptr && wgrad != nullptr, UNIVL_EINVAL, "univl_gemm_pair: null descriptor");
    UNIVL_CHECK_ARG(!has_pairs(dgrad) && !has_pairs(wgrad), UNIVL_EUNSUPPORTED, "univl_gemm_pair: operand pairs are carried by univl_gemm / the rider and fold launches only");
    PairArgs a;
    int ksd, ksw;
    Choice cd, cw;
    int rc = prepare(dgrad, a.d, ksd, cd);
    if (rc != UNIVL_OK) return rc;
    rc = prepare(wgrad, a.w, ksw, cw);
    if (rc != UNIVL_OK) return rc;
    // both products must be ones the single launch would run on the 64 tile (sizes below the big-tile regime)
    UNIVL_CHECK_ARG(dgrad->dtype == UNIVL_BF16 && wgrad->dtype == UNIVL_BF16 && !dgrad->trans_a && dgrad->trans_b && wgrad->trans_a &&
                        wgrad->trans_b && cd.tile == 64 && cw.tile == 64 && cd.nc == 4 && (cw.nc == 4 || cw.nc == 6) && !dgrad->sumsq,
                    UNIVL_EUNSUPPORTED, "univl_gemm_pair: needs a bf16 (K-major, T-major) product and a bf16 (T-major, T-major) product on the 64 tile");
    // the dgrad body: rectangular from 384 tokens on (gemm_pair_kernel) -- unless the dgrad slice is deeper than 1536: its workgroups
    // then walk more than 24 of the 64-deep K steps alone (a 3072-deep unsplit dgrad: 48 steps, 42 us at 768 tokens against 22 us for
    // the square form's 24 steps of 128; profiles/r04d_bench_lines.txt: +7 % per step on the caption / FT-Align configurations before
    // this rule).  The weight-gradient body: 128 x 64 tiles wherever its out-feature count allows (multiples of 128 rows keep the
    // fused gradient-norm slots exact), 64-deep K steps or the single 192-deep one.  An explicit tile in either descriptor keeps
    // round 2's square form (the A/B switch of engine.EncoderStack, and the form the bit-identity test compares against).
    const long dslice = ((long)dgrad->K + ksd - 1) / ksd;
    const bool plain = dgrad->tile == 0 && wgrad->tile == 0;
    const bool drect = plain && dgrad->M >= 384 && dslice <= 1536;
    const bool wrect = plain && wgrad->M % 128 == 0 && (!wgrad->sumsq || wgrad->sumsq_rows % 128 == 0);
    const bool wone = cw.nc == 6;                     // K == 192, unsplit: the single-step body
    if (drect) {
        rc = prepare(dgrad, a.d, ksd, cd, 64, 2);
        if (rc != UNIVL_OK) return rc;
        UNIVL_CHECK_ARG(cd.nc == 2, UNIVL_EUNSUPPORTED, "univl_gemm_pair: rectangular dgrad body needs 64-deep K steps");
    }
    if (wrect && !wone) {
        rc = prepare(wgrad, a.w, ksw, cw, 64, 2);
        if (rc != UNIVL_OK) return rc;
        UNIVL_CHECK_ARG(cw.nc == 2, UNIVL_EUNSUPPORTED, "univl_gemm_pair: rectangular weight-gradient body needs 64-deep K steps");
    }
    const int dbn = drect ? 128 : 64, wbm = wrect ? 128 : 64;
    a.dnx = (dgrad->N + dbn - 1) / dbn; a.dny = (dgrad->M + 63) / 64; a.dnz = ksd;
    a.wnx = (wgrad->N + 63) / 64; a.wny = (wgrad->M + wbm - 1) / wbm; a.wnz = ksw;
    a.nd = a.dnx * a.dny * a.dnz;
    a.nd_pad = (a.nd + 7) / 8 * 8;
    a.nw = a.wnx * a.wny * a.wnz;
    // the weight gradient's bias gradient (column sums of its T-major A over the contraction) as workgroups of their own between the
    // two products instead of an LDS walk inside the product's column-0 workgroups (colsum_tile); up to 2048 tokens -- beyond, the
    // plans take their bias gradients elsewhere (engine.EncoderStack: big-tile weight gradients) and a 64-column workgroup would
    // stream megabytes alone
    a.cs = ColsumArgs{nullptr, 0, 0, 0, nullptr, 0, 0};
    if (wgrad->dbias && wgrad->K <= 2048 && wgrad->M % 8 == 0) {
        a.cs.x = reinterpret_cast<const __bf16*>(wgrad->A);
        a.cs.ld = wgrad->lda;
        a.cs.rows = wgrad->K;
        a.cs.n = wgrad->M;
        a.cs.out = wgrad->dbias;
        a.cs.n_tiles = (wgrad->M + 63) / 64;
        a.cs.n_pad = (a.cs.n_tiles + 7) / 8 * 8;
        a.w.dbias = nullptr;
    }
    if (ln != nullptr) {
        // the LayerNorm backward whose upstream gradient this dgrad produces, finished by the dgrad's last workgroups per 64-row block
        // (ln_fold_bwd): fp32 output over N = 768 columns, pre-zeroed (atomics); the rectangular dgrad body (384+ tokens) only together
        // with the 128 x 64 weight-gradient body (round 5: gemm_pair_kernel<true, 2, true>, two workgroups per compute unit)
        UNIVL_CHECK_ARG(!univl_deterministic() && (!drect || (wrect && !wone)) && counters != nullptr && dgrad->C32 && !dgrad->C16 && dgrad->N == 768 &&
                            dgrad->ldc == 768 && !(dgrad->flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD | UNIVL_GEMM_ACCUM)) &&
                            a.dnx * a.dnz >= LN_SHARE && a.dny <= LN_FOLD_MAX_BLOCKS && ln->dtype == UNIVL_BF16 && ln->N == 768 && ln->rows == dgrad->M &&
                            ln->dout == (const float*)dgrad->C32 && ln->gamma && ln->y && ln->stats && !ln->dpos &&
                            (const void*)ln->dx32 != (const void*)dgrad->C32,
                        UNIVL_EUNSUPPORTED, "univl_gemm_pair_ln: not a (dgrad, LayerNorm backward) pair this launch carries");
        a.d.flags |= UNIVL_GEMM_ATOMIC;
    }
    if (dry_run) return UNIVL_OK;
    if (ln != nullptr) {
        LnFold f;
        f.ln = *ln;
        f.counters = counters;
        if (drect) return launch_pair<true, 2, true>(a, stream, &f);
        if (wrect) return wone ? launch_pair<false, 3, true>(a, stream, &f) : launch_pair<false, 2, true>(a, stream, &f);
        return wone ? launch_pair<false, 1, true>(a, stream, &f) : launch_pair<false, 0, true>(a, stream, &f);
    }
    if (drect) return wrect ? launch_pair<true, 2>(a, stream) : launch_pair<true, 0>(a, stream);
    if (wrect) return wone ? launch_pair<false, 3>(a, stream) : launch_pair<false, 2>(a, stream);
    return wone ? launch_pair<false, 1>(a, stream) : launch_pair<false, 0>(a, stream);
}

extern "C" int univl_gemm_pair(const UnivlGemm* dgrad, const UnivlGemm* wgrad, int32_t dry_run, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    return pair_impl(dgrad, wgrad, nullptr, nullptr, dry_run, stream);
}

// univl_gemm_pair with the LayerNorm BACKWARD that consumes the dgrad's fp32 output (ln->dout == dgrad->C32) finished inside the launch;
// counters as for univl_gemm_ln.  UNIVL_EUNSUPPORTED where the launch does not carry it (callers: univl_gemm_pair + univl_layernorm_bwd).
extern "C" int univl_gemm_pair_ln(const UnivlGemm* dgrad, const UnivlGemm* wgrad, const UnivlLayerNorm* ln, int32_t* counters,
                                  int32_t dry_run, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(ln != nullptr && counters != nullptr, UNIVL_EINVAL, "univl_gemm_pair_ln: null argument");
    return pair_impl(dgrad, wgrad, ln, counters, dry_run, stream);
}

// univl_attention_bwd with the attention-output dgrad that produces its upstream gradient computed inside the launch, and (optionally)
// the weight gradient of the same projection riding in it (attn_bwd_odgrad_kernel above).  bf16, sequences of at most 64 positions,
// odgrad: K-major dY [tokens, K] x T-major W [K, H * 64] -> bf16 at->dout (which is NOT written: nothing else reads it), K a multiple
// of 128, no epilogue; owgrad: what univl_gemm_pair would take on the 64 tile (may be NULL).  UNIVL_EUNSUPPORTED otherwise: callers
// enqueue univl_gemm_pair / univl_gemm + univl_attention_bwd.
extern "C" int univl_attention_bwd_fused(const UnivlAttention* at, const UnivlGemm* odgrad, const UnivlGemm* owgrad, int32_t dry_run,
                                         hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(at != nullptr && odgrad != nullptr, UNIVL_EINVAL, "univl_attention_bwd_fused: null descriptor");
    int rc = attn_check(at, "univl_attention_bwd_fused", true);
    if (rc) return rc;
    using C = AttnCfg<__bf16>;
    const int Sk_pad = (at->Sk + 31) / 32 * 32, Sq_pad = (at->Sq + 31) / 32 * 32;
    const UnivlGemm* g = odgrad;
    UNIVL_CHECK_ARG(at->dtype == UNIVL_DT_BF16 && Sk_pad <= 64 && Sq_pad <= 64 && g->dtype == UNIVL_BF16 && !g->trans_a && g->trans_b &&
                        g->M == at->B * at->Sq && g->N == at->H * 64 && g->K % 128 == 0 && g->K >= 128 && g->C16 == at->dout && !g->C32 &&
                        g->ldc == at->lddo && !g->bias && !g->R && !g->dbias && !g->sumsq && g->alpha == 1.0f && g->ksplit <= 1 &&
                        (g->flags & ~(UNIVL_GEMM_XCD_MAP)) == 0 && aligned16(g->A) && aligned16(g->B) && g->lda % 8 == 0 && g->ldb % 8 == 0 &&
                        !g->A_lo && !g->B_lo && !g->C16_lo,
                    UNIVL_EUNSUPPORTED, "univl_attention_bwd_fused: not an (attention backward, attention-output dgrad) pair this launch carries");
    AttnFusedArgs a;
    a.at = *at;
    a.Sk_pad = Sk_pad; a.Sq_pad = Sq_pad;
    a.dY = reinterpret_cast<const __bf16*>(g->A); a.lddy = g->lda;
    a.W = reinterpret_cast<const __bf16*>(g->B); a.ldw = g->ldb;
    a.K = g->K;
    a.n_attn = 2 * at->B * at->H;
    a.n_attn_pad = (a.n_attn + 7) / 8 * 8;
    a.nw = 0; a.wnx = a.wny = a.wnz = 1;
    bool wone = false;
    if (owgrad != nullptr) {
        int ksw;
        Choice cw;
        rc = prepare(owgrad, a.w, ksw, cw);
        if (rc != UNIVL_OK) return rc;
        UNIVL_CHECK_ARG(owgrad->dtype == UNIVL_BF16 && owgrad->trans_a && owgrad->trans_b && cw.tile == 64 && (cw.nc == 4 || cw.nc == 6) &&
                            !owgrad->dbias, UNIVL_EUNSUPPORTED, "univl_attention_bwd_fused: the riding weight gradient must be a bf16 (T-major, T-major) product on the 64 tile");
        wone = cw.nc == 6;
        a.wnx = (owgrad->N + 63) / 64; a.wny = (owgrad->M + 63) / 64; a.wnz = ksw;
        a.nw = a.wnx * a.wny * a.wnz;
    } else {
        a.w = GemmArgs{};
    }
    if (dry_run) return UNIVL_OK;
    constexpr bool DUAL = true;
    // LDS: the product's two stages (64 KB), then -- in the same bytes -- the attention images + the dO tile of role 0
    const size_t smemA = (size_t)Sk_pad * (C::PT + 2 * C::PK) * 2 + Sk_pad * 4 + 64 * C::PK * 2;
    const size_t smemB = (size_t)Sq_pad * (2 * C::PT + 2 * C::PK) * 2 + (Sk_pad + 2 * Sq_pad) * 4;
    size_t smem = 4 * (size_t)64 * 128 * 2;
    smem = smem > smemA ? smem : smemA;
    smem = smem > smemB ? smem : smemB;
    constexpr int TR = (64 * 8 + 511) / 512;
    static bool done4[UNIVL_MAX_DEVICES] = {}, done6[UNIVL_MAX_DEVICES] = {};
    const dim3 grid(a.n_attn_pad + a.nw);
    if (wone) {
        univl_allow_lds(attn_bwd_odgrad_kernel<TR, DUAL, 6>, 160 * 1024, done6);
        hipLaunchKernelGGL((attn_bwd_odgrad_kernel<TR, DUAL, 6>), grid, dim3(256), smem, stream, a);
    } else {
        univl_allow_lds(attn_bwd_odgrad_kernel<TR, DUAL, 4>, 160 * 1024, done4);
        hipLaunchKernelGGL((attn_bwd_odgrad_kernel<TR, DUAL, 4>), grid, dim3(256), smem, stream, a);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// univl_attention_fwd with the q / k / v projection that produces its inputs computed inside the launch (attn_fwd_qkv_kernel above), and
// optionally BertAdam chunks riding like in univl_gemm_rider.  bf16 self-attention over at most 64 positions; qkv: K-major x K-major,
// [tokens, K] x [3 * H * 64, K] -> bf16 C16 with at->q / k / v = C16 + 0 / H * 64 / 2 * H * 64 and ldq = ldk = ldv = ldc, K a multiple
// of 64, bias optional, no other epilogue.  Results (the qkv buffer, the attention output, the log-sum-exp) are bit-identical to
// univl_gemm + univl_attention_fwd.  UNIVL_EUNSUPPORTED otherwise; adam may be NULL (chunk_count 0).
static const size_t ATTN_FWD_FUSED_SMEM = 2 * (size_t)(64 + 192) * 64 * sizeof(__bf16);
static const size_t ATTN_FWD_FUSED_SMEM_BIG = 2 * (size_t)(128 + 192) * 64 * sizeof(__bf16);      // 80 KB: the images of 128 positions (57 KB) fit inside
static void attn_fwd_fused_allow_lds() {
    static bool done_nt[UNIVL_MAX_DEVICES] = {}, done_t[UNIVL_MAX_DEVICES] = {}, done_ntb[UNIVL_MAX_DEVICES] = {}, done_tb[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(attn_fwd_qkv_kernel<true, false>, ATTN_FWD_FUSED_SMEM, done_nt);
    univl_allow_lds(attn_fwd_qkv_kernel<false, false>, ATTN_FWD_FUSED_SMEM, done_t);
    univl_allow_lds(attn_fwd_qkv_kernel<true, true>, ATTN_FWD_FUSED_SMEM_BIG, done_ntb);
    univl_allow_lds(attn_fwd_qkv_kernel<false, true>, ATTN_FWD_FUSED_SMEM_BIG, done_tb);
}

extern "C" int univl_attention_fwd_fused(const UnivlAttention* at, const UnivlGemm* qkv, const UnivlAdam* adam, int32_t chunk_begin,
                                         int32_t chunk_count, int32_t max_blocks, int32_t dry_run, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(at != nullptr && qkv != nullptr, UNIVL_EINVAL, "univl_attention_fwd_fused: null descriptor");
    UNIVL_CHECK_ARG(chunk_count == 0 || (adam != nullptr && adam->p && adam->g && adam->m && adam->v && adam->segs && adam->chunk_seg &&
                                         adam->chunk_off && adam->chunk_len && adam->seg_scalars && chunk_begin >= 0 && chunk_count > 0 &&
                                         chunk_begin + chunk_count <= adam->nchunk),
                    UNIVL_EINVAL, "univl_attention_fwd_fused: chunks [%d, +%d)", chunk_begin, chunk_count);
    int rc = attn_check(at, "univl_attention_fwd_fused", false);
    if (rc) return rc;
    const int Sk_pad = (at->Sk + 31) / 32 * 32;
    const UnivlGemm* g = qkv;
    const long hd = (long)at->H * 64;
    const __bf16* c16 = reinterpret_cast<const __bf16*>(g->C16);
    UNIVL_CHECK_ARG(at->dtype == UNIVL_DT_BF16 && at->Sq == at->Sk && Sk_pad <= 128 && !at->causal && at->bsk == 0 && at->bsv == 0 &&
                        g->dtype == UNIVL_BF16 && !g->trans_a && !g->trans_b && g->M == at->B * at->Sq && g->N == 3 * hd && g->K % 64 == 0 &&
                        g->K >= 64 && c16 != nullptr && !g->C32 && !g->R && !g->dbias && !g->sumsq && !g->aux && g->alpha == 1.0f &&
                        g->ksplit <= 1 && (g->flags & ~(UNIVL_GEMM_XCD_MAP)) == 0 && aligned16(g->A) && aligned16(g->B) && g->lda % 8 == 0 &&
                        aligned16(g->A_lo) && aligned16(g->B_lo) && !g->C16_lo &&
                        g->ldb % 8 == 0 && at->q == (const void*)c16 && at->k == (const void*)(c16 + hd) && at->v == (const void*)(c16 + 2 * hd) &&
                        at->ldq == g->ldc && at->ldk == g->ldc && at->ldv == g->ldc,
                    UNIVL_EUNSUPPORTED, "univl_attention_fwd_fused: not a (q|k|v projection, self-attention) pair this launch carries");
    if (dry_run) return UNIVL_OK;
    AttnFwdFusedArgs a;
    a.at = *at;
    a.Sk_pad = Sk_pad;
    a.X = reinterpret_cast<const __bf16*>(g->A); a.ldx = g->lda;
    a.W = reinterpret_cast<const __bf16*>(g->B); a.ldw = g->ldb;
    a.bias = g->bias;
    a.qkv = reinterpret_cast<__bf16*>(g->C16); a.ldqkv = g->ldc;
    a.K = g->K;
    a.X_lo = reinterpret_cast<const __bf16*>(g->A_lo); a.W_lo = reinterpret_cast<const __bf16*>(g->B_lo);
    const bool big = Sk_pad > 64;                      // 65 .. 128 positions: two workgroups (query blocks) per (batch row, head)
    a.n_attn = at->B * at->H * (big ? 2 : 1);
    a.n_attn_pad = (a.n_attn + 7) / 8 * 8;
    const int nb = chunk_count <= 0 ? 0 : ((max_blocks > 0 && max_blocks < chunk_count) ? max_blocks : chunk_count);
    UnivlAdam none = {};
    const UnivlAdam& ad = chunk_count > 0 ? *adam : none;
    attn_fwd_fused_allow_lds();
    const int cb = chunk_begin, ce = chunk_begin + (chunk_count > 0 ? chunk_count : 0);
    const dim3 grid(a.n_attn_pad + nb);
    if (big) {
        if (univl_adam_nt()) hipLaunchKernelGGL((attn_fwd_qkv_kernel<true, true>), grid, dim3(512), ATTN_FWD_FUSED_SMEM_BIG, stream, a, ad, cb, ce);
        else hipLaunchKernelGGL((attn_fwd_qkv_kernel<false, true>), grid, dim3(512), ATTN_FWD_FUSED_SMEM_BIG, stream, a, ad, cb, ce);
    } else {
        if (univl_adam_nt()) hipLaunchKernelGGL((attn_fwd_qkv_kernel<true, false>), grid, dim3(512), ATTN_FWD_FUSED_SMEM, stream, a, ad, cb, ce);
        else hipLaunchKernelGGL((attn_fwd_qkv_kernel<false, false>), grid, dim3(512), ATTN_FWD_FUSED_SMEM, stream, a, ad, cb, ce);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

static const size_t RIDER_SMEM = 4 * (size_t)Tile<__bf16, false, 64, 128, 512>::BYTES;          // two stages of A and B
static void rider_allow_lds() {
    static bool done_nt[UNIVL_MAX_DEVICES] = {}, done_t[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(gemm_adam_kernel<true>, RIDER_SMEM, done_nt);
    univl_allow_lds(gemm_adam_kernel<false>, RIDER_SMEM, done_t);
}

// 0: univl_gemm_rider enqueues product and update one after the other; 1: gemm_adam_kernel (64 x 64 tile); 2: gemm_adam_rect_kernel
static int rider_form(const UnivlGemm* gemm, const Choice& c) {
    if (gemm->dtype != UNIVL_BF16 || gemm->trans_a || gemm->trans_b || gemm->sumsq || gemm->dbias) return 0;
    if (c.tile == 64 && c.nc == 4) return 1;
    if (c.tile == 64128 && c.waves == 8) return 2;
    return 0;
}

// The tile of a riding product: what the square-tile choice says where that is the 64 x 64 tile (a few hundred tokens: the forms rounds
// 3 / 4 measured), else what univl_gemm itself would pick (rectangular tiles allowed).
static int rider_prepare(const UnivlGemm* gemm, GemmArgs& a, int& ks, int& form) {
    Choice c;
    int rc = prepare(gemm, a, ks, c);
    if (rc != UNIVL_OK) return rc;
    form = rider_form(gemm, c);
    if (form == 0) {
        rc = prepare(gemm, a, ks, c, 0, 0, true);
        if (rc != UNIVL_OK) return rc;
        form = rider_form(gemm, c);
    }
    return UNIVL_OK;
}

// Host-side question (no device work): does univl_gemm_rider carry optimizer chunks INSIDE this product's launch (1) or would it enqueue
// the update behind it (0)?  A host distributes a layer's chunks over the products that answer 1.
extern "C" int univl_gemm_rider_fits(const UnivlGemm* gemm) {
    UNIVL_CHECK_ARG(gemm != nullptr, UNIVL_EINVAL, "univl_gemm_rider_fits: null descriptor");
    GemmArgs a;
    int ks, form;
    const int rc = rider_prepare(gemm, a, ks, form);
    if (rc != UNIVL_OK) return rc;
    return form != 0 ? 1 : 0;
}

extern "C" int univl_gemm_rider(const UnivlGemm* gemm, const UnivlAdam* adam, int32_t chunk_begin, int32_t chunk_count,
                                int32_t max_blocks, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(gemm != nullptr && adam != nullptr, UNIVL_EINVAL, "univl_gemm_rider: null descriptor");
    UNIVL_CHECK_ARG(adam->p && adam->g && adam->m && adam->v && adam->segs && adam->chunk_seg && adam->chunk_off && adam->chunk_len &&
                        adam->seg_scalars && adam->nchunk > 0 && chunk_begin >= 0 && chunk_count >= 0 &&
                        chunk_begin + chunk_count <= adam->nchunk,
                    UNIVL_EINVAL, "univl_gemm_rider: chunks [%d, +%d) of %d", chunk_begin, chunk_count, adam ? adam->nchunk : 0);
    GemmArgs a;
    int ks, fits;
    int rc = rider_prepare(gemm, a, ks, fits);
    if (rc != UNIVL_OK) return rc;
    if (fits == 2 && adam->p16_lo != nullptr) fits = 0;      // the 64 x 128 rider kernel does not keep the lo half of the shadow pair (adam_chunk<.., LO = false>)
    if (!fits || chunk_count == 0) {          // not a product these kernels carry: the two launches one after the other (same result)
        const int r1 = univl_gemm(gemm, stream);
        if (r1 != UNIVL_OK || chunk_count == 0) return r1;
        return univl_bert_adam_range(adam, chunk_begin, chunk_count, 0, max_blocks, stream);
    }
    const int nx = (gemm->N + (fits == 2 ? 127 : 63)) / (fits == 2 ? 128 : 64), ny = (gemm->M + 63) / 64;
    const int nd = nx * ny * ks, nd_pad = (nd + 7) / 8 * 8;
    const int nb = (max_blocks > 0 && max_blocks < chunk_count) ? max_blocks : chunk_count;
    const bool nt = univl_adam_nt();              // optim.hip: the one switch of the update's cache policy (UNIVL_ADAM_NT)
    if (fits == 2) {                              // the 64 x 128 tile: 48 KB of LDS, no opt-in
        constexpr size_t smem = 2 * (size_t)(64 + 128) * 64 * sizeof(__bf16);
        const int nb8 = (nb + 7) / 8 * 8;         // whole groups of 8 (the kernel's interleaving); surplus workgroups find no chunk
        if (nt) hipLaunchKernelGGL(gemm_adam_rect_kernel<true>, dim3(nd_pad + nb8), dim3(512), smem, stream, a, nd, nd_pad, nx, ny, ks, *adam,
                                   chunk_begin, chunk_begin + chunk_count);
        else hipLaunchKernelGGL(gemm_adam_rect_kernel<false>, dim3(nd_pad + nb8), dim3(512), smem, stream, a, nd, nd_pad, nx, ny, ks, *adam,
                                chunk_begin, chunk_begin + chunk_count);
        UNIVL_LAUNCH_CHECK();
        return UNIVL_OK;
    }
    rider_allow_lds();
    if (nt) {
        hipLaunchKernelGGL(gemm_adam_kernel<true>, dim3(nd_pad + nb), dim3(512), RIDER_SMEM, stream, a, nd, nd_pad, nx, ny, ks, *adam,
                           chunk_begin, chunk_begin + chunk_count);
    } else {
        hipLaunchKernelGGL(gemm_adam_kernel<false>, dim3(nd_pad + nb), dim3(512), RIDER_SMEM, stream, a, nd, nd_pad, nx, ny, ks, *adam,
                           chunk_begin, chunk_begin + chunk_count);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// The same launch with the LayerNorm that consumes the product's fp32 output folded in (gemm_ln_kernel); adam may be null (no chunks).
// The product must be one the rider kernel carries (bf16, K-major operands, the 64 x 64 tile) with fp32 output = ln->x over ln->rows
// = M rows of N = 768; the LayerNorm a bf16-output, fp32-input, non-positional one.  Returns UNIVL_EUNSUPPORTED otherwise (callers
// then enqueue the two launches), and in deterministic mode (the fold's row sums meet in hardware order like any split-K product's).
extern "C" int univl_gemm_ln(const UnivlGemm* gemm, const UnivlLayerNorm* ln, int32_t* counters, const UnivlAdam* adam,
                             int32_t chunk_begin, int32_t chunk_count, int32_t max_blocks, int32_t dry_run, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(gemm != nullptr && ln != nullptr && counters != nullptr, UNIVL_EINVAL, "univl_gemm_ln: null argument");
    UNIVL_CHECK_ARG(chunk_count == 0 || (adam != nullptr && adam->p && adam->g && adam->m && adam->v && adam->segs && adam->chunk_seg &&
                                         adam->chunk_off && adam->chunk_len && adam->seg_scalars && chunk_begin >= 0 && chunk_count > 0 &&
                                         chunk_begin + chunk_count <= adam->nchunk),
                    UNIVL_EINVAL, "univl_gemm_ln: chunks [%d, +%d)", chunk_begin, chunk_count);
    GemmArgs a;
    int ks;
    Choice c;
    const int rc = prepare(gemm, a, ks, c);
    if (rc != UNIVL_OK) return rc;
    const int nx = (gemm->N + 63) / 64, ny = (gemm->M + 63) / 64;
    UNIVL_CHECK_ARG(!univl_deterministic() && gemm->dtype == UNIVL_BF16 && !gemm->trans_a && !gemm->trans_b && c.tile == 64 && c.nc == 4 &&
                        !gemm->sumsq && !gemm->dbias && gemm->C32 && !gemm->C16 && gemm->N == 768 && gemm->ldc == 768 &&
                        !(gemm->flags & (UNIVL_GEMM_GELU_FWD | UNIVL_GEMM_GELU_BWD | UNIVL_GEMM_ACCUM)) && nx * ks >= LN_SHARE &&
                        ny <= LN_FOLD_MAX_BLOCKS && ln->dtype == UNIVL_BF16 && ln->N == 768 && ln->rows == gemm->M && !ln->x_f64 &&
                        ln->x == (const void*)gemm->C32 && ln->gamma && ln->beta && (ln->out16 || ln->out32),
                    UNIVL_EUNSUPPORTED, "univl_gemm_ln: not a (product, LayerNorm) pair this launch carries");
    if (dry_run) return UNIVL_OK;
    a.flags |= UNIVL_GEMM_ATOMIC;            // every contribution an fp32 atomic into the pre-zeroed output, also when the product is not split
    const int nd = nx * ny * ks, nd_pad = (nd + 7) / 8 * 8;
    const int nb = chunk_count <= 0 ? 0 : ((max_blocks > 0 && max_blocks < chunk_count) ? max_blocks : chunk_count);
    UnivlAdam none = {};
    const UnivlAdam& ad = chunk_count > 0 ? *adam : none;
    LnFold f;
    f.ln = *ln;
    f.counters = counters;
    static bool done_nt[UNIVL_MAX_DEVICES] = {}, done_t[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(gemm_ln_kernel<true>, RIDER_SMEM, done_nt);
    univl_allow_lds(gemm_ln_kernel<false>, RIDER_SMEM, done_t);
    if (univl_adam_nt()) {
        hipLaunchKernelGGL(gemm_ln_kernel<true>, dim3(nd_pad + nb), dim3(512), RIDER_SMEM, stream, a, nd, nd_pad, nx, ny, ks, ad,
                           chunk_begin, chunk_begin + (chunk_count > 0 ? chunk_count : 0), f);
    } else {
        hipLaunchKernelGGL(gemm_ln_kernel<false>, dim3(nd_pad + nb), dim3(512), RIDER_SMEM, stream, a, nd, nd_pad, nx, ny, ks, ad,
                           chunk_begin, chunk_begin + (chunk_count > 0 ? chunk_count : 0), f);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// The rider kernels' large-LDS opt-in, outside any stream capture: their first launch happens INSIDE the capture of the pipelined
// training step (the eager iteration before it has no pending update to carry).
extern "C" int univl_gemm_rider_prime(hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    rider_allow_lds();
    attn_fwd_fused_allow_lds();
    static bool done_nt[UNIVL_MAX_DEVICES] = {}, done_t[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(gemm_ln_kernel<true>, RIDER_SMEM, done_nt);
    univl_allow_lds(gemm_ln_kernel<false>, RIDER_SMEM, done_t);
    return UNIVL_OK;
}

#ifdef UNIVL_PROTO
// ---------------------------------------------------------------------------------------------------------------------------------
// MEASUREMENT PROTOTYPE (round 6; `python univl_amd/build.py --variant proto -DUNIVL_PROTO`, scripts/mb_layer_proto.py; never in the product
// library): the FFN half of an encoder layer's forward -- attention-output product + LayerNorm, FFN1 + GELU, FFN2 + LayerNorm
// (module_bert.py:207-211, 233-250), three launches in the plans -- as ONE launch whose later products wait on row-block flags instead of
// kernel boundaries (VERDICT r5 next 2 asked for a one-layer prototype with a kill criterion).  Workgroup roles in dispatch order: the
// attention-output tiles (their last arrivals per 64-row block finish LayerNorm 1 and raise flags1[block] once per sharer), the FFN1 tiles
// (wait for flags1[block] == LN_SHARE, raise flags2[block]), the FFN2 tiles (wait for flags2[block] == FFN1 column tiles, fold LayerNorm
// 2).  Producers always have lower workgroup ids than their consumers and the whole grid is resident (<= 512 workgroups), so the waits
// cannot deadlock; every wait gives up after 2^22 polls.  The hand-offs carry NO write-through / L2-bypass: a consumer on another XCD may
// read stale operand lines, the RESULTS ARE NOT VALID -- the timing is a lower bound on what a correct version (sc1 stores of the operands
// + sc1 LDS-DMA) would take.
struct ProtoArgs {
    int n_o, n_o_pad, onx, ony, onz;
    int n_f1, n_f1_pad, f1nx, f1ny;
    int n_f2, f2nx, f2ny, f2nz;
    int* flags1; int* flags2;
};

__device__ __forceinline__ void proto_raise(int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void proto_wait(int* flag, int need) {
    if (threadIdx.x == 0) {
        int polls = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++polls < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(512, 2) void layer_ffn_proto_kernel(GemmArgs o, GemmArgs f1, GemmArgs f2, LnFold ln1, LnFold ln2, ProtoArgs q) {
    const int w0 = blockIdx.x;
    int bx, by, bz;
    if (w0 < q.n_o_pad) {
        if (w0 >= q.n_o) return;
        pair_tile(w0, q.n_o, q.onx, q.ony, q.onz, o.flags, o.gm, bx, by, bz);
        gemm_tile<__bf16, false, false, 64, 64, 4, 2, 4>(o, bx, by, bz, q.onz);
        int* cnt = ln1.counters + 2 * by;
        const int slab = fold_arrive(cnt, q.onx * q.onz);
        if (slab < 0) return;
        const int row = by * 64 + slab * 8 + (int)(threadIdx.x >> 6);
        if (row < o.M) ln_fwd_row<768, __bf16, false, true>(ln1.ln, row, (int)(threadIdx.x & 63));
        proto_raise(q.flags1 + by);
        fold_leave(cnt);
    } else if (w0 < q.n_o_pad + q.n_f1_pad) {
        const int t = w0 - q.n_o_pad;
        if (t >= q.n_f1) return;
        pair_tile(t, q.n_f1, q.f1nx, q.f1ny, 1, f1.flags, f1.gm, bx, by, bz);
        proto_wait(q.flags1 + by, LN_SHARE);
        gemm_tile<__bf16, false, false, 64, 64, 4, 2, 4>(f1, bx, by, 0, 1);
        proto_raise(q.flags2 + by);
    } else {
        const int t = w0 - q.n_o_pad - q.n_f1_pad;
        if (t >= q.n_f2) return;
        pair_tile(t, q.n_f2, q.f2nx, q.f2ny, q.f2nz, f2.flags, f2.gm, bx, by, bz);
        proto_wait(q.flags2 + by, q.f1nx);
        gemm_tile<__bf16, false, false, 64, 64, 4, 2, 4>(f2, bx, by, bz, q.f2nz);
        ln_fold(ln2, by, q.f2nx * q.f2nz, f2.M);
    }
}

// o / f2: fp32 outputs into zeroed buffers (= ln1->x / ln2->x), f1: the GELU product; ctr1 / ctr2: the folds' arrival counters (zero);
// flags: 2 * ceil(M / 64) ints, ZERO before every launch.
extern "C" int univl_proto_layer_ffn(const UnivlGemm* o, const UnivlLayerNorm* ln1, const UnivlGemm* f1, const UnivlGemm* f2,
                                     const UnivlLayerNorm* ln2, int32_t* ctr1, int32_t* ctr2, int32_t* flags, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    GemmArgs ao, a1, a2;
    int kso, ks1, ks2;
    Choice co, c1, c2;
    int rc = prepare(o, ao, kso, co);
    if (rc) return rc;
    rc = prepare(f1, a1, ks1, c1);
    if (rc) return rc;
    rc = prepare(f2, a2, ks2, c2);
    if (rc) return rc;
    UNIVL_CHECK_ARG(co.tile == 64 && c1.tile == 64 && c2.tile == 64 && co.nc == 4 && c1.nc == 4 && c2.nc == 4 && ks1 == 1 && o->N == 768 && f2->N == 768 &&
                        o->M == f1->M && o->M == f2->M && o->M <= 64 * LN_FOLD_MAX_BLOCKS && ctr1 && ctr2 && flags, UNIVL_EUNSUPPORTED,
                    "univl_proto_layer_ffn: shapes");
    ao.flags |= UNIVL_GEMM_ATOMIC;
    a2.flags |= UNIVL_GEMM_ATOMIC;
    ProtoArgs q;
    const int ny = (o->M + 63) / 64;
    q.onx = 12; q.ony = ny; q.onz = kso; q.n_o = 12 * ny * kso; q.n_o_pad = (q.n_o + 7) / 8 * 8;
    q.f1nx = (f1->N + 63) / 64; q.f1ny = ny; q.n_f1 = q.f1nx * ny; q.n_f1_pad = (q.n_f1 + 7) / 8 * 8;
    q.f2nx = 12; q.f2ny = ny; q.f2nz = ks2; q.n_f2 = 12 * ny * ks2;
    q.flags1 = flags; q.flags2 = flags + ny;
    UNIVL_CHECK_ARG(q.n_o_pad + q.n_f1_pad + q.n_f2 <= 512, UNIVL_EUNSUPPORTED, "univl_proto_layer_ffn: %d workgroups do not fit one resident round",
                    q.n_o_pad + q.n_f1_pad + q.n_f2);
    LnFold l1, l2;
    l1.ln = *ln1; l1.counters = ctr1;
    l2.ln = *ln2; l2.counters = ctr2;
    static bool done[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(layer_ffn_proto_kernel, RIDER_SMEM, done);
    hipLaunchKernelGGL(layer_ffn_proto_kernel, dim3(q.n_o_pad + q.n_f1_pad + q.n_f2), dim3(512), RIDER_SMEM, stream, ao, a1, a2, l1, l2, q);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
#endif  // UNIVL_PROTO

static int vocab_ce_prepare(const UnivlVocabCE* d, VocabCeArgs& a, const char* who, bool bwd) {
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "%s: null descriptor", who);
    UNIVL_CHECK_ARG(d->dtype == UNIVL_F32 || d->dtype == UNIVL_BF16, UNIVL_EUNSUPPORTED, "%s: dtype %d", who, d->dtype);
    const int bk = d->dtype == UNIVL_BF16 ? 64 : 32;
    UNIVL_CHECK_ARG(d->rows > 0 && d->V > 0 && d->K > 0 && d->K % bk == 0, UNIVL_EINVAL, "%s: rows %d V %d K %d (K must be a multiple of %d)", who,
                    d->rows, d->V, d->K, bk);
    UNIVL_CHECK_ARG(d->x && d->table && d->labels && d->partial && d->label_logit && d->lse && d->rowloss && d->scratch2 && d->loss, UNIVL_EINVAL,
                    "%s: null buffer", who);
    UNIVL_CHECK_ARG(d->ldx >= d->K && d->ldt >= d->K && d->slots >= (d->V + 127) / 128, UNIVL_EINVAL, "%s: ldx %ld ldt %ld slots %d", who, (long)d->ldx,
                    (long)d->ldt, d->slots);
    const int al = d->dtype == UNIVL_BF16 ? 8 : 4;      // 16-byte rows for the LDS-DMA
    UNIVL_CHECK_ARG(d->ldx % al == 0 && d->ldt % al == 0 && ((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->table & 15) == 0, UNIVL_EINVAL,
                    "%s: operands must be 16-byte aligned with 16-byte row pitches", who);
    if (bwd) UNIVL_CHECK_ARG(d->dlogits && d->lddl >= d->V, UNIVL_EINVAL, "%s: dlogits %p lddl %ld", who, d->dlogits, (long)d->lddl);
    a.X = d->x; a.ldx = d->ldx; a.E = d->table; a.lde = d->ldt; a.bias = d->bias;
    a.rows = d->rows; a.V = d->V; a.K = d->K;
    a.labels = d->labels; a.ignore = d->ignore_index;
    a.partial = d->partial; a.slots = d->slots; a.label_logit = d->label_logit;
    a.lse = d->lse; a.scal = d->scratch2; a.gout = d->gout; a.dl = d->dlogits; a.lddl = d->lddl;
    a.nx = (d->V + 127) / 128; a.ny = (d->rows + 127) / 128;
    return UNIVL_OK;
}

extern "C" int univl_vocab_ce_fwd(const UnivlVocabCE* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    VocabCeArgs a;
    const int rc = vocab_ce_prepare(d, a, "univl_vocab_ce_fwd", false);
    if (rc != UNIVL_OK) return rc;
    const int r2 = d->dtype == UNIVL_BF16 ? vocab_ce_launch<__bf16, 4>(a, false, stream) : vocab_ce_launch<float, 2>(a, false, stream);
    if (r2 != UNIVL_OK) return r2;
    hipLaunchKernelGGL(vocab_ce_rows_kernel, dim3((d->rows + 3) / 4), dim3(256), 0, stream, d->partial, d->slots, a.nx, d->label_logit, d->labels,
                       d->ignore_index, d->rows, d->lse, d->rowloss);
    hipLaunchKernelGGL(vocab_ce_loss_kernel, dim3(1), dim3(256), 0, stream, d->rowloss, d->labels, d->ignore_index, d->rows, d->scratch2, d->loss);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_vocab_ce_bwd(const UnivlVocabCE* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    VocabCeArgs a;
    const int rc = vocab_ce_prepare(d, a, "univl_vocab_ce_bwd", true);
    if (rc != UNIVL_OK) return rc;
    return d->dtype == UNIVL_BF16 ? vocab_ce_launch<__bf16, 4>(a, true, stream) : vocab_ce_launch<float, 2>(a, true, stream);
}

extern "C" int univl_gemm_group(const UnivlGemm* d, int n, hipStream_t stream) {
    return univl_gemm_group_limited(d, n, 0, stream);
}

extern "C" int univl_gemm_group_limited(const UnivlGemm* d, int n, int max_blocks, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(d != nullptr && n >= 1 && n <= UNIVL_GEMM_GROUP_MAX, UNIVL_EINVAL, "univl_gemm_group: n=%d (1..%d)", n,
                    UNIVL_GEMM_GROUP_MAX);
    if (n == 1 && max_blocks <= 0) return univl_gemm(d, stream);
    GroupArgs g;
    // the group runs one kernel instantiation: the smallest tile / fewest waves any member would pick alone
    int tile_all = max_blocks > 0 ? 128 : 256, waves_all = 8;       // the walking form (max_blocks) exists on the older tiles only
    for (int i = 0; i < n; ++i) {
        UNIVL_CHECK_ARG(d[i].dtype == d[0].dtype && d[i].trans_a == d[0].trans_a && d[i].trans_b == d[0].trans_b, UNIVL_EINVAL,
                        "univl_gemm_group: members must share dtype and operand layouts");
        const Choice ci = choose(&d[i], 0, 0, false, max_blocks <= 0);
        tile_all = ci.tile < tile_all ? ci.tile : tile_all;
        waves_all = ci.waves < waves_all ? ci.waves : waves_all;
    }
    // Weight gradients over thousands of tokens on the 64 tile (the members are too small for the 128 tile): 64-deep K steps, four
    // workgroups per compute unit.  Measured at 128 pairs x 48 tokens (profiles/r03i_ab_summary.txt): 13.55 / 13.49 vs 14.01 / 14.04 ms
    // per step.
    int forced_nc = 0;
    if (tile_all == 64 && d[0].dtype == UNIVL_BF16 && d[0].trans_a && d[0].trans_b) {
        forced_nc = 2;
        for (int i = 0; i < n; ++i) if (d[i].K < GEMM_NC64_MIN || (d[i].ksplit > 1)) forced_nc = 0;
        // ... on 4 waves (2 x 2, a 32 x 32 sub-tile each: 4 transpose-read fragments per 4 MFMAs) instead of 8 (32 x 16: 3 per 2):
        // isolated, a layer's group at 6144 tokens runs 231 vs 286 us (profiles/r03w_gemm_group_variants_b128.txt)
        if (forced_nc == 2) {
            bool plain = true;
            for (int i = 0; i < n; ++i) plain = plain && d[i].waves == 0;
            if (plain) waves_all = 4;
        }
    }
    int total = 0, nc_all = 0, cs_total = 0;
    const int bm = tile_all, bn = tile_all;
    // On the 128 tile (weight gradients over thousands of tokens) a member's bias gradient is taken by column-sum workgroups in front
    // of the tiles (colsum_tile) instead of the product's own column-0 workgroups, which were the stragglers of a one-round launch
    // (322 vs 138 us per layer at 6144 tokens, profiles/r03w / r03y2); round 3 used a separate column-sum launch on the chain.
    const bool cs_roles = tile_all >= 128 && d[0].dtype == UNIVL_BF16 && d[0].trans_a;
    for (int i = 0; i < UNIVL_GEMM_GROUP_MAX; ++i) {
        g.cs_first[i] = cs_total;
        g.cs[i] = ColsumArgs{nullptr, 0, 0, 0, nullptr, 0, 0};
        if (i < n && cs_roles && d[i].dbias && d[i].M % 8 == 0) {
            g.cs[i] = ColsumArgs{reinterpret_cast<const __bf16*>(d[i].A), d[i].lda, d[i].K, d[i].M, d[i].dbias, (d[i].M + 63) / 64, 0};
            cs_total += g.cs[i].n_tiles;
        }
    }
    cs_total = (cs_total + 7) / 8 * 8;
    g.cs_first[UNIVL_GEMM_GROUP_MAX] = cs_total;
    for (int i = 0; i < UNIVL_GEMM_GROUP_MAX; ++i) {
        g.first[i] = total;
        if (i >= n) { g.p[i] = g.p[0]; g.nx[i] = g.nxy[i] = g.nz[i] = 1; continue; }
        int ksplit;
        Choice ci;
        UNIVL_CHECK_ARG(!(d[i].A_lo || d[i].B_lo || d[i].C16_lo), UNIVL_EUNSUPPORTED, "univl_gemm_group: member %d carries operand pairs (univl_gemm only)", i);
        const int rc = prepare(&d[i], g.p[i], ksplit, ci, tile_all, forced_nc, false, tile_all == 256);
        if (rc != UNIVL_OK) return rc;
        if (g.cs[i].x) g.p[i].dbias = nullptr;
        UNIVL_CHECK_ARG(ci.tile == tile_all, UNIVL_EINVAL, "univl_gemm_group: member %d cannot run the group's %d tile (sumsq_rows %d)", i, tile_all, d[i].sumsq_rows);
        nc_all = (i == 0) ? ci.nc : (nc_all == ci.nc ? ci.nc : -1);
        g.nx[i] = (d[i].N + bn - 1) / bn;
        g.nxy[i] = g.nx[i] * ((d[i].M + bm - 1) / bm);
        g.nz[i] = ksplit;
        total += g.nxy[i] * ksplit;
    }
    g.first[UNIVL_GEMM_GROUP_MAX] = total;
    const bool ta = d[0].trans_a, tb = d[0].trans_b;
#define UNIVL_GROUP_CASE(T, BMV, BNV, NCV, WM_, WN_)                                                                   \
    do {                                                                                                               \
        if (!ta && !tb) return launch_group<T, false, false, BMV, BNV, NCV, WM_, WN_>(g, max_blocks, stream);          \
        if (!ta && tb) return launch_group<T, false, true, BMV, BNV, NCV, WM_, WN_>(g, max_blocks, stream);            \
        if (ta && tb) return launch_group<T, true, true, BMV, BNV, NCV, WM_, WN_>(g, max_blocks, stream);              \
        return launch_group<T, true, false, BMV, BNV, NCV, WM_, WN_>(g, max_blocks, stream);                           \
    } while (0)
    UNIVL_CHECK_ARG(nc_all > 0, UNIVL_EINVAL, "univl_gemm_group: members disagree on the K-step depth (mixed contraction lengths)");
    if (d[0].dtype == UNIVL_BF16) {
        const bool w8 = waves_all == 8 && nc_all != 6;
        if (tile_all == 256) {
            if (!ta && !tb) return launch_group256<false, false>(g, stream);
            if (!ta && tb) return launch_group256<false, true>(g, stream);
            return launch_group256<true, true>(g, stream);
        }
        if (tile_all == 128) { if (w8) UNIVL_GROUP_CASE(__bf16, 128, 128, 2, 2, 4); UNIVL_GROUP_CASE(__bf16, 128, 128, 2, 2, 2); }
        if (nc_all == 6) return launch_group<__bf16, true, true, 64, 64, 6>(g, max_blocks, stream);
        if (nc_all == 2) { if (w8) UNIVL_GROUP_CASE(__bf16, 64, 64, 2, 2, 4); UNIVL_GROUP_CASE(__bf16, 64, 64, 2, 2, 2); }
        if (w8) UNIVL_GROUP_CASE(__bf16, 64, 64, 4, 2, 4);
        UNIVL_GROUP_CASE(__bf16, 64, 64, 4, 2, 2);
    }
    if (tile_all == 128) UNIVL_GROUP_CASE(float, 128, 128, 2, 2, 2);
    UNIVL_GROUP_CASE(float, 64, 64, 4, 2, 2);
#undef UNIVL_GROUP_CASE
}
