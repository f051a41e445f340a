// The MFMA-bound GEMM body (round 5): a 256 x 256 output tile on 8 waves, for the products of thousands of tokens
// (32 / 64 / 128 pairs per GPU: 1536 / 3072 / 6144 rows), where gemm_tile's 64 / 128 tiles sit at 0.16 - 0.20 of the dense bf16 peak
// with their waves parked on the single in-flight tile (profiles/r04_final2_pmc_stall_b128.txt).  Included by gemm.hip (same
// translation unit: GemmArgs, smem_raw and the tile maps live there).
//
// Structure (the guide's 256^2 "8-phase" schedule, written for this library's three operand layouts):
//   * v_mfma_f32_32x32x16_bf16; a wave owns 128 x 64 outputs as 2 x 2 QUADRANTS of 64 x 32 (two 32 x 32 accumulator blocks each); the
//     wave's rows / columns are interleaved over the tile's halves (rows  i * 128 + wr * 64 + [0, 64),  columns  j * 128 + wc * 32 +
//     [0, 32);  wr = wave >> 2, wc = wave & 3) so that quadrant (i, j) of EVERY wave reads operand half-tiles A_i and B_j.
//   * LDS: two stages x four half-tiles (A0 A1 B0 B1) of 128 rows x 64 contraction indices = 16 KB each, 128 KB in all, filled by
//     LDS-DMA (global_load_lds_dwordx4: every wave two 1-KB pieces per half-tile).  The images are lane-linear for the DMA; bank
//     conflicts are removed by an XOR of the 16-byte piece index applied to the per-lane SOURCE address and to every fragment read:
//        K-major half  [128 rows][64 k]   (128-byte rows): piece (r, q) at  r * 128 + ((q ^ ((r >> 1) & 7)) << 4)   -- ds_read_b128
//        T-major half  [64 k][128 rows]   (256-byte rows): piece (k, q) at  k * 256 + ((q ^ ((k & 3) << 2)) << 4)   -- ds_read_b64_tr_b16
//     (ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27} ...: their 16 rows hit 16 distinct 16-byte slots of the 256-byte
//     bank row; a transpose read's 32-lane half covers 4 k-rows x 64 bytes = all 64 banks once.)
//   * One K tile (64 deep) = four PHASES, one quadrant each:  P1 reads A0 -> (0,0);  P2 reads B1 -> (0,1);  P3 reads A1 -> (1,1);
//     P4 reads the NEXT tile's B0 -> (1,0) with this tile's B0 (two register sets, alternating with the tile parity): 8 / 4 / 8 / 4
//     fragment reads per phase.  (First version: P1 read B0 + A0 = 12 and P4 nothing -- the K loop ran 1.2 - 1.6 us per K tile against
//     0.9 of MFMA time, the 12-read phase longer than the partner's 8 MFMAs; profiles/r05b_trace_gemm256.txt.)  A phase = { fragment
//     reads, ONE half-tile of DMA issued, counted wait, s_barrier, 8 MFMAs under s_setprio 1, s_barrier }.  Waves 4-7 run ONE BARRIER
//     BEHIND waves 0-3, so that on every SIMD one wave multiplies while its partner reads and issues DMA (the matrix pipe is per SIMD
//     and in order: two waves multiplying at once gain nothing, one reading beside one multiplying hides the reads).
//   * The DMA stream is NEVER drained inside the loop: half-tiles go out in the order they are consumed (B0 A0 B1 A1 of tile 0, of tile
//     1, ...), phase p of tile t issues stream element 4 t + 6 + p and then waits `s_waitcnt vmcnt(10)`: FIVE half-tiles (80 KB) stay in
//     flight, the element issued six phases earlier (4 t + 1 + p) has landed.  Ordering rules (guide, "256^2 8-phase template"):
//       RAW  a half-tile is read in the phase AFTER the wait that retires it (P3's wait retires the next tile's B0, read in P4; P4's
//            retires its A0 -> P1; P1's this tile's B1 -> P2; P2's A1 -> P3): the wait sits in front of the phase's first barrier, the
//            read behind its second one, which the staggered half has reached with ITS wait done;
//       WAR  a half-tile is re-issued two phases after its last fragment read (B0 of tile t: read in P4 of t - 1, element B0 of t + 2
//            issued in P2 of t; A0: P1 -> P3; B1: P2 -> P4; A1: P3 -> P1 of the next tile): the staggered half's reads of phase p are
//            retired (its MFMAs consumed them) before it passes the barrier that lets the leading half enter phase p + 2.
//   * Epilogue from registers, one 32 x 32 block at a time.  The MFMAs take the B fragment as their first operand, so an accumulator
//     block is the transposed output block: a lane holds one output row and 4 x 4 consecutive columns of it -- 16-byte fp32 / 8-byte
//     bf16 loads and stores.  Same epilogue semantics as gemm_tile (alpha, bias, fp32 residual, erf-GELU forward with the saved
//     pre-activation, GELU', accumulate, fp32 and / or bf16 output, split-K atomics, per-wave sum of squares).
//
// Requirements (the host's `choose` checks them; everything else keeps gemm_tile): bf16, M and N multiples of 256, every K slice a
// multiple of 128 (two K tiles per loop trip: the stage index is a compile-time constant), no in-tile bias gradient (the grouped
// launch takes bias gradients as column-sum roles).
#pragma once

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int G256_HALF = 16384;                 // one operand half-tile image
constexpr int G256_STAGE = 4 * G256_HALF;        // A0 A1 B0 B1
constexpr int G256_SMEM = 2 * G256_STAGE;        // 128 KB: one workgroup per compute unit
constexpr int G256_A0 = 0, G256_A1 = G256_HALF, G256_B0 = 2 * G256_HALF, G256_B1 = 3 * G256_HALF;

template <bool TR> struct Op256 {
    // element offset, relative to (first row of the half-tile, contraction index 0 of the K tile), of the 16-byte piece that the DMA
    // drops at lane-linear LDS piece L of the half-tile image
    __host__ __device__ static __forceinline__ long src(int L, long ld) {
        if (TR) {
            const int k = L >> 4, q = (L & 15) ^ ((k & 3) << 2);
            return (long)k * ld + q * 8;
        } else {
            const int r = L >> 3, q = (L & 7) ^ ((r >> 1) & 7);
            return (long)r * ld + q * 8;
        }
    }
    __device__ static __forceinline__ long ktile(long ld) { return TR ? 64 * ld : 64; }      // next K tile
    __device__ static __forceinline__ long half(long ld) { return TR ? 128 : 128 * ld; }     // second half-tile of the operand

    // Per-lane byte offsets inside a half-tile image for the fragments of a 32-row block starting at row `r32` (a multiple of 32).
    // K-major: off[ks] for the four 16-deep k steps (one ds_read_b128 each).  T-major: off[0] only; k step ks / second read add the
    // immediates 4096 * ks and 1024.
    struct Lane { int off[4]; };
    __host__ __device__ static __forceinline__ Lane lane_offsets(int lane, int r32) {
        Lane o;
        if (TR) {
            const int i = lane & 15, u = (lane >> 4) & 1, h = lane >> 5;
            const int krow = 8 * h + (i >> 2);                       // + 16 ks + 4 rd
            const int slot = ((r32 >> 3) + 2 * u + ((i & 3) >> 1)) ^ (((i >> 2) & 3) << 2);
            o.off[0] = krow * 256 + (slot << 4) + (i & 1) * 8;
            o.off[1] = o.off[2] = o.off[3] = 0;
        } else {
            const int row = lane & 31, h = lane >> 5, f = (row >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) o.off[ks] = (r32 + row) * 128 + (((2 * ks + h) ^ f) << 4);
        }
        return o;
    }
    // fragment (8 bf16: contraction indices 16 KS + 8 (lane >> 5) + 0..7 of the block row / column lane & 31) as 4 dwords.
    // OFF: compile-time byte offset of the half-tile image (K-major: + block offset) from `stage`;  tr: T-major only, the 32-bit LDS
    // address of `stage` + o.off[0].
    // T-major reads are INLINE ASM: for the builtin (__builtin_amdgcn_ds_read_tr16_b64) hipcc cannot tell that the read does not alias
    // the LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of every batch -- the DMA stream drained four times per K tile (the
    // first version of this body: 1.5 us per K tile with T-major operands against 1.2 with K-major ones; profiles/r05b_*).  hipcc does
    // not count an asm load: G256_FRAGS_LANDED below is the wait (guide 5.7 item 1, form ii: the wait statement names every
    // destination "+v", so nothing reads, copies or reuses one before it).
    template <int OFF, int KS>
    __device__ static __forceinline__ void frag(u32x2_t& lo, u32x2_t& hi, const unsigned char* stage, unsigned tr, const Lane& o) {
        if constexpr (TR) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(tr), "i"(OFF + 4096 * KS));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(tr), "i"(OFF + 4096 * KS + 1024));
        } else {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stage + OFF + o.off[KS]);
            lo = u32x2_t{v[0], v[1]};
            hi = u32x2_t{v[2], v[3]};
        }
    }
};

struct Frag256 { u32x2_t lo, hi; };              // one operand fragment: 8 bf16 in 4 VGPRs
__device__ __forceinline__ bf16x8_t g256_bf16(const Frag256& f) {
    return __builtin_bit_cast(bf16x8_t, u32x4_t{f.lo[0], f.lo[1], f.hi[0], f.hi[1]});
}
#define G256_FRAG4(OP, dst, OFF, stage, tr, lo_)                              \
    do {                                                                       \
        OP::template frag<OFF, 0>(dst[0].lo, dst[0].hi, stage, tr, lo_);       \
        OP::template frag<OFF, 1>(dst[1].lo, dst[1].hi, stage, tr, lo_);       \
        OP::template frag<OFF, 2>(dst[2].lo, dst[2].hi, stage, tr, lo_);       \
        OP::template frag<OFF, 3>(dst[3].lo, dst[3].hi, stage, tr, lo_);       \
    } while (0)
// every fragment read of the phase has landed (compiler-issued K-major reads included: lgkmcnt(0)); the four fragments of `f` are
// (re)defined here as far as the compiler is concerned; pin the MFMAs below it (guide rule 18)
#define G256_FRAGS_LANDED4(f)                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo),   \
                 "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi) :: "memory")

// LDS-DMA of one half-tile: this thread's two 16-byte pieces (L = tid, tid + 512); wave-uniform destination + lane * 16
__device__ __forceinline__ void g256_dma(const __bf16* p0, const __bf16* p1, long off, unsigned char* img, int wave) {
    unsigned char* d = img + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p0 + off),
                                     (__attribute__((address_space(3))) void*)(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p1 + off),
                                     (__attribute__((address_space(3))) void*)(d + 8192), 16, 0, 0);
}

// (the branch-free erf-GELU of this body's epilogue, g256_gelu / g256_gelu_grad, lives in common.h: gemm_tile's bf16 epilogues use it too)

#define G256_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G256_WAIT_VM_N(n)                                                                        \
    do {                                                                                         \
        if ((n) == 10) G256_WAIT_VM(10); else if ((n) == 8) G256_WAIT_VM(8); else if ((n) == 6) G256_WAIT_VM(6);       \
        else if ((n) == 4) G256_WAIT_VM(4); else if ((n) == 2) G256_WAIT_VM(2); else if ((n) == 0) G256_WAIT_VM(0);    \
    } while (0)
#define G256_BARRIER()                              \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_s_barrier();               \
        asm volatile("" ::: "memory");              \
    } while (0)

template <bool TA, bool TB>
__device__ __forceinline__ void gemm256_tile(const GemmArgs& p, const int bx, const int by, const int bz) {
    using OA = Op256<TA>;
    using OB = Op256<TB>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int m0 = by * 256, n0 = bx * 256;
    const int kbeg = bz * p.ksplit_len;
    const int ntiles = (min(p.K, kbeg + p.ksplit_len) - kbeg) >> 6;          // even, >= 2 (host)

    unsigned char* const st0 = smem_raw;
    unsigned char* const st1 = smem_raw + G256_STAGE;

    // this thread's DMA sources: pieces L = tid and tid + 512 of the operand's FIRST half-tile at K tile 0
    const __bf16* Ab = reinterpret_cast<const __bf16*>(p.A) + (TA ? (long)kbeg * p.lda + m0 : (long)m0 * p.lda + kbeg);
    const __bf16* Bb = reinterpret_cast<const __bf16*>(p.B) + (TB ? (long)kbeg * p.ldb + n0 : (long)n0 * p.ldb + kbeg);
    const __bf16* const pa0 = Ab + OA::src(tid, p.lda);
    const __bf16* const pa1 = Ab + OA::src(tid + 512, p.lda);
    const __bf16* const pb0 = Bb + OB::src(tid, p.ldb);
    const __bf16* const pb1 = Bb + OB::src(tid + 512, p.ldb);
    const long ktA = OA::ktile(p.lda), ktB = OB::ktile(p.ldb), hA = OA::half(p.lda), hB = OB::half(p.ldb);

    // fragment read offsets: the wave's two 32-row blocks of an A half, its one 32-column block of a B half
    // (K-major: the second block is the first one + 32 rows x 128 bytes, an immediate offset of the same address registers)
    const typename OA::Lane la0 = OA::lane_offsets(lane, wr * 64);
    const typename OA::Lane la1 = TA ? OA::lane_offsets(lane, wr * 64 + 32) : la0;
    constexpr int A1OFF = TA ? 0 : 32 * 128;
    const typename OB::Lane lb = OB::lane_offsets(lane, wc * 32);

    f32x16_t acc[2][2][2];                       // [row half i][column half j][32-row block a]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][a][r] = 0.0f;

    // K ROTATION: workgroup (bx, by) walks its K tiles cyclically from tile (bx + by) % ntiles.  The sum order of a fp32 accumulator
    // changes (deterministically, per tile); what it buys: the workgroups of one XCD that share an operand panel (same by: the A
    // panel, same bx: the B panel) no longer ask the L2 for the same 32-KB slab in the same microsecond -- the first one pulls it in,
    // the others find it there a K tile or more later (G256_KROT=0: every workgroup starts at tile 0, the A/B build).
#ifndef G256_KROT
#define G256_KROT 1
#endif
    const int krot = G256_KROT ? (bx + by) % ntiles : 0;
    // the DMA stream: element s = 4 t + e,  e = 0 B0, 1 A0, 2 B1, 3 A1 of the t-th K tile of this workgroup's walk, into stage t & 1
    auto issue = [&](int t, int e, unsigned char* stage) {
        t += krot;
        t -= (t >= ntiles) ? ntiles : 0;
        if (e == 0) g256_dma(pb0, pb1, (long)t * ktB, stage + G256_B0, wave);
        else if (e == 1) g256_dma(pa0, pa1, (long)t * ktA, stage + G256_A0, wave);
        else if (e == 2) g256_dma(pb0, pb1, (long)t * ktB + hB, stage + G256_B1, wave);
        else g256_dma(pa0, pa1, (long)t * ktA + hA, stage + G256_A1, wave);
    };

    UNIVL_TRACE_AT(0);
    // prologue: K tile 0 and the first three half-tiles of K tile 1 (stream elements 0 .. 6)
    issue(0, 0, st0); issue(0, 1, st0); issue(0, 2, st0); issue(0, 3, st0);
    issue(1, 0, st1); issue(1, 1, st1); issue(1, 2, st1);
    G256_WAIT_VM(10);                            // B0 and A0 of K tile 0 have landed (this wave's share)
    G256_BARRIER();                              // ... every wave's share
    if (wr == 1) G256_BARRIER();                 // waves 4-7 run one barrier behind (wr is wave-uniform: an scc branch)
    UNIVL_TRACE_AT(1);

    Frag256 fa0[4], fa1[4], fb0e[4], fb0o[4], fb1[4];  // A: the wave's two 32-row blocks;  B0 of even / odd K tiles: the next tile's are read while this tile's are in use
    // 32-bit LDS addresses for the asm transpose reads (stage 1 = stage 0 + 64 KB: the 16-bit immediate cannot hold it)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)st0;
    const unsigned tra0_0 = lds0 + la0.off[0], tra1_0 = lds0 + la1.off[0], trb_0 = lds0 + lb.off[0];
    const unsigned tra0_1 = tra0_0 + G256_STAGE, tra1_1 = tra1_0 + G256_STAGE, trb_1 = trb_0 + G256_STAGE;
    G256_FRAG4(OB, fb0e, G256_B0, st0, trb_0, lb);
    // (this one read sits a phase later than its steady-state place, P4 of the previous tile: retire it HERE, in front of the next
    // barrier -- the slot is re-issued by the leading half in P2 of tile 0, which starts at the staggered half's next barrier but one)
    if constexpr (TB) { G256_FRAGS_LANDED4(fb0e); } else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_sched_barrier(0);

    auto mma8 = [&](f32x16_t (&c)[2], const Frag256 (&b)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g256_bf16(b[ks]), g256_bf16(fa0[ks]), c[0], 0, 0, 0);      // D^T block: see the epilogue
            c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g256_bf16(b[ks]), g256_bf16(fa1[ks]), c[1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // One K tile.  ST / OT: this tile's stage / the other one (S = 0 / 1: its index, for the asm reads' addresses);  FB / FBN: the B0
    // fragments of this tile / of the next one (read in P4);  I1: P1 issues A1 of tile t + 1;  I234: P2 - P4 issue B0 A0 B1 of tile
    // t + 2;  W1 .. W4: the counted wait of each phase (-1: none) = 2 x (half-tiles allowed to stay in flight): 10 in the steady
    // state, shrinking over the last two tiles as the stream ends;  NXT: P4 reads the next tile's B0.
#define G256_KTILE(S, ST, OT, FB, FBN, t, I1, I234, W1, W2, W3, W4, NXT)                                          \
    do {                                                                                                          \
        /* P1: A0 -> quadrant (0, 0) */                                                                           \
        G256_FRAG4(OA, fa0, G256_A0, ST, (S ? tra0_1 : tra0_0), la0);                                             \
        G256_FRAG4(OA, fa1, G256_A0 + A1OFF, ST, (S ? tra1_1 : tra1_0), la1);                                     \
        if (I1) issue((t) + 1, 3, OT);                                                                            \
        G256_WAIT_VM_N(W1);                                                                                       \
        G256_BARRIER();                                                                                           \
        if constexpr (TA) { G256_FRAGS_LANDED4(fa0); G256_FRAGS_LANDED4(fa1); }                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        mma8(acc[0][0], FB);                                                                                      \
        G256_BARRIER();                                                                                           \
        /* P2: B1 -> (0, 1) */                                                                                    \
        G256_FRAG4(OB, fb1, G256_B1, ST, (S ? trb_1 : trb_0), lb);                                                \
        if (I234) issue((t) + 2, 0, ST);                                                                          \
        G256_WAIT_VM_N(W2);                                                                                       \
        G256_BARRIER();                                                                                           \
        if constexpr (TB) { G256_FRAGS_LANDED4(fb1); }                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        mma8(acc[0][1], fb1);                                                                                     \
        G256_BARRIER();                                                                                           \
        /* P3: A1 -> (1, 1) */                                                                                    \
        G256_FRAG4(OA, fa0, G256_A1, ST, (S ? tra0_1 : tra0_0), la0);                                             \
        G256_FRAG4(OA, fa1, G256_A1 + A1OFF, ST, (S ? tra1_1 : tra1_0), la1);                                     \
        if (I234) issue((t) + 2, 1, ST);                                                                          \
        G256_WAIT_VM_N(W3);                                                                                       \
        G256_BARRIER();                                                                                           \
        if constexpr (TA) { G256_FRAGS_LANDED4(fa0); G256_FRAGS_LANDED4(fa1); }                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        mma8(acc[1][1], fb1);                                                                                     \
        G256_BARRIER();                                                                                           \
        /* P4: the NEXT tile's B0 (landed by P3's wait) -> registers; (1, 0) with this tile's B0 */               \
        if (NXT) G256_FRAG4(OB, FBN, G256_B0, OT, (S ? trb_0 : trb_1), lb);                                       \
        if (I234) issue((t) + 2, 2, ST);                                                                          \
        G256_WAIT_VM_N(W4);                                                                                       \
        G256_BARRIER();                                                                                           \
        if constexpr (TB) { if (NXT) G256_FRAGS_LANDED4(FBN); }                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        mma8(acc[1][0], FB);                                                                                      \
        G256_BARRIER();                                                                                           \
    } while (0)

    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        G256_KTILE(0, st0, st1, fb0e, fb0o, t, true, true, 10, 10, 10, 10, true);
        G256_KTILE(1, st1, st0, fb0o, fb0e, t + 1, true, true, 10, 10, 10, 10, true);
    }
    // the last two tiles: stream elements stop at A1 of the last tile, every wait leaves what is still allowed in flight
    G256_KTILE(0, st0, st1, fb0e, fb0o, t, true, false, 10, 8, 6, 4, true);
    G256_KTILE(1, st1, st0, fb0o, fb0e, t + 1, false, false, 2, 0, -1, -1, false);
#undef G256_KTILE
    if (wr == 0) G256_BARRIER();                 // waves 0-3 take the barrier waves 4-7 took in front of the loop
    UNIVL_TRACE_AT(2);

    // ------------------------------------------------------------------------------------------ epilogue
    // The MFMAs ran as (B fragment) x (A fragment): an accumulator block is the TRANSPOSE of the 32 x 32 output block, i.e. a lane
    // holds ONE output row (m = lane & 31 of block a) and, per group g of four registers, FOUR CONSECUTIVE output columns
    // n = 8 g + 4 (lane >> 5) + 0..3.  Every epilogue access is 16 bytes of fp32 / 8 bytes of bf16 per lane: 32 store instructions per
    // wave and output instead of 128 -- with ONE workgroup per compute unit nothing overlaps the epilogue, and the first version
    // (a column per lane, 4-byte stores) spent 16 us of a 33 us launch behind the K loop (profiles/r05a_mb_gemm256.txt).
    const bool first_slice = (bz == 0);
    const bool atomic = (p.flags & UNIVL_GEMM_ATOMIC) != 0;
    const bool nt_out = (p.flags & UNIVL_GEMM_NT_OUT) != 0;
    __bf16* C16 = reinterpret_cast<__bf16*>(p.C16);
    __bf16* aux = reinterpret_cast<__bf16*>(p.aux);
    const int lrow = lane & 31, lcol = 4 * (lane >> 5);
    float ssq = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col0 = n0 + j * 128 + wc * 32 + lcol;                // + 8 g
        f32x4_t bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias && first_slice) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4_t*>(p.bias + col0 + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const long row = m0 + i * 128 + wr * 64 + a * 32 + lrow;
                f32x4_t ev[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ev[g][e] = acc[i][j][a][4 * g + e] * p.alpha + bv[g][e];
                if (p.R && first_slice) {
                    f32x4_t rv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[g] = *reinterpret_cast<const f32x4_t*>(p.R + row * p.ldr + col0 + 8 * g);
#pragma unroll
                    for (int g = 0; g < 4; ++g) ev[g] += rv[g];
                }
                if (p.flags & UNIVL_GEMM_GELU_FWD) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4_t u;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { u[e] = (__bf16)ev[g][e]; ev[g][e] = g256_gelu(ev[g][e]); }
                        *reinterpret_cast<bf16x4_t*>(aux + row * p.ldaux + col0 + 8 * g) = u;
                    }
                }
                if (p.flags & UNIVL_GEMM_GELU_BWD) {
                    bf16x4_t uv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) uv[g] = *reinterpret_cast<const bf16x4_t*>(aux + row * p.ldaux + col0 + 8 * g);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) ev[g][e] *= g256_gelu_grad((float)uv[g][e]);
                }
                if ((p.flags & UNIVL_GEMM_ACCUM) && !atomic) {
                    f32x4_t cv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) cv[g] = *reinterpret_cast<const f32x4_t*>(p.C32 + row * p.ldc + col0 + 8 * g);
#pragma unroll
                    for (int g = 0; g < 4; ++g) ev[g] += cv[g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long o = row * p.ldc + col0 + 8 * g;
                    if (atomic) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(p.C32 + o + e, ev[g][e]);
                    } else {
                        if (p.C32) {
                            if (nt_out) __builtin_nontemporal_store(ev[g], reinterpret_cast<f32x4_t*>(p.C32 + o));
                            else *reinterpret_cast<f32x4_t*>(p.C32 + o) = ev[g];
                        }
                        if (C16) {
                            bf16x4_t c;
#pragma unroll
                            for (int e = 0; e < 4; ++e) c[e] = (__bf16)ev[g][e];
                            *reinterpret_cast<bf16x4_t*>(C16 + o) = c;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) ssq += ev[g][e] * ev[g][e];
                    }
                }
            }
    }
    if (p.sumsq) {                               // per-wave partial sums, plain stores: 8 slots per tile (univl_hip.h)
        ssq = wave_sum(ssq);
        const int tensor = p.sumsq_rows > 0 ? m0 / p.sumsq_rows : 0;
        const int mloc = p.sumsq_rows > 0 ? m0 % p.sumsq_rows : m0;
        const int nx = p.N >> 8;
        if (lane == 0) p.sumsq[(long)tensor * p.sumsq_stride + ((mloc >> 8) * nx + bx) * 8 + wave] = ssq;
    }
    UNIVL_TRACE_AT(3);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    xcd_tile(bx, by, bz, p.gm);
    gemm256_tile<TA, TB>(p, bx, by, bz);
}

template <bool TA, bool TB>
int launch256(const GemmArgs& a, int ksplit, hipStream_t stream) {
    static bool attr_done[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(gemm256_kernel<TA, TB>, G256_SMEM, attr_done);
    dim3 grid(a.N / 256, a.M / 256, ksplit);
    hipLaunchKernelGGL((gemm256_kernel<TA, TB>), grid, dim3(512), G256_SMEM, stream, a);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
