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
//   * One K tile (64 deep) = four PHASES, one quadrant each:  P1 reads B0 + A0 -> (0,0);  P2 reads B1 -> (0,1);  P3 reads A1 -> (1,1);
//     P4 reads nothing -> (1,0).  A phase = { fragment reads, ONE half-tile of DMA issued, s_barrier, 8 MFMAs under s_setprio 1,
//     s_barrier }.  Waves 4-7 run ONE BARRIER BEHIND waves 0-3, so that on every SIMD one wave multiplies while its partner reads and
//     issues DMA (the matrix pipe is per SIMD and in order: two waves multiplying at once gain nothing, one reading beside one
//     multiplying hides the reads).
//   * The DMA stream is NEVER drained inside the loop: half-tiles go out in the order they are consumed (B0 A0 B1 A1 of tile 0, of tile
//     1, ...), phase p of tile t issues stream element 4 t + 6 + p, and ONE counted wait per K tile (s_waitcnt vmcnt(6) in P4: three
//     half-tiles stay in flight) makes tile t + 1 complete.  Ordering rules (guide, "256^2 8-phase template"):
//       RAW  a half-tile is read at least one phase after the wait that retires it, with a barrier every wave has passed in between
//            (the staggered half adds one: P4's wait sits in front of P4's FIRST barrier, the first read of the retired stage is P1 of
//            the next tile -- two barriers later for either half);
//       WAR  a half-tile is re-issued two phases after its last fragment read (A1: read in P3, re-issued in P1 of the next tile; A0:
//            P1 -> P3; B1: P2 -> P4) -- or ONE phase after it where an lgkmcnt in front of the reading phase's first barrier retired
//            the reads (B0: read first in P1, `s_waitcnt lgkmcnt(<reads issued behind them>)`, re-issued in P2).
//   * Epilogue from registers, one 32 x 32 block at a time: a lane holds 16 rows of ONE column, so 32 lanes write 32 consecutive
//     outputs of a row (128 bytes of fp32).  Same epilogue semantics as gemm_tile (alpha, bias, fp32 residual, erf-GELU forward with the
//     saved pre-activation, GELU', accumulate, fp32 and / or bf16 output, split-K atomics, per-wave sum of squares).
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
    // fragment (8 bf16: contraction indices 16 ks + 8 (lane >> 5) + 0..7 of the block row / column lane & 31)
    __device__ static __forceinline__ bf16x8_t frag(const unsigned char* img, const Lane& o, int ks) {
        if (TR) {
            typedef __attribute__((address_space(3))) short4_t lds_s4;
            const unsigned char* p = img + o.off[0] + 4096 * ks;
            const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
            const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 1024));
            union { short s[8]; bf16x8_t f; } u;
            u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
            u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
            return u.f;
        } else {
            return *reinterpret_cast<const bf16x8_t*>(img + o.off[ks]);
        }
    }
    static constexpr int READS = TR ? 2 : 1;      // LDS instructions per fragment
};

// LDS-DMA of one half-tile: this thread's two 16-byte pieces (L = tid, tid + 512); wave-uniform destination + lane * 16
__device__ __forceinline__ void g256_dma(const __bf16* p0, const __bf16* p1, long off, unsigned char* img, int wave) {
    unsigned char* d = img + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p0 + off),
                                     (__attribute__((address_space(3))) void*)(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p1 + off),
                                     (__attribute__((address_space(3))) void*)(d + 8192), 16, 0, 0);
}

// erf-GELU (until_module.py:28-33) and its derivative for this body's epilogue.  libm's erff is two divergent branches of ~35 VALU
// instructions each -- at 128 results per lane and ONE workgroup per compute unit (nothing else to overlap with) that is ~15 us per
// 256 x 256 tile, longer than a 12-tile K loop.  Here: Abramowitz-Stegun 7.1.26, branch-free, |error| <= 1.5e-7 in erf (fp32
// round-off class; the results are rounded to bf16 = 4e-3 next):  with z = |x| / sqrt 2, t = 1 / (1 + p z), E = exp(-z^2) = exp(-x^2 / 2),
// q = poly(t) E:   Phi(x) = 0.5 (1 + erf(x / sqrt 2)) = 1 - q / 2  (x >= 0),  q / 2  (x < 0)   -- no cancellation in the negative tail;
// gelu = x Phi,  gelu' = Phi + x E / sqrt(2 pi)  (the same exponential).
__device__ __forceinline__ void g256_phi(float x, float& phi, float& E) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    E = __expf(-z * z);
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(t, q, 1.421413741f);
    q = fmaf(t, q, -0.284496736f);
    q = fmaf(t, q, 0.254829592f);
    q = q * t * E * 0.5f;
    phi = x >= 0.0f ? 1.0f - q : q;
}
__device__ __forceinline__ float g256_gelu(float x) { float phi, E; g256_phi(x, phi, E); return x * phi; }
__device__ __forceinline__ float g256_gelu_grad(float x) { float phi, E; g256_phi(x, phi, E); return fmaf(x * 0.39894228040143267794f, E, phi); }

#define G256_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G256_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
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
    const typename OA::Lane la0 = OA::lane_offsets(lane, wr * 64), la1 = OA::lane_offsets(lane, wr * 64 + 32);
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

    // the DMA stream: element s = 4 t + e,  e = 0 B0, 1 A0, 2 B1, 3 A1 of K tile t, into stage t & 1
    auto issue = [&](int t, int e, unsigned char* stage) {
        if (e == 0) g256_dma(pb0, pb1, (long)t * ktB, stage + G256_B0, wave);
        else if (e == 1) g256_dma(pa0, pa1, (long)t * ktA, stage + G256_A0, wave);
        else if (e == 2) g256_dma(pb0, pb1, (long)t * ktB + hB, stage + G256_B1, wave);
        else g256_dma(pa0, pa1, (long)t * ktA + hA, stage + G256_A1, wave);
    };

    UNIVL_TRACE_AT(0);
    // prologue: K tile 0 and the first three half-tiles of K tile 1
    issue(0, 0, st0); issue(0, 1, st0); issue(0, 2, st0); issue(0, 3, st0);
    issue(1, 0, st1); issue(1, 1, st1); issue(1, 2, st1);
    G256_WAIT_VM(6);                             // K tile 0 has landed (this wave's share)
    G256_BARRIER();                              // ... every wave's share
    if (wr == 1) G256_BARRIER();                 // waves 4-7 run one barrier behind (wr is wave-uniform: an scc branch)
    UNIVL_TRACE_AT(1);

    bf16x8_t fa[2][4], fb0[4], fb1[4];

    // One K tile.  ST / OT: this tile's stage / the other one.  I1: P1 issues A1 of tile t + 1;  I234: P2 - P4 issue B0 A0 B1 of tile
    // t + 2;  W: 6 = the steady-state wait, 0 = drain (the second to last tile), -1 = none (the last tile).
    auto mma8 = [&](f32x16_t (&c)[2], const bf16x8_t (&b)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][ks], b[ks], c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][ks], b[ks], c[1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define G256_KTILE(ST, OT, t, I1, I234, W)                                                                        \
    do {                                                                                                          \
        /* P1: B0 first (retired before the barrier: its slot is re-issued in P2), then A0 */                     \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fb0[ks] = OB::frag(ST + G256_B0, lb, ks);                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                        \
            fa[0][ks] = OA::frag(ST + G256_A0, la0, ks);                                                          \
            fa[1][ks] = OA::frag(ST + G256_A0, la1, ks);                                                          \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (I1) issue((t) + 1, 3, OT);                                                                            \
        if (TA) G256_WAIT_LGKM(15); else G256_WAIT_LGKM(8);                                                       \
        G256_BARRIER();                                                                                           \
        mma8(acc[0][0], fb0);                                                                                     \
        G256_BARRIER();                                                                                           \
        /* P2 */                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fb1[ks] = OB::frag(ST + G256_B1, lb, ks);                \
        if (I234) issue((t) + 2, 0, ST);                                                                          \
        G256_BARRIER();                                                                                           \
        mma8(acc[0][1], fb1);                                                                                     \
        G256_BARRIER();                                                                                           \
        /* P3 */                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                        \
            fa[0][ks] = OA::frag(ST + G256_A1, la0, ks);                                                          \
            fa[1][ks] = OA::frag(ST + G256_A1, la1, ks);                                                          \
        }                                                                                                         \
        if (I234) issue((t) + 2, 1, ST);                                                                          \
        G256_BARRIER();                                                                                           \
        mma8(acc[1][1], fb1);                                                                                     \
        G256_BARRIER();                                                                                           \
        /* P4 */                                                                                                  \
        if (I234) issue((t) + 2, 2, ST);                                                                          \
        if ((W) == 6) G256_WAIT_VM(6); else if ((W) == 0) G256_WAIT_VM(0);                                        \
        G256_BARRIER();                                                                                           \
        mma8(acc[1][0], fb0);                                                                                     \
        G256_BARRIER();                                                                                           \
    } while (0)

    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        G256_KTILE(st0, st1, t, true, true, 6);
        G256_KTILE(st1, st0, t + 1, true, true, 6);
    }
    G256_KTILE(st0, st1, t, true, false, 0);
    G256_KTILE(st1, st0, t + 1, false, false, -1);
#undef G256_KTILE
    if (wr == 0) G256_BARRIER();                 // waves 0-3 take the barrier waves 4-7 took in front of the loop
    UNIVL_TRACE_AT(2);

    // ------------------------------------------------------------------------------------------ epilogue
    const bool first_slice = (bz == 0);
    const bool atomic = (p.flags & UNIVL_GEMM_ATOMIC) != 0;
    const bool nt_out = (p.flags & UNIVL_GEMM_NT_OUT) != 0;
    __bf16* C16 = reinterpret_cast<__bf16*>(p.C16);
    __bf16* aux = reinterpret_cast<__bf16*>(p.aux);
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    float ssq = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 128 + wc * 32 + lcol;
        const float bv = (p.bias && first_slice) ? p.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const long row0 = m0 + i * 128 + wr * 64 + a * 32 + lrow;      // + (r & 3) + 8 (r >> 2)
                float ev[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ev[r] = acc[i][j][a][r] * p.alpha + bv;
                if (p.R && first_slice) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = p.R[(row0 + (r & 3) + 8 * (r >> 2)) * p.ldr + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ev[r] += rv[r];
                }
                if (p.flags & UNIVL_GEMM_GELU_FWD) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        aux[(row0 + (r & 3) + 8 * (r >> 2)) * p.ldaux + col] = (__bf16)ev[r];
                        ev[r] = g256_gelu(ev[r]);
                    }
                }
                if (p.flags & UNIVL_GEMM_GELU_BWD) {
                    __bf16 uv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) uv[r] = aux[(row0 + (r & 3) + 8 * (r >> 2)) * p.ldaux + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ev[r] *= g256_gelu_grad((float)uv[r]);
                }
                if ((p.flags & UNIVL_GEMM_ACCUM) && !atomic) {
                    float cv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) cv[r] = p.C32[(row0 + (r & 3) + 8 * (r >> 2)) * p.ldc + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ev[r] += cv[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long o = (row0 + (r & 3) + 8 * (r >> 2)) * p.ldc + col;
                    if (atomic) {
                        unsafeAtomicAdd(p.C32 + o, ev[r]);
                    } else {
                        if (p.C32) { if (nt_out) __builtin_nontemporal_store(ev[r], p.C32 + o); else p.C32[o] = ev[r]; }
                        if (C16) C16[o] = (__bf16)ev[r];
                        ssq += ev[r] * ev[r];
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
