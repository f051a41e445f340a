// Shared pieces of the attention kernels (attention.hip) and of the fused backward launch in gemm.hip (round 5): LDS image pitches, the
// two-matrix staging, mask / seed helpers, argument validation, and the BODY of the backward kernel as a device function -- so that a
// workgroup which has just computed the upstream gradient dO of its (batch row, head) as a product tile can run the attention backward
// on it without dO ever passing through global memory.  Included inside an anonymous namespace by both translation units.
#pragma once


constexpr int HD = 64;   // head dim

template <typename T> struct AttnCfg {
    static constexpr int CH = Mma<T>::CH;
    static constexpr int TPC = CH / 16;              // 16-wide accumulator tiles per contraction chunk
    static constexpr int NCD = HD / CH;              // chunks across the head dim
    // LDS row pitches (elements).  A staged matrix is read in one or both of two ways, and the two want different pitches (round 3's
    // stall pass, profiles/r03z_pmc_stall_b128.txt: 52-59 % of the kernels' LDS cycles were bank conflicts with ONE 160-byte pitch):
    //   PT  image consumed by the transpose read (ds_read_b64_tr_b16: eight 32-byte row segments per half wave): 160 B rows;
    //   PK  image consumed K-major (two 8-byte reads per lane over 16 consecutive rows): 144 B rows -- 36 dwords, so the 16 rows of a
    //       fragment start 4 banks apart (with 160 B rows i and i + 8 start on the same bank).
    // A matrix that is read both ways is staged as TWO images where the launch can afford the LDS (DUAL, bwd kernel), else as one PT image.
    static constexpr int PT = HD + (sizeof(T) == 2 ? 16 : 4);
    static constexpr int PK = HD + (sizeof(T) == 2 ? 8 : 4);
    static constexpr int EPC = Mma<T>::EPC;
};

// Stage `rows` x 64 of TWO [.., ld] matrices (head slices: K and V, or Q and dO) into LDS, zero-filling up to rows_pad.
// All global loads of a trip (2 pieces x 2 matrices per thread, clamped rows) are issued before the first LDS store: at
// S = 48 that is the whole tile in ONE memory round trip -- a plain "load, store, next piece" loop is not unrolled by
// hipcc (rows_pad is a run-time value) and pays one round trip per piece and matrix.
// TRIPS > 0: compile-time trip count (rows_pad <= 64 * TRIPS for bf16), no run-time loop at all -- a loop header makes
// hipcc drain every load that is already in flight (the per-wave Q / mask / seed requests issued before the staging).
template <typename T, int TRIPS, int PA, int PA2, int PB, int PB2, bool SKIPB = false>
__device__ __forceinline__ void stage_pair(T* ldsA, T* ldsA2, const T* gA, long ldA, T* ldsB, T* ldsB2, const T* gB, long ldB, int rows,
                                           int rows_pad, int tid) {
    // PA / PB: pitch of the (first) image of matrix A / B; PA2 / PB2 > 0: a second image of the same rows with that pitch
    constexpr int EPC = AttnCfg<T>::EPC;
    constexpr int CPR = HD / EPC;   // 16-byte pieces per row
    constexpr int UNR = 2;
    const int total = rows_pad * CPR;
    const int c_end = TRIPS > 0 ? tid + TRIPS * 256 * UNR : total;
#pragma unroll
    for (int c0 = tid; c0 < c_end; c0 += 256 * UNR) {
        uint4 va[UNR], vb[UNR];
        int rr[UNR], ee[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int c = c0 + 256 * u;
            ok[u] = c < total;
            const int cc = min(c, total - 1);
            rr[u] = cc / CPR;
            ee[u] = (cc % CPR) * EPC;
            const long row = min(rr[u], rows - 1);
            va[u] = *reinterpret_cast<const uint4*>(gA + row * ldA + ee[u]);
            if constexpr (!SKIPB) vb[u] = *reinterpret_cast<const uint4*>(gB + row * ldB + ee[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (rr[u] >= rows) { va[u] = make_uint4(0u, 0u, 0u, 0u); if constexpr (!SKIPB) vb[u] = make_uint4(0u, 0u, 0u, 0u); }
            if (ok[u]) {
                *reinterpret_cast<uint4*>(ldsA + rr[u] * PA + ee[u]) = va[u];
                if constexpr (!SKIPB) *reinterpret_cast<uint4*>(ldsB + rr[u] * PB + ee[u]) = vb[u];
                if constexpr (PA2 > 0) *reinterpret_cast<uint4*>(ldsA2 + rr[u] * PA2 + ee[u]) = va[u];
                if constexpr (PB2 > 0) *reinterpret_cast<uint4*>(ldsB2 + rr[u] * PB2 + ee[u]) = vb[u];
            }
        }
    }
}

// additive mask per key for a fixed query: 0, -10000 or -inf (key beyond Sk)
__device__ __forceinline__ float key_bias(const float* sM, int key, int q, int Sk, int causal) {
    if (key >= Sk) return -INFINITY;
    float m = sM[key];
    if (causal && key > q) m = -10000.0f;
    return m;
}

// Additive key mask into LDS in two halves so that its global load travels together with the tile staging loads:
// mask_fetch issues the loads (clamped index; a null mask reads a dummy word through a SELECTED pointer -- a branch
// around a load makes hipcc wait for every outstanding load at the join), mask_store writes sM after the staging.
struct MaskRegs { long v[2]; };
__device__ __forceinline__ MaskRegs mask_fetch(const int64_t* key_mask, const void* dummy, int b, int Sk, int tid) {
    const int64_t* km = key_mask ? key_mask + (long)b * Sk : reinterpret_cast<const int64_t*>(dummy);
    MaskRegs m;
#pragma unroll
    for (int u = 0; u < 2; ++u) m.v[u] = km[key_mask ? min(tid + 256 * u, Sk - 1) : 0];
    return m;
}
__device__ __forceinline__ void mask_store(float* sM, const MaskRegs& m, bool has_mask, int Sk, int Sk_pad, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = tid + 256 * u;
        if (k < Sk_pad) sM[k] = (has_mask && k < Sk && m.v[u] == 0) ? -10000.0f : 0.0f;
    }
}
__device__ __forceinline__ uint64_t seed_fetch(const UnivlAttention& p) {
    const uint64_t* sp = p.seed_dev ? p.seed_dev : reinterpret_cast<const uint64_t*>(p.q);
    const uint64_t dv = *sp;
    return p.seed + (p.seed_dev ? dv : 0ull);
}

template <typename T>
__device__ __forceinline__ void store4(T* p, const f32x4_t& v, float s) {
    if (sizeof(T) == 2) {
        bf16x4_t w;
        w[0] = (__bf16)(v[0] * s); w[1] = (__bf16)(v[1] * s); w[2] = (__bf16)(v[2] * s); w[3] = (__bf16)(v[3] * s);
        *reinterpret_cast<bf16x4_t*>(p) = w;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    }
}


// ------------------------------------------------------------------------------------------------ forward
// FUSED (round 5, gemm.hip: attn_fwd_qkv_kernel): the calling workgroup has just computed q, k, v of this (batch row, head) as product
// tiles and written them (bf16, rows beyond the sequence zero) into the K image, the V image and a Q image ([64][PK] behind the mask):
// nothing is staged from global memory.  One block of at most 64 queries; waves beyond the fourth leave at the first barrier.
template <typename T> struct AttnFwdImages {
    T* sK; T* sV; float* sM; T* sQ;
    __device__ AttnFwdImages(unsigned char* smem, int Sk_pad) {
        sK = reinterpret_cast<T*>(smem);                          // K-major reads only: PK image
        sV = sK + Sk_pad * AttnCfg<T>::PK;                        // transpose reads only: PT image
        sM = reinterpret_cast<float*>(sV + Sk_pad * AttnCfg<T>::PT);
        sQ = reinterpret_cast<T*>(sM + Sk_pad);                   // FUSED only
    }
};
template <typename T, int MAXKT, bool FUSED>
__device__ __forceinline__ void attn_fwd_body(const UnivlAttention& p, const int Sk_pad, const float scale, const int bh, const int yblk,
                                              unsigned char* smem_raw) {
    using C = AttnCfg<T>;
    using M = Mma<T>;
    const AttnFwdImages<T> im(smem_raw, Sk_pad);
    T* sK = im.sK;
    T* sV = im.sV;
    float* sM = im.sM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int b = bh / p.H, h = bh % p.H;
    const T* Kg = reinterpret_cast<const T*>(p.k) + (long)b * (p.bsk ? p.bsk : (long)p.Sk * p.ldk) + h * HD;
    const T* Vg = reinterpret_cast<const T*>(p.v) + (long)b * (p.bsv ? p.bsv : (long)p.Sk * p.ldv) + h * HD;
    // per-wave operands straight from global memory are requested BEFORE the K/V staging round trip (clamped rows:
    // lanes / waves beyond Sq read a valid row and never store), so the kernel pays one global latency, not two
    const int q0 = yblk * 64 + wave * 16;
    const int q = q0 + i;
    const bool qv = q < p.Sq;
    const int qc = min(q, p.Sq - 1);
    const T* Qg = reinterpret_cast<const T*>(p.q) + ((long)b * p.Sq + qc) * p.ldq + h * HD;
    typename M::frag fq[C::NCD];
    if constexpr (!FUSED) {
#pragma unroll
        for (int c = 0; c < C::NCD; ++c) fq[c] = M::gmem_kmajor(Qg + c * C::CH, g);
    }
    const uint64_t seed = seed_fetch(p);
    const MaskRegs mk = mask_fetch(p.key_mask, p.q, b, p.Sk, tid);
    constexpr int TRIPS = (MAXKT * 16 * (HD / C::EPC) <= 1024) ? (MAXKT * 16 * (HD / C::EPC) + 511) / 512 : 0;
    if constexpr (!FUSED) stage_pair<T, TRIPS, C::PK, 0, C::PT, 0>(sK, nullptr, Kg, p.ldk, sV, nullptr, Vg, p.ldv, p.Sk, Sk_pad, tid);
    mask_store(sM, mk, p.key_mask != nullptr, p.Sk, Sk_pad, tid);
    __syncthreads();
    if (q0 >= p.Sq || wave >= 4) return;
    if constexpr (FUSED) {
#pragma unroll
        for (int c = 0; c < C::NCD; ++c) fq[c] = M::lds_kmajor(im.sQ + (q0 + i) * C::PK + c * C::CH, g);
    }

    const int nkt = Sk_pad / 16;
    f32x4_t s[MAXKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
        s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (kt < nkt) {
#pragma unroll
            for (int c = 0; c < C::NCD; ++c)
                s[kt] = M::mma(M::lds_kmajor(sK + (kt * 16 + i) * C::PK + c * C::CH, g), fq[c], s[kt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                const float v = s[kt][r] * scale + key_bias(sM, key, q, p.Sk, p.causal);
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (p.lse && qv && g == 0) p.lse[((long)bh) * p.Sq + q] = mx + logf(sum);
    const float inv_keep = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint64_t drow = ((uint64_t)bh * p.Sq + q) * (uint64_t)p.Sk;
#pragma unroll
    for (int kt = 0; kt < MAXKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][r] * inv;
                if (p.p_drop > 0.f) v *= dropout_scale(seed, p.offset, drow + (uint64_t)(kt * 16 + 4 * g + r), p.p_drop, inv_keep);
                s[kt][r] = v;
            }
        }
    }
    // O^T[d, q] = sum_key V^T[d, key] P^T[key, q]
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < MAXKT / C::TPC; ++kc) {
        if (kc * C::TPC < nkt) {
            const typename M::frag fp = M::from_acc(s[kc * C::TPC], s[kc * C::TPC + C::TPC - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = M::mma(M::lds_tmajor(sV + (kc * C::CH) * C::PT + dt * 16, C::PT, lane), fp, o[dt]);
        }
    }
    if (qv) {
        T* Og = reinterpret_cast<T*>(p.out) + ((long)b * p.Sq + q) * p.ldo + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4<T>(Og + dt * 16 + 4 * g, o[dt], 1.0f);
        if (sizeof(T) == 2 && p.out_lo != nullptr) {          // lo half of the context pair (UnivlAttention.out_lo)
            T* Ol = reinterpret_cast<T*>(p.out_lo) + ((long)b * p.Sq + q) * p.ldo + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4_t r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = o[dt][e] - (float)(__bf16)o[dt][e];
                store4<T>(Ol + dt * 16 + 4 * g, r, 1.0f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// role A (blockIdx.y < nqb): dQ for a block of 64 queries.   role B: dK, dV for a block of 64 keys.
// DUAL: the matrices that are read both ways (K in role A, Q and dO in role B) are staged as a PK image and a PT image.
// FUSED (round 5, gemm.hip: attn_bwd_odgrad_kernel): the upstream gradient dO of this (batch row, head) is not in global memory -- the
// calling workgroup has just computed it as a 64 x 64 product tile (the attention-output dgrad, module_bert.py:207 differentiated) and
// `dow(image, pitch)` writes it (bf16, rows beyond Sq zero) into an LDS image.  Needs Sq_pad, Sk_pad <= 64 (one block per role).
struct AttnNoDo { __device__ void operator()(void*, int) const {} };
template <typename T, int TRIPS, bool DUAL, bool FUSED, typename DOW>
__device__ __forceinline__ void attn_bwd_body(const UnivlAttention& p, const int Sk_pad, const int Sq_pad, const int nqb, const float scale,
                                              const int bh, const int yblk, unsigned char* smem_raw, const DOW& dow) {
    using C = AttnCfg<T>;
    using M = Mma<T>;
    constexpr int PB = DUAL ? C::PK : C::PT;         // pitch of the image the K-major reads of a two-way matrix use
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int b = bh / p.H, h = bh % p.H;
    const uint64_t seed = seed_fetch(p);
    const MaskRegs mk = mask_fetch(p.key_mask, p.q, b, p.Sk, tid);
    const float inv_keep = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.Sq * p.ldq + h * HD;
    const T* Kb = reinterpret_cast<const T*>(p.k) + (long)b * (p.bsk ? p.bsk : (long)p.Sk * p.ldk) + h * HD;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)b * (p.bsv ? p.bsv : (long)p.Sk * p.ldv) + h * HD;
    const T* Ob = reinterpret_cast<const T*>(p.out) + (long)b * p.Sq * p.ldo + h * HD;
    const T* dOb = reinterpret_cast<const T*>(p.dout) + (long)b * p.Sq * p.lddo + h * HD;

    if (yblk < nqb) {
        // ---------------------------------------------------------------- role A: dQ
        T* sKt = reinterpret_cast<T*>(smem_raw);                  // transpose reads (dQ)
        T* sV = sKt + Sk_pad * C::PT;                             // K-major reads only
        T* sKk = DUAL ? sV + Sk_pad * C::PK : sKt;                // K-major reads (scores)
        float* sM = reinterpret_cast<float*>(sV + Sk_pad * C::PK + (DUAL ? Sk_pad * C::PK : 0));
        const int q0 = yblk * 64 + wave * 16;
        const int q = q0 + i;
        const bool qv = q < p.Sq;
        typename M::frag fq[C::NCD], fdo[C::NCD], fo[C::NCD];
        const int qc = min(q, p.Sq - 1);
#pragma unroll
        for (int c = 0; c < C::NCD; ++c) {                      // requested before the staging round trip (see forward)
            fq[c] = M::gmem_kmajor(Qb + (long)qc * p.ldq + c * C::CH, g);
            if constexpr (!FUSED) fdo[c] = M::gmem_kmajor(dOb + (long)qc * p.lddo + c * C::CH, g);
            fo[c] = M::gmem_kmajor(Ob + (long)qc * p.ldo + c * C::CH, g);
        }
        const float lse = p.lse[(long)bh * p.Sq + qc];
        T* sDOa = reinterpret_cast<T*>(sM + Sk_pad);              // FUSED: the dO tile, K-major reads, [64][PK] behind the mask
        if constexpr (FUSED) dow(sDOa, C::PK);
        stage_pair<T, TRIPS, C::PT, DUAL ? C::PK : 0, C::PK, 0>(sKt, sKk, Kb, p.ldk, sV, nullptr, Vb, p.ldv, p.Sk, Sk_pad, tid);
        mask_store(sM, mk, p.key_mask != nullptr, p.Sk, Sk_pad, tid);
        __syncthreads();
        if (q0 >= p.Sq) return;
        if constexpr (FUSED) {
#pragma unroll
            for (int c = 0; c < C::NCD; ++c) fdo[c] = M::lds_kmajor(sDOa + (q0 + i) * C::PK + c * C::CH, g);
        }
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < C::NCD; ++c)
#pragma unroll
            for (int e = 0; e < C::EPC; ++e) dsum += to_f32<T>(fdo[c][e]) * to_f32<T>(fo[c][e]);
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        const uint64_t drow = ((uint64_t)bh * p.Sq + q) * (uint64_t)p.Sk;
        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int nkc = Sk_pad / C::CH;
        for (int kc = 0; kc < nkc; ++kc) {
            f32x4_t ds[C::TPC];
#pragma unroll
            for (int t = 0; t < C::TPC; ++t) {
                const int kt = kc * C::TPC + t;
                f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C::NCD; ++c) {
                    st = M::mma(M::lds_kmajor(sKk + (kt * 16 + i) * PB + c * C::CH, g), fq[c], st);
                    dp = M::mma(M::lds_kmajor(sV + (kt * 16 + i) * C::PK + c * C::CH, g), fdo[c], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    const float sv = st[r] * scale + key_bias(sM, key, q, p.Sk, p.causal);
                    const float pr = (qv && key < p.Sk) ? expf(sv - lse) : 0.0f;
                    float dpr = dp[r];
                    if (p.p_drop > 0.f) dpr *= dropout_scale(seed, p.offset, drow + (uint64_t)key, p.p_drop, inv_keep);
                    ds[t][r] = pr * (dpr - dsum);
                }
            }
            const typename M::frag fds = M::from_acc(ds[0], ds[C::TPC - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dq[dt] = M::mma(M::lds_tmajor(sKt + (kc * C::CH) * C::PT + dt * 16, C::PT, lane), fds, dq[dt]);
        }
        if (qv) {
            T* dQg = reinterpret_cast<T*>(p.dq) + ((long)b * p.Sq + q) * p.lddq + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4<T>(dQg + dt * 16 + 4 * g, dq[dt], scale);
        }
    } else {
        // ---------------------------------------------------------------- role B: dK, dV
        T* sQt = reinterpret_cast<T*>(smem_raw);                  // transpose reads (dK, dV)
        T* sDOt = sQt + Sq_pad * C::PT;
        T* sQk = DUAL ? sDOt + Sq_pad * C::PT : sQt;              // K-major reads (scores, dP)
        T* sDOk = DUAL ? sQk + Sq_pad * C::PK : sDOt;
        float* sM = reinterpret_cast<float*>(sDOt + Sq_pad * C::PT + (DUAL ? 2 * Sq_pad * C::PK : 0));   // [Sk_pad]
        float* sL = sM + Sk_pad;                                      // [Sq_pad] lse
        float* sD = sL + Sq_pad;                                      // [Sq_pad] rowsum(dO*O)
        const int k0 = (yblk - nqb) * 64 + wave * 16;
        const int key = k0 + i;
        const bool kv = key < p.Sk;
        typename M::frag fk[C::NCD], fv[C::NCD];
        const int keyc = min(key, p.Sk - 1);
#pragma unroll
        for (int c = 0; c < C::NCD; ++c) {                      // requested before the staging round trip
            fk[c] = M::gmem_kmajor(Kb + (long)keyc * p.ldk + c * C::CH, g);
            fv[c] = M::gmem_kmajor(Vb + (long)keyc * p.ldv + c * C::CH, g);
        }
        // D[q] = sum_d dO[q,d] O[q,d]: 4 threads per query row, 16 d each, as 16-byte vectors from a clamped row.  With at
        // most 64 query rows (TRIPS > 0) the row of this thread is known up front and its loads join the staging loads.
        constexpr int NV = 16 / C::EPC;                          // 16-byte vectors per 16 elements
        auto d_fetch = [&](int qq, typename M::frag (&vo)[NV], typename M::frag (&vd)[NV], float& l) {
            const int qr = min(qq, p.Sq - 1);
            const T* o = Ob + (long)qr * p.ldo + (tid & 3) * 16;
            const T* d = dOb + (long)qr * p.lddo + (tid & 3) * 16;
            l = p.lse[(long)bh * p.Sq + qr];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                vo[u] = *reinterpret_cast<const typename M::frag*>(o + u * C::EPC);
                if constexpr (!FUSED) vd[u] = *reinterpret_cast<const typename M::frag*>(d + u * C::EPC);
            }
        };
        auto d_store = [&](int qq, const typename M::frag (&vo)[NV], const typename M::frag (&vd)[NV], float l) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < NV; ++u)
#pragma unroll
                for (int e = 0; e < C::EPC; ++e) acc += to_f32<T>(vo[u][e]) * to_f32<T>(vd[u][e]);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if ((tid & 3) == 0 && qq < Sq_pad) {
                sD[qq] = qq < p.Sq ? acc : 0.0f;
                sL[qq] = qq < p.Sq ? l : 0.0f;
            }
        };
        typename M::frag vo0[NV], vd0[NV];
        float l0 = 0.f;
        if (TRIPS > 0) d_fetch(tid >> 2, vo0, vd0, l0);
        if constexpr (FUSED) {
            static_assert(!FUSED || TRIPS > 0, "the fused form serves one block of at most 64 queries");
            dow(sDOt, C::PT);
            if constexpr (DUAL) dow(sDOk, C::PK);
            stage_pair<T, TRIPS, C::PT, DUAL ? C::PK : 0, C::PT, 0, true>(sQt, sQk, Qb, p.ldq, nullptr, nullptr, Qb, p.ldq, p.Sq, Sq_pad, tid);
            mask_store(sM, mk, p.key_mask != nullptr, p.Sk, Sk_pad, tid);
            __syncthreads();                                      // the dO images are complete: this thread's 16 elements of its query row
            const int qq = tid >> 2;
#pragma unroll
            for (int u = 0; u < NV; ++u)
                vd0[u] = *reinterpret_cast<const typename M::frag*>(sDOk + min(qq, Sq_pad - 1) * PB + (tid & 3) * 16 + u * C::EPC);
        } else {
        stage_pair<T, TRIPS, C::PT, DUAL ? C::PK : 0, C::PT, DUAL ? C::PK : 0>(sQt, sQk, Qb, p.ldq, sDOt, sDOk, dOb, p.lddo, p.Sq, Sq_pad, tid);
        mask_store(sM, mk, p.key_mask != nullptr, p.Sk, Sk_pad, tid);
        }
        if (TRIPS > 0) {
            d_store(tid >> 2, vo0, vd0, l0);
        } else {
            for (int qq = tid >> 2; qq < Sq_pad; qq += 64) {
                typename M::frag vo[NV], vd[NV];
                float l;
                d_fetch(qq, vo, vd, l);
                d_store(qq, vo, vd, l);
            }
        }
        __syncthreads();
        if (k0 >= p.Sk) return;
        const float mkey = kv ? sM[key] : 0.0f;
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        const int nqc = Sq_pad / C::CH;
        for (int qc = 0; qc < nqc; ++qc) {
            f32x4_t ds[C::TPC], pd[C::TPC];
#pragma unroll
            for (int t = 0; t < C::TPC; ++t) {
                const int qt = qc * C::TPC + t;
                f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C::NCD; ++c) {
                    st = M::mma(M::lds_kmajor(sQk + (qt * 16 + i) * PB + c * C::CH, g), fk[c], st);
                    dp = M::mma(M::lds_kmajor(sDOk + (qt * 16 + i) * PB + c * C::CH, g), fv[c], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt * 16 + 4 * g + r;
                    float m = mkey;
                    if (p.causal && key > q) m = -10000.0f;
                    const float sv = st[r] * scale + m;
                    const float pr = (kv && q < p.Sq) ? expf(sv - sL[q]) : 0.0f;
                    float dsc = 1.0f;
                    if (p.p_drop > 0.f)
                        dsc = dropout_scale(seed, p.offset, ((uint64_t)bh * p.Sq + q) * (uint64_t)p.Sk + (uint64_t)key, p.p_drop, inv_keep);
                    pd[t][r] = pr * dsc;
                    ds[t][r] = pr * (dp[r] * dsc - sD[q]);
                }
            }
            const typename M::frag fpd = M::from_acc(pd[0], pd[C::TPC - 1]);
            const typename M::frag fds = M::from_acc(ds[0], ds[C::TPC - 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = M::mma(M::lds_tmajor(sDOt + (qc * C::CH) * C::PT + dt * 16, C::PT, lane), fpd, dv[dt]);
                dk[dt] = M::mma(M::lds_tmajor(sQt + (qc * C::CH) * C::PT + dt * 16, C::PT, lane), fds, dk[dt]);
            }
        }
        if (kv) {
            T* dKg = reinterpret_cast<T*>(p.dk) + ((long)b * p.Sk + key) * p.lddk + h * HD;
            T* dVg = reinterpret_cast<T*>(p.dv) + ((long)b * p.Sk + key) * p.lddv + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store4<T>(dKg + dt * 16 + 4 * g, dk[dt], scale);
                store4<T>(dVg + dt * 16 + 4 * g, dv[dt], 1.0f);
            }
        }
    }
}


static inline int attn_check(const UnivlAttention* d, const char* who, bool bwd) {
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "%s: null descriptor", who);
    UNIVL_CHECK_ARG(d->dtype == UNIVL_DT_F32 || d->dtype == UNIVL_DT_BF16, UNIVL_EUNSUPPORTED, "%s: dtype %d", who, d->dtype);
    UNIVL_CHECK_ARG(d->B > 0 && d->H > 0 && d->Sq > 0 && d->Sk > 0, UNIVL_EINVAL, "%s: empty problem", who);
    const int lim = d->dtype == UNIVL_DT_BF16 ? 384 : 256;
    UNIVL_CHECK_ARG(d->Sk <= lim && d->Sq <= lim, UNIVL_EUNSUPPORTED,
                    "%s: sequence %dx%d exceeds the single-pass LDS limit %d", who, d->Sq, d->Sk, lim);
    const int epc = d->dtype == UNIVL_DT_BF16 ? 8 : 4;
    UNIVL_CHECK_ARG(d->q && d->k && d->v && d->out, UNIVL_EINVAL, "%s: null q/k/v/out", who);
    UNIVL_CHECK_ARG(aligned16(d->q) && aligned16(d->k) && aligned16(d->v) && aligned16(d->out) &&
                        d->ldq % epc == 0 && d->ldk % epc == 0 && d->ldv % epc == 0 && d->ldo % epc == 0,
                    UNIVL_EALIGN, "%s: q/k/v/out must be 16-byte aligned with ld %% %d == 0", who, epc);
    if (bwd) {
        UNIVL_CHECK_ARG(d->dout && d->dq && d->dk && d->dv && d->lse, UNIVL_EINVAL, "%s: null dout/dq/dk/dv/lse", who);
        UNIVL_CHECK_ARG(aligned16(d->dout) && aligned16(d->dq) && aligned16(d->dk) && aligned16(d->dv) &&
                            d->lddo % epc == 0 && d->lddq % epc == 0 && d->lddk % epc == 0 && d->lddv % epc == 0,
                        UNIVL_EALIGN, "%s: gradients must be 16-byte aligned with ld %% %d == 0", who, epc);
    }
    return UNIVL_OK;
}

