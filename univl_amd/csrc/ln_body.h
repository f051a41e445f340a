// One row of the TF-style LayerNorm forward (reference: modules/until_module.py:40-53) with the elementwise work around it (dropout
// before / after, residual, position row, float64 input): the body of layernorm.hip's ln_fwd_kernel as a device function, so that a
// GEMM launch can finish the LayerNorm that consumes its output itself (gemm.hip: ln_fold).  One wave per row; `lane` = lane of the wave.
#pragma once
#include "common.h"
#include "univl_hip.h"

// XC: x was produced by OTHER workgroups of the running launch (fp32 atomics): its loads carry agent scope (sc1), so they are served
// from the point the atomics were performed at, not from a line this XCD's L2 may still hold.
template <int N, typename TO, bool F64, bool XC = false>
__device__ __forceinline__ void ln_fwd_row(const UnivlLayerNorm& p, const int row, const int lane) {
    constexpr int NV = N / 256;
    float v[NV][4];
    // Every global operand of the row (seed, x, residual, position row, gamma, beta) is requested up front, from
    // SELECTED (always valid) pointers instead of inside "if (ptr)" blocks: a branch per operand kind costs one full
    // memory round trip each (hipcc drains the loads in flight at every join), five in a row for the post-GEMM LayerNorm.
    const uint64_t* sp = p.seed_dev ? p.seed_dev : reinterpret_cast<const uint64_t*>(p.gamma);
    const uint64_t sdv = *sp;
    const uint64_t seed = p.seed + (p.seed_dev ? sdv : 0ull);
    const float inv_keep_pre = p.p_pre > 0.f ? 1.0f / (1.0f - p.p_pre) : 1.0f;
    const float inv_keep_post = p.p_post > 0.f ? 1.0f / (1.0f - p.p_post) : 1.0f;
    const float* rp = p.residual ? p.residual + (long)row * N : p.gamma;
    const float* pp = p.pos ? p.pos + (long)(row % (p.pos ? p.pos_period : 1)) * N : p.gamma;
    float4 rr[NV], pr[NV], gav[NV], bev[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const long o = (long)row * N + 4 * lane + 256 * j;
        if (F64) {
            const double* x = reinterpret_cast<const double*>(p.x) + o;
            const double2 a = *reinterpret_cast<const double2*>(x);
            const double2 b = *reinterpret_cast<const double2*>(x + 2);
            v[j][0] = (float)a.x; v[j][1] = (float)a.y; v[j][2] = (float)b.x; v[j][3] = (float)b.y;
        } else if (XC) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(reinterpret_cast<const float*>(p.x) + (long)row * N), 0, N * 4, 0x00020000);
            const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs, (4 * lane + 256 * j) * 4, 0, 16);      // aux 16 = sc1
            v[j][0] = __uint_as_float(a[0]); v[j][1] = __uint_as_float(a[1]); v[j][2] = __uint_as_float(a[2]); v[j][3] = __uint_as_float(a[3]);
        } else {
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.x) + o);
            v[j][0] = a.x; v[j][1] = a.y; v[j][2] = a.z; v[j][3] = a.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = 4 * lane + 256 * j;
        rr[j] = *reinterpret_cast<const float4*>(rp + col);
        pr[j] = *reinterpret_cast<const float4*>(pp + col);
        gav[j] = *reinterpret_cast<const float4*>(p.gamma + col);
        bev[j] = *reinterpret_cast<const float4*>(p.beta + col);
    }
    if (p.p_pre > 0.f) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const long o = (long)row * N + 4 * lane + 256 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] *= dropout_scale(seed, p.off_pre, (uint64_t)(o + e), p.p_pre, inv_keep_pre);
        }
    }
    const bool useR = p.residual != nullptr, useP = p.pos != nullptr;      // selects, not multiplies: dummy words may be anything
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        v[j][0] += (useR ? rr[j].x : 0.f) + (useP ? pr[j].x : 0.f); v[j][1] += (useR ? rr[j].y : 0.f) + (useP ? pr[j].y : 0.f);
        v[j][2] += (useR ? rr[j].z : 0.f) + (useP ? pr[j].z : 0.f); v[j][3] += (useR ? rr[j].w : 0.f) + (useP ? pr[j].w : 0.f);
    }
    if (p.y) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            *reinterpret_cast<float4*>(p.y + (long)row * N + 4 * lane + 256 * j) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float c = v[j][e] - mean; q += c * c; }
    const float var = wave_sum(q) * (1.0f / N);
    const float rstd = 1.0f / sqrtf(var + p.eps);
    if (p.stats && lane == 0) { p.stats[2 * (long)row] = mean; p.stats[2 * (long)row + 1] = rstd; }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = 4 * lane + 256 * j;
        const long o = (long)row * N + col;
        const float4 ga = gav[j], be = bev[j];
        float r[4];
        r[0] = (v[j][0] - mean) * rstd * ga.x + be.x;
        r[1] = (v[j][1] - mean) * rstd * ga.y + be.y;
        r[2] = (v[j][2] - mean) * rstd * ga.z + be.z;
        r[3] = (v[j][3] - mean) * rstd * ga.w + be.w;
        if (p.p_post > 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] *= dropout_scale(seed, p.off_post, (uint64_t)(o + e), p.p_post, inv_keep_post);
        }
        if (p.out32) *reinterpret_cast<float4*>(p.out32 + o) = make_float4(r[0], r[1], r[2], r[3]);
        if (p.out16) {
            TO* d = reinterpret_cast<TO*>(p.out16) + o;
            if (sizeof(TO) == 2) {
                bf16x4_t w;
                w[0] = (__bf16)r[0]; w[1] = (__bf16)r[1]; w[2] = (__bf16)r[2]; w[3] = (__bf16)r[3];
                *reinterpret_cast<bf16x4_t*>(d) = w;
                if (p.out16_lo) {                         // lo half of the output pair (UnivlLayerNorm.out16_lo)
                    bf16x4_t l;
                    l[0] = (__bf16)(r[0] - (float)w[0]); l[1] = (__bf16)(r[1] - (float)w[1]);
                    l[2] = (__bf16)(r[2] - (float)w[2]); l[3] = (__bf16)(r[3] - (float)w[3]);
                    *reinterpret_cast<bf16x4_t*>(reinterpret_cast<TO*>(p.out16_lo) + o) = l;
                }
            } else {
                *reinterpret_cast<float4*>(d) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
    }
}

// Backward of the same LayerNorm for NW consecutive rows, one per wave of an NW-wave workgroup (the body of layernorm.hip's ln_bwd_kernel
// at one row per wave, without the position-row scatter): dx32 / dxd32 / dxd16 rows, and the workgroup's column sums of dgamma / dbeta /
// dbias through `red` ([NW][N] floats of LDS) as one fp32 atomic per column.  EVERY thread of the workgroup must call it (barriers);
// rows at or beyond p.rows contribute nothing.  XC: dout was produced by other workgroups of the running launch (see ln_fwd_row).
template <int N, typename TO, int NW, bool XC>
__device__ __forceinline__ void ln_bwd_rows(const UnivlLayerNorm& p, const int row0, float* red) {
    constexpr int NV = N / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = row0 + wave;
    const bool live = row < p.rows;
    const int rc = live ? row : p.rows - 1;             // clamped: every load below is valid, nothing is stored for a dead row
    const uint64_t* sp = p.seed_dev ? p.seed_dev : reinterpret_cast<const uint64_t*>(p.gamma);
    const uint64_t sdv = *sp;
    const uint64_t seed = p.seed + (p.seed_dev ? sdv : 0ull);
    const float inv_keep_pre = p.p_pre > 0.f ? 1.0f / (1.0f - p.p_pre) : 1.0f;
    const float inv_keep_post = p.p_post > 0.f ? 1.0f / (1.0f - p.p_post) : 1.0f;
    float ga[NV][4], dy[NV][4], xh[NV][4], dg[NV][4], db[NV][4], dbi[NV][4];
    const float mean = p.stats[2 * (long)rc], rstd = p.stats[2 * (long)rc + 1];
    float4 yy[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const long o = (long)rc * N + 4 * lane + 256 * j;
        if constexpr (XC) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout + (long)rc * N), 0, N * 4, 0x00020000);
            const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs, (4 * lane + 256 * j) * 4, 0, 16);      // aux 16 = sc1
            dy[j][0] = __uint_as_float(a[0]); dy[j][1] = __uint_as_float(a[1]); dy[j][2] = __uint_as_float(a[2]); dy[j][3] = __uint_as_float(a[3]);
        } else {
            const float4 a = *reinterpret_cast<const float4*>(p.dout + o);
            dy[j][0] = a.x; dy[j][1] = a.y; dy[j][2] = a.z; dy[j][3] = a.w;
        }
        yy[j] = *reinterpret_cast<const float4*>(p.y + o);
        const float4 t = *reinterpret_cast<const float4*>(p.gamma + 4 * lane + 256 * j);
        ga[j][0] = t.x; ga[j][1] = t.y; ga[j][2] = t.z; ga[j][3] = t.w;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const long o = (long)rc * N + 4 * lane + 256 * j;
        xh[j][0] = (yy[j].x - mean) * rstd; xh[j][1] = (yy[j].y - mean) * rstd;
        xh[j][2] = (yy[j].z - mean) * rstd; xh[j][3] = (yy[j].w - mean) * rstd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p.p_post > 0.f) dy[j][e] *= dropout_scale(seed, p.off_post, (uint64_t)(o + e), p.p_post, inv_keep_post);
            if (!live) dy[j][e] = 0.f;
            const float gq = dy[j][e] * ga[j][e];
            s1 += gq; s2 += gq * xh[j][e];
            dg[j][e] = dy[j][e] * xh[j][e];
            db[j][e] = dy[j][e];
        }
    }
    s1 = wave_sum(s1) * (1.0f / N);
    s2 = wave_sum(s2) * (1.0f / N);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const long o = (long)rc * N + 4 * lane + 256 * j;
        float dx[4], dd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dx[e] = rstd * (dy[j][e] * ga[j][e] - s1 - xh[j][e] * s2);
            dd[e] = dx[e];
            if (p.p_pre > 0.f) dd[e] *= dropout_scale(seed, p.off_pre, (uint64_t)(o + e), p.p_pre, inv_keep_pre);
            dbi[j][e] = live ? dd[e] : 0.f;
        }
        if (live) {
            if (p.dx32) *reinterpret_cast<float4*>(p.dx32 + o) = make_float4(dx[0], dx[1], dx[2], dx[3]);
            if (p.dxd32) *reinterpret_cast<float4*>(p.dxd32 + o) = make_float4(dd[0], dd[1], dd[2], dd[3]);
            if (p.dxd16) {
                TO* d = reinterpret_cast<TO*>(p.dxd16) + o;
                if (sizeof(TO) == 2) {
                    bf16x4_t w;
                    w[0] = (__bf16)dd[0]; w[1] = (__bf16)dd[1]; w[2] = (__bf16)dd[2]; w[3] = (__bf16)dd[3];
                    *reinterpret_cast<bf16x4_t*>(d) = w;
                } else {
                    *reinterpret_cast<float4*>(d) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                }
            }
        }
    }
    // the workgroup's column sums: one LDS pass and one atomic per column for each of the three (pointers are workgroup-uniform)
    auto colsum = [&](const float (&part)[NV][4], float* dst) {
        if (dst == nullptr) return;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave * N + 4 * lane + 256 * j + e] = part[j][e];
        __syncthreads();
        for (int c = threadIdx.x; c < N; c += 64 * NW) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w * N + c];
            unsafeAtomicAdd(dst + c, s);
        }
    };
    colsum(dg, p.dgamma);
    colsum(db, p.dbeta);
    colsum(dbi, p.dbias);
}
