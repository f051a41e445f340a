// TF-style LayerNorm (reference: modules/until_module.py:40-53) fused with the elementwise work that surrounds it
// at every call site of the reference (residual add, dropout before / after, position-embedding add, the
// float64 -> float32 cast of NormalizeVideo, modeling.py:88-92).  HBM-bound: one wave per row, the row lives in
// registers (N/64 = 12 or 16 floats per lane), 16-byte coalesced accesses, wave-shuffle reductions, two-pass
// (mean, then centred variance) exactly as the reference computes it.
#include <stdlib.h>
#include "common.h"
#include "univl_hip.h"
#include "ln_body.h"

namespace {

template <int N, typename TO, bool F64>
__global__ __launch_bounds__(256) void ln_fwd_kernel(UnivlLayerNorm p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    ln_fwd_row<N, TO, F64>(p, row, lane);
}

// Backward.  Each wave walks RPW rows, keeps per-column partial sums of dgamma / dbeta / dbias in registers, the
// block combines its 4 waves through LDS and issues one fp32 atomic per column per block.
constexpr int LN_RPW = 4;

template <int N, int NW = 4>
__device__ __forceinline__ void block_colsum(float (*red)[N], const float (&part)[N / 256][4], float* dst, int lane, int wave) {
    if (dst == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N / 256; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][4 * lane + 256 * j + e] = part[j][e];
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 64 * NW) {
        float s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
#pragma unroll
        for (int w = 4; w < NW; w += 4) s += (red[w][c] + red[w + 1][c]) + (red[w + 2][c] + red[w + 3][c]);
        unsafeAtomicAdd(dst + c, s);
    }
}

// Deterministic form (common.h): the block's column sums go to its slot of a scratch array instead of an atomic.
template <int N>
__device__ __forceinline__ void block_colsum_store(float (*red)[N], const float (&part)[N / 256][4], float* slot, int lane, int wave) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N / 256; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][4 * lane + 256 * j + e] = part[j][e];
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) slot[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// DET: dgamma / dbeta / dbias partials per block in det_part[3][gridDim.x][N], the last block to arrive adds them in block order;
// position-row gradients are not scattered here: the fp32 dx rows (p.dx32, which the host points at scratch when the caller gave
// none) are summed per position by ln_dpos_gather_kernel afterwards.
// NW = 8 (512 threads; round 5, thousands of rows): every workgroup ends in 3 x N fp32 atomics onto the same 3 N addresses, and an
// address takes one update at a time -- at 6144 rows and 384 four-wave workgroups the column sums were 9 of the kernel's 24 us
// (profiles/r05l_mb_ln_parts.txt: 15.1 us without them) while only 6 waves per compute unit kept one row each in flight.  (Sixteen
// waves would halve the updates again, but cap the kernel at 128 VGPRs: 23 spilled.)
template <int N, typename TO, bool DET = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void ln_bwd_kernel(UnivlLayerNorm p, int rpw, float* det_part = nullptr, int* det_counter = nullptr) {
    constexpr int NV = N / 256;
    static_assert(!DET || NW == 4, "the deterministic form keeps its 4-wave summation order");
    __shared__ float red[NW][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t* sp = p.seed_dev ? p.seed_dev : reinterpret_cast<const uint64_t*>(p.gamma);
    const uint64_t sdv = *sp;
    const uint64_t seed = p.seed + (p.seed_dev ? sdv : 0ull);
    const float inv_keep_pre = p.p_pre > 0.f ? 1.0f / (1.0f - p.p_pre) : 1.0f;
    const float inv_keep_post = p.p_post > 0.f ? 1.0f / (1.0f - p.p_post) : 1.0f;
    float dg[NV][4], db[NV][4], dbi[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[j][e] = 0.f; db[j][e] = 0.f; dbi[j][e] = 0.f; }
    float ga[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(p.gamma + 4 * lane + 256 * j);
        ga[j][0] = t.x; ga[j][1] = t.y; ga[j][2] = t.z; ga[j][3] = t.w;
    }
    for (int rr = 0; rr < rpw; ++rr) {
        const int row = (blockIdx.x * NW + wave) * rpw + rr;
        if (row >= p.rows) break;
        const float mean = p.stats[2 * (long)row], rstd = p.stats[2 * (long)row + 1];
        float dy[NV][4], xh[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const long o = (long)row * N + 4 * lane + 256 * j;
            const float4 a = *reinterpret_cast<const float4*>(p.dout + o);
            const float4 yy = *reinterpret_cast<const float4*>(p.y + o);
            dy[j][0] = a.x; dy[j][1] = a.y; dy[j][2] = a.z; dy[j][3] = a.w;
            xh[j][0] = (yy.x - mean) * rstd; xh[j][1] = (yy.y - mean) * rstd;
            xh[j][2] = (yy.z - mean) * rstd; xh[j][3] = (yy.w - mean) * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.p_post > 0.f) dy[j][e] *= dropout_scale(seed, p.off_post, (uint64_t)(o + e), p.p_post, inv_keep_post);
                const float gq = dy[j][e] * ga[j][e];
                s1 += gq; s2 += gq * xh[j][e];
                dg[j][e] += dy[j][e] * xh[j][e];
                db[j][e] += dy[j][e];
            }
        }
        s1 = wave_sum(s1) * (1.0f / N);
        s2 = wave_sum(s2) * (1.0f / N);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int col = 4 * lane + 256 * j;
            const long o = (long)row * N + col;
            float dx[4], dd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dx[e] = rstd * (dy[j][e] * ga[j][e] - s1 - xh[j][e] * s2);
                dd[e] = dx[e];
                if (p.p_pre > 0.f) dd[e] *= dropout_scale(seed, p.off_pre, (uint64_t)(o + e), p.p_pre, inv_keep_pre);
                dbi[j][e] += dd[e];
            }
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
            if (!DET && p.dpos) {
                float* dp = p.dpos + (long)(row % p.pos_period) * N + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dp + e, dx[e]);
            }
        }
    }
    if constexpr (DET) {
        __shared__ int last_flag;
        const long nb = gridDim.x;
        block_colsum_store<N>(red, dg, det_part + (0 * nb + blockIdx.x) * N, lane, wave);
        block_colsum_store<N>(red, db, det_part + (1 * nb + blockIdx.x) * N, lane, wave);
        block_colsum_store<N>(red, dbi, det_part + (2 * nb + blockIdx.x) * N, lane, wave);
        if (det_last_block(det_counter, (int)nb, &last_flag)) {
            float* dst[3] = {p.dgamma, p.dbeta, p.dbias};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (dst[k] == nullptr) continue;
                for (int c = threadIdx.x; c < N; c += 256) {
                    const float* q = det_part + (long)k * nb * N + c;
                    float acc = 0.f;
                    for (long b = 0; b < nb; ++b) acc += q[b * N];
                    dst[k][c] += acc;
                }
            }
        }
    } else {
    // block reduction of the three column sums, one LDS pass each (pointers are block-uniform)
    block_colsum<N, NW>(red, dg, p.dgamma, lane, wave);
    block_colsum<N, NW>(red, db, p.dbeta, lane, wave);
    block_colsum<N, NW>(red, dbi, p.dbias, lane, wave);
    }
}

// dpos[s, c] += sum over the rows r with r % period == s of dx[r, c], in row order (deterministic form of the scatter above)
__global__ __launch_bounds__(256) void ln_dpos_gather_kernel(const float* dx, int rows, int period, int n, float* dpos) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)period * n) return;
    const int s = (int)(i / n), c = (int)(i % n);
    // four independent partial sums (rows s, s + period, s + 2 period, s + 3 period of every group of four), folded in a fixed
    // order: four loads in flight per thread instead of one dependent add per row
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long r = s;
    const long p = period;
    for (; r + 3 * p < rows; r += 4 * p) {
        a0 += dx[r * n + c]; a1 += dx[(r + p) * n + c]; a2 += dx[(r + 2 * p) * n + c]; a3 += dx[(r + 3 * p) * n + c];
    }
    if (r < rows) a0 += dx[r * n + c];
    if (r + p < rows) a1 += dx[(r + p) * n + c];
    if (r + 2 * p < rows) a2 += dx[(r + 2 * p) * n + c];
    dpos[i] += (a0 + a1) + (a2 + a3);
}

int check_common(const UnivlLayerNorm* d, const char* who) {
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "%s: null descriptor", who);
    UNIVL_CHECK_ARG(d->N == 768 || d->N == 1024, UNIVL_EUNSUPPORTED, "%s: N=%d (768 or 1024 supported)", who, d->N);
    UNIVL_CHECK_ARG(d->rows > 0, UNIVL_EINVAL, "%s: rows=%d", who, d->rows);
    UNIVL_CHECK_ARG(d->dtype == UNIVL_DT_F32 || d->dtype == UNIVL_DT_BF16, UNIVL_EUNSUPPORTED, "%s: dtype %d", who, d->dtype);
    UNIVL_CHECK_ARG(d->gamma != nullptr, UNIVL_EINVAL, "%s: gamma is null", who);
    UNIVL_CHECK_ARG(!(d->pos || d->dpos) || d->pos_period > 0, UNIVL_EINVAL, "%s: pos_period must be > 0", who);
    return UNIVL_OK;
}

}  // namespace

extern "C" int univl_layernorm_fwd(const UnivlLayerNorm* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    int rc = check_common(d, "univl_layernorm_fwd");
    if (rc) return rc;
    UNIVL_CHECK_ARG(d->x && d->beta && (d->out32 || d->out16), UNIVL_EINVAL, "univl_layernorm_fwd: null x/beta/out");
    UNIVL_CHECK_ARG(aligned16(d->x) && aligned16(d->out32) && aligned16(d->out16) && aligned16(d->y) &&
                        aligned16(d->residual) && aligned16(d->pos) && aligned16(d->gamma) && aligned16(d->beta),
                    UNIVL_EALIGN, "univl_layernorm_fwd: pointers must be 16-byte aligned");
    dim3 grid((d->rows + 3) / 4), block(256);
#define LN_FWD(NN, TT, FF) hipLaunchKernelGGL((ln_fwd_kernel<NN, TT, FF>), grid, block, 0, stream, *d)
    const bool bf = d->dtype == UNIVL_DT_BF16;
    if (d->N == 768) {
        if (d->x_f64) { if (bf) LN_FWD(768, __bf16, true); else LN_FWD(768, float, true); }
        else          { if (bf) LN_FWD(768, __bf16, false); else LN_FWD(768, float, false); }
    } else {
        if (d->x_f64) { if (bf) LN_FWD(1024, __bf16, true); else LN_FWD(1024, float, true); }
        else          { if (bf) LN_FWD(1024, __bf16, false); else LN_FWD(1024, float, false); }
    }
#undef LN_FWD
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_layernorm_bwd(const UnivlLayerNorm* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    int rc = check_common(d, "univl_layernorm_bwd");
    if (rc) return rc;
    UNIVL_CHECK_ARG(d->dout && d->y && d->stats, UNIVL_EINVAL, "univl_layernorm_bwd: null dout/y/stats");
    UNIVL_CHECK_ARG(aligned16(d->dout) && aligned16(d->y) && aligned16(d->dx32) && aligned16(d->dxd32) &&
                        aligned16(d->dxd16) && aligned16(d->gamma),
                    UNIVL_EALIGN, "univl_layernorm_bwd: pointers must be 16-byte aligned");
    // rows per wave: 1 while the grid is small (parallelism first), more once it fills the chip -- every workgroup issues 3 x N fp32
    // atomics for its column sums, whatever the number of rows it walked.  Measured per launch (profiles/r04f_mb_ln_bwd_rows_per_wave.txt,
    // 1 / 2 / 4 rows per wave): 192 rows 6.1 / 7.2 / 10.3 us; 768 rows 10.4 / 9.1 / 11.3; 1536 rows 17.6 / 13.3 / 12.9; 6144 rows
    // 47.0 / 30.6 / 23.3 (the round-3 heuristic rows / 2048 gave 10.3 / 17.0 / 24.8 at 768 / 1536 / 6144 rows).
    int rpw = d->rows < 640 ? 1 : (d->rows < 1280 ? 2 : 4);
#ifdef UNIVL_TRACE
    if (const char* e = getenv("UNIVL_LN_RPW")) { if (atoi(e) > 0) rpw = atoi(e); }      // measurement build only (scripts/mb_ln_bwd.py)
#endif
    rpw = rpw < 1 ? 1 : (rpw > LN_RPW ? LN_RPW : rpw);
    dim3 grid((d->rows + 4 * rpw - 1) / (4 * rpw)), block(256);
    const bool bf = d->dtype == UNIVL_DT_BF16;
    if (univl_deterministic()) {
        UnivlLayerNorm q = *d;
        float* part = static_cast<float*>(univl_det_alloc((size_t)3 * grid.x * d->N * sizeof(float)));
        int* counter = univl_det_counter();
        if (!part || !counter) return UNIVL_EINVAL;
        if (q.dpos && !q.dx32) {
            q.dx32 = static_cast<float*>(univl_det_alloc((size_t)d->rows * d->N * sizeof(float)));
            if (!q.dx32) return UNIVL_EINVAL;
        }
        if (d->N == 768) {
            if (bf) hipLaunchKernelGGL((ln_bwd_kernel<768, __bf16, true>), grid, block, 0, stream, q, rpw, part, counter);
            else hipLaunchKernelGGL((ln_bwd_kernel<768, float, true>), grid, block, 0, stream, q, rpw, part, counter);
        } else {
            if (bf) hipLaunchKernelGGL((ln_bwd_kernel<1024, __bf16, true>), grid, block, 0, stream, q, rpw, part, counter);
            else hipLaunchKernelGGL((ln_bwd_kernel<1024, float, true>), grid, block, 0, stream, q, rpw, part, counter);
        }
        if (q.dpos)
            hipLaunchKernelGGL(ln_dpos_gather_kernel, dim3((unsigned)(((long)q.pos_period * d->N + 255) / 256)), block, 0, stream,
                               q.dx32, d->rows, q.pos_period, d->N, q.dpos);
        UNIVL_LAUNCH_CHECK();
        return UNIVL_OK;
    }
    // 8 waves x 3 rows from 3072 rows on: 20.5 vs 23.7 us at 6144 rows, 14.4 vs 16.7 at 3072 (profiles/r05m_mb_ln_nw.txt)
    int nw = (d->rows >= 3072 && d->N == 768 && !d->dpos) ? 8 : 4;
#ifdef UNIVL_TRACE
    if (const char* e = getenv("UNIVL_LN_NW")) { if (atoi(e) == 4 || atoi(e) == 8) nw = atoi(e); }      // measurement build only
#endif
    if (nw == 8 && d->N == 768) {
        int rp = 3;
#ifdef UNIVL_TRACE
        if (const char* e = getenv("UNIVL_LN_RPW")) { if (atoi(e) > 0) rp = atoi(e); }
#endif
        dim3 g16((d->rows + 8 * rp - 1) / (8 * rp)), b16(512);
        if (bf) hipLaunchKernelGGL((ln_bwd_kernel<768, __bf16, false, 8>), g16, b16, 0, stream, *d, rp);
        else hipLaunchKernelGGL((ln_bwd_kernel<768, float, false, 8>), g16, b16, 0, stream, *d, rp);
        UNIVL_LAUNCH_CHECK();
        return UNIVL_OK;
    }
    if (d->N == 768) {
        if (bf) hipLaunchKernelGGL((ln_bwd_kernel<768, __bf16>), grid, block, 0, stream, *d, rpw);
        else hipLaunchKernelGGL((ln_bwd_kernel<768, float>), grid, block, 0, stream, *d, rpw);
    } else {
        if (bf) hipLaunchKernelGGL((ln_bwd_kernel<1024, __bf16>), grid, block, 0, stream, *d, rpw);
        else hipLaunchKernelGGL((ln_bwd_kernel<1024, float>), grid, block, 0, stream, *d, rpw);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// out[s, c] += sum over the rows r = s, s + period, ... of rows[r, c]  (r ascending; fixed order, no atomics): the position-table
// gradient of an embedding layer from the per-token gradient rows.  With B rows per position the scatter-add of the fused backward
// kernels puts B fp32 atomics on every element of a [period, n] table -- at 128 pairs that contention is most of those kernels
// (embed_bwd 173 us, the video embedding's LayerNorm backward 189 us); the callers switch to this gather from 32 rows per position.
extern "C" int univl_rows_gather_sum(const float* rows, int32_t n_rows, int32_t period, int32_t n, float* out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(rows && out && n_rows > 0 && period > 0 && n > 0, UNIVL_EINVAL, "univl_rows_gather_sum: bad argument");
    hipLaunchKernelGGL(ln_dpos_gather_kernel, dim3((unsigned)(((long)period * n + 255) / 256)), dim3(256), 0, stream, rows, n_rows, period, n, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
