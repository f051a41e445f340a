// Text / decoder embeddings (reference: BertEmbeddings.forward modules/module_bert.py:132-146, DecoderEmbeddings
// module_decoder.py:309-320): gather word + position (+ token-type) rows, TF-style LayerNorm, dropout.
// One wave per token; the 768-wide row lives in registers.  Backward: LayerNorm backward in registers, then
// scatter-add (fp32 atomics) into the word / position / type tables -- tied tables (decoder <-> BERT,
// modeling.py:137-138) simply receive several scatter passes.
#include "common.h"
#include "univl_hip.h"

namespace {

constexpr int N = 768;
constexpr int NV = N / 256;

template <typename TO>
__global__ __launch_bounds__(256) void embed_fwd_kernel(UnivlEmbedText p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.B * p.S) return;
    const int s = row % p.S;
    const long id = p.ids[row];
    const long tt = (p.type && p.type_ids) ? p.type_ids[row] : 0;
    float v[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = 4 * lane + 256 * j;
        const float4 a = *reinterpret_cast<const float4*>(p.word + id * N + col);
        const float4 b = *reinterpret_cast<const float4*>(p.pos + (long)s * N + col);
        v[j][0] = a.x + b.x; v[j][1] = a.y + b.y; v[j][2] = a.z + b.z; v[j][3] = a.w + b.w;
        if (p.type) {
            const float4 c = *reinterpret_cast<const float4*>(p.type + tt * N + col);
            v[j][0] += c.x; v[j][1] += c.y; v[j][2] += c.z; v[j][3] += c.w;
        }
        if (p.y) *reinterpret_cast<float4*>(p.y + (long)row * N + col) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
    }
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) sm += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(sm) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float c = v[j][e] - mean; q += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / N) + p.eps);
    if (p.stats && lane == 0) { p.stats[2 * (long)row] = mean; p.stats[2 * (long)row + 1] = rstd; }
    const uint64_t* sp = p.seed_dev ? p.seed_dev : reinterpret_cast<const uint64_t*>(p.gamma);
    const uint64_t sdv = *sp;
    const uint64_t seed = p.seed + (p.seed_dev ? sdv : 0ull);
    const float inv_keep = p.p_post > 0.f ? 1.0f / (1.0f - p.p_post) : 1.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = 4 * lane + 256 * j;
        const long o = (long)row * N + col;
        const float4 ga = *reinterpret_cast<const float4*>(p.gamma + col);
        const float4 be = *reinterpret_cast<const float4*>(p.beta + col);
        float r[4];
        r[0] = (v[j][0] - mean) * rstd * ga.x + be.x;
        r[1] = (v[j][1] - mean) * rstd * ga.y + be.y;
        r[2] = (v[j][2] - mean) * rstd * ga.z + be.z;
        r[3] = (v[j][3] - mean) * rstd * ga.w + be.w;
        if (p.p_post > 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] *= dropout_scale(seed, p.off_post, (uint64_t)(o + e), p.p_post, inv_keep);
        }
        if (p.out32) *reinterpret_cast<float4*>(p.out32 + o) = make_float4(r[0], r[1], r[2], r[3]);
        if (p.out16) {
            TO* d = reinterpret_cast<TO*>(p.out16) + o;
            if (sizeof(TO) == 2) {
                bf16x4_t w;
                w[0] = (__bf16)r[0]; w[1] = (__bf16)r[1]; w[2] = (__bf16)r[2]; w[3] = (__bf16)r[3];
                *reinterpret_cast<bf16x4_t*>(d) = w;
                if (p.out16_lo) {                         // lo half of the output pair (UnivlEmbedText.out16_lo)
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

// DET (deterministic mode, common.h): nothing is scattered from here.  Every token's table-gradient row goes to p.drows (scratch
// when the caller gave none); the word / position / token-type tables are then filled by the gather kernels below, each
// destination row summing its source rows in token order.  dgamma / dbeta: per-block partials in det_part[2][gridDim.x][N], added
// in block order by the last block to arrive.
template <bool DET = false>
__global__ __launch_bounds__(256) void embed_bwd_kernel(UnivlEmbedText p, float* det_part = nullptr, int* det_counter = nullptr) {
    __shared__ float red[4][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const bool valid = row < p.B * p.S;
    float dgm[NV][4], dbt[NV][4], dxs[NV][4];
    int row_type = -1;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dgm[j][e] = 0.f; dbt[j][e] = 0.f; dxs[j][e] = 0.f; }
    if (valid) {
        const int s = row % p.S;
        const long id = p.ids[row];
        const long tt = (p.dtype_emb && p.type_ids) ? p.type_ids[row] : 0;
        const float mean = p.stats[2 * (long)row], rstd = p.stats[2 * (long)row + 1];
        const uint64_t* sp = p.seed_dev ? p.seed_dev : reinterpret_cast<const uint64_t*>(p.gamma);
    const uint64_t sdv = *sp;
    const uint64_t seed = p.seed + (p.seed_dev ? sdv : 0ull);
    const float inv_keep = p.p_post > 0.f ? 1.0f / (1.0f - p.p_post) : 1.0f;
        float dy[NV][4], xh[NV][4], ga[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int col = 4 * lane + 256 * j;
            const long o = (long)row * N + col;
            const float4 a = *reinterpret_cast<const float4*>(p.dout + o);
            const float4 yy = *reinterpret_cast<const float4*>(p.y + o);
            const float4 gg = *reinterpret_cast<const float4*>(p.gamma + col);
            dy[j][0] = a.x; dy[j][1] = a.y; dy[j][2] = a.z; dy[j][3] = a.w;
            ga[j][0] = gg.x; ga[j][1] = gg.y; ga[j][2] = gg.z; ga[j][3] = gg.w;
            xh[j][0] = (yy.x - mean) * rstd; xh[j][1] = (yy.y - mean) * rstd;
            xh[j][2] = (yy.z - mean) * rstd; xh[j][3] = (yy.w - mean) * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.p_post > 0.f) dy[j][e] *= dropout_scale(seed, p.off_post, (uint64_t)(o + e), p.p_post, inv_keep);
                const float gq = dy[j][e] * ga[j][e];
                s1 += gq; s2 += gq * xh[j][e];
                dgm[j][e] = dy[j][e] * xh[j][e];
                dbt[j][e] = dy[j][e];
            }
        }
        s1 = wave_sum(s1) * (1.0f / N);
        s2 = wave_sum(s2) * (1.0f / N);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int col = 4 * lane + 256 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx = rstd * (dy[j][e] * ga[j][e] - s1 - xh[j][e] * s2);
                if (p.drows) p.drows[(long)row * N + col + e] = dx;
                else unsafeAtomicAdd(p.dword + id * N + col + e, dx);
                if (!DET && p.dpos) unsafeAtomicAdd(p.dpos + (long)s * N + col + e, dx);
                // token types 0 / 1 (every row of a batch hits the same one or two table rows) are combined per
                // block below; anything else goes straight to the table
                if (p.dtype_emb && tt > 1) unsafeAtomicAdd(p.dtype_emb + tt * N + col + e, dx);
                dxs[j][e] = dx;
            }
        }
        row_type = (int)tt;
    }
    if constexpr (DET) {
        __shared__ int last_flag;
        const long nb = gridDim.x;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave][4 * lane + 256 * j + e] = t == 0 ? dgm[j][e] : dbt[j][e];
            __syncthreads();
            float* slot = det_part + ((long)t * nb + blockIdx.x) * N;
            for (int c = threadIdx.x; c < N; c += 256) slot[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        }
        if (det_last_block(det_counter, (int)nb, &last_flag)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* dst = t == 0 ? p.dgamma : p.dbeta;
                for (int c = threadIdx.x; c < N; c += 256) {
                    const float* q = det_part + (long)t * nb * N + c;
                    float acc = 0.f;
                    for (long b = 0; b < nb; ++b) acc += q[b * N];
                    dst[c] += acc;
                }
            }
        }
        return;
    }
    // dgamma / dbeta / token-type rows 0 and 1: combine the 4 rows of the block, one atomic per column per block
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= 2 && !p.dtype_emb) break;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                red[wave][4 * lane + 256 * j + e] = t == 0 ? dgm[j][e] : t == 1 ? dbt[j][e] : (row_type == t - 2 ? dxs[j][e] : 0.f);
        __syncthreads();
        float* dst = t == 0 ? p.dgamma : t == 1 ? p.dbeta : p.dtype_emb + (long)(t - 2) * N;
        for (int c = threadIdx.x; c < N; c += 256) {
            const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            if (t < 2 || v != 0.0f) unsafeAtomicAdd(dst + c, v);
        }
    }
}

}  // namespace

extern "C" int univl_embed_text_fwd(const UnivlEmbedText* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "univl_embed_text_fwd: null descriptor");
    UNIVL_CHECK_ARG(d->N == 768, UNIVL_EUNSUPPORTED, "univl_embed_text_fwd: N=%d (768 supported)", d->N);
    UNIVL_CHECK_ARG(d->B > 0 && d->S > 0 && d->ids && d->word && d->pos && d->gamma && d->beta && (d->out32 || d->out16),
                    UNIVL_EINVAL, "univl_embed_text_fwd: null / empty argument");
    dim3 grid((d->B * d->S + 3) / 4), block(256);
    if (d->dtype == UNIVL_DT_BF16) hipLaunchKernelGGL((embed_fwd_kernel<__bf16>), grid, block, 0, stream, *d);
    else hipLaunchKernelGGL((embed_fwd_kernel<float>), grid, block, 0, stream, *d);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

namespace {
// Deterministic scatter: token t owns destination row ids[t] iff no earlier token has the same id; the owner adds the rows of ALL
// tokens with that id in token order, then adds the total to the table (one writer per table row and launch).
__global__ __launch_bounds__(256) void embed_scatter_det_kernel(const int64_t* ids, const float* rows, long n, float scale, float* dword) {
    const long t = blockIdx.x;
    const int64_t id = ids[t];
    int dup = 0;
    for (long j = threadIdx.x; j < t; j += 256) dup |= (ids[j] == id) ? 1 : 0;
    if (__syncthreads_or(dup)) return;
    float acc[NV] = {0.f, 0.f, 0.f};
    for (long u = t; u < n; ++u) {
        if (ids[u] != id) continue;                       // block-uniform
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] += rows[u * N + threadIdx.x + 256 * j] * scale;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) dword[id * N + threadIdx.x + 256 * j] += acc[j];
}

// dpos[s, c] += sum_b rows[b * S + s, c]   (b ascending)
__global__ __launch_bounds__(256) void embed_dpos_gather_kernel(const float* rows, int B, int S, float* dpos) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)S * N) return;
    const int s = (int)(i / N), c = (int)(i % N);
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += rows[((long)b * S + s) * N + c];
    dpos[i] += acc;
}

// token-type rows 0 and 1: dtype[k, c] += sum over the tokens of type k of rows[t, c]   (t ascending); blockIdx.y = k
__global__ __launch_bounds__(256) void embed_dtype_gather_kernel(const int64_t* type_ids, const float* rows, long n, float* dtype_emb) {
    const int c = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    float acc = 0.f;
    for (long t = 0; t < n; ++t) {
        const int64_t tt = type_ids ? type_ids[t] : 0;    // uniform
        if (tt == k) acc += rows[t * N + c];
    }
    if (acc != 0.0f) dtype_emb[(long)k * N + c] += acc;
}

__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* ids, const float* rows, float scale, float* dword) {
    const long t = blockIdx.x;
    const long id = ids[t];
    for (int c = threadIdx.x; c < N; c += 256) unsafeAtomicAdd(dword + id * N + c, rows[t * N + c] * scale);
}
}  // namespace

extern "C" int univl_embed_scatter(const int64_t* ids, const float* rows, int64_t n, float scale, float* dword, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(ids && rows && dword && n > 0, UNIVL_EINVAL, "univl_embed_scatter: bad argument");
    if (univl_deterministic()) hipLaunchKernelGGL(embed_scatter_det_kernel, dim3((unsigned)n), dim3(256), 0, stream, ids, rows, (long)n, scale, dword);
    else hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)n), dim3(256), 0, stream, ids, rows, scale, dword);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// ---------------------------------------------------------------------------------- sparse bookkeeping of the word table
// In the retrieval configurations the token gather is the ONLY source of the 30522 x 768 word-embedding gradient: at most
// B*W of its rows are non-zero.  Instead of clearing 94 MB before every backward and streaming 94 MB again for its norm,
// the rows written by the previous backward(s) are kept in a device list:  list[0 .. meta[0])  (meta[1] != 0: the list
// overflowed or somebody wrote the table densely -> every row is treated as listed).
namespace {
constexpr int ROWS_GRID = 8192;

__global__ __launch_bounds__(256) void rows_zero_kernel(float* table, long rows_total, const int64_t* list, const int* meta) {
    const int n = meta[0], over = meta[1];
    if (!over && (int)blockIdx.x >= n) return;
    for (long r = over ? blockIdx.x : list[blockIdx.x]; r < rows_total; r += ROWS_GRID) {
        float4* p = reinterpret_cast<float4*>(table + r * N);
        for (int c = threadIdx.x; c < N / 4; c += 256) p[c] = float4{0.f, 0.f, 0.f, 0.f};
        if (!over) break;
    }
}

__global__ __launch_bounds__(256) void rows_append_kernel(const int64_t* ids, int n, int64_t* list, int cap, int* meta, int reset,
                                                          uint8_t* ever, long rows_total) {
    __shared__ int base, over;
    if (threadIdx.x == 0) {
        const int cur = reset ? 0 : meta[0];
        int ov = reset ? 0 : meta[1];
        if (cur + n > cap) ov = 1;
        base = cur; over = ov;
        meta[1] = ov;
        meta[0] = ov ? cur : cur + n;
    }
    __syncthreads();
    if (!over) for (int i = threadIdx.x; i < n; i += 256) list[base + i] = ids[i];
    if (ever) {          // sticky "ever written" flags (UnivlAdam.row_flags); an overflowing list means any row may have been written
        if (over) for (long r = threadIdx.x; r < rows_total; r += 256) ever[r] = 1;
        else for (int i = threadIdx.x; i < n; i += 256) ever[ids[i]] = 1;
    }
}

__global__ __launch_bounds__(256) void rows_sumsq_kernel(const float* table, long rows_total, const int64_t* list, const int* meta,
                                                         float* out, float* det_part) {
    __shared__ float red[4];
    const int n = meta[0], over = meta[1];
    if (!over && (int)blockIdx.x >= n) { if (det_part && threadIdx.x == 0) det_part[blockIdx.x] = 0.f; return; }
    float acc = 0.f;
    if (over) {
        for (long r = blockIdx.x; r < rows_total; r += ROWS_GRID) {
            const float4* p = reinterpret_cast<const float4*>(table + r * N);
            for (int c = threadIdx.x; c < N / 4; c += 256) { const float4 v = p[c]; acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
        }
    } else {
        const int64_t r = list[blockIdx.x];
        int dup = 0;                                   // a row listed twice is counted by its first occurrence only
        for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) dup |= (list[j] == r) ? 1 : 0;
        if (__syncthreads_or(dup)) { if (det_part && threadIdx.x == 0) det_part[blockIdx.x] = 0.f; return; }
        const float4* p = reinterpret_cast<const float4*>(table + r * N);
        for (int c = threadIdx.x; c < N / 4; c += 256) { const float4 v = p[c]; acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (red[0] + red[1]) + (red[2] + red[3]);
        if (det_part) det_part[blockIdx.x] = s;          // deterministic mode: rows_sumsq_finish_kernel adds the slots in block order
        else unsafeAtomicAdd(out, s);
    }
}

__global__ __launch_bounds__(256) void rows_sumsq_finish_kernel(const float* part, int n, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}
}  // namespace

extern "C" int univl_rows_zero(float* table, int64_t rows_total, const int64_t* list, const int32_t* meta, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(table && list && meta && rows_total > 0 && aligned16(table), UNIVL_EINVAL, "univl_rows_zero: bad argument");
    hipLaunchKernelGGL(rows_zero_kernel, dim3(ROWS_GRID), dim3(256), 0, stream, table, (long)rows_total, list, meta);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_rows_append(const int64_t* ids, int32_t n, int64_t* list, int32_t cap, int32_t* meta, int32_t reset,
                                 uint8_t* ever, int64_t rows_total, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(ids && list && meta && n > 0 && cap > 0 && cap <= ROWS_GRID && (!ever || rows_total > 0), UNIVL_EINVAL,
                    "univl_rows_append: bad argument (cap <= %d)", ROWS_GRID);
    hipLaunchKernelGGL(rows_append_kernel, dim3(1), dim3(256), 0, stream, ids, n, list, cap, meta, reset, ever, (long)rows_total);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_rows_sumsq(const float* table, int64_t rows_total, const int64_t* list, const int32_t* meta, float* out,
                                hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(table && list && meta && out && rows_total > 0 && aligned16(table), UNIVL_EINVAL, "univl_rows_sumsq: bad argument");
    float* part = nullptr;
    if (univl_deterministic()) {
        part = static_cast<float*>(univl_det_alloc(ROWS_GRID * sizeof(float)));
        if (!part) return UNIVL_EINVAL;
    }
    hipLaunchKernelGGL(rows_sumsq_kernel, dim3(ROWS_GRID), dim3(256), 0, stream, table, (long)rows_total, list, meta, out, part);
    if (part) hipLaunchKernelGGL(rows_sumsq_finish_kernel, dim3(1), dim3(256), 0, stream, part, ROWS_GRID, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_embed_text_bwd(const UnivlEmbedText* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(d != nullptr, UNIVL_EINVAL, "univl_embed_text_bwd: null descriptor");
    UNIVL_CHECK_ARG(d->N == 768, UNIVL_EUNSUPPORTED, "univl_embed_text_bwd: N=%d (768 supported)", d->N);
    // dpos may be null when drows is given: the caller builds the position-table gradient from the rows itself (univl_rows_gather_sum)
    UNIVL_CHECK_ARG(d->B > 0 && d->S > 0 && d->ids && d->dout && d->y && d->stats && d->gamma && d->dword && (d->dpos || d->drows) &&
                        d->dgamma && d->dbeta,
                    UNIVL_EINVAL, "univl_embed_text_bwd: null / empty argument");
    dim3 grid((d->B * d->S + 3) / 4), block(256);
    if (univl_deterministic()) {
        UnivlEmbedText q = *d;
        const long T = (long)d->B * d->S;
        float* part = static_cast<float*>(univl_det_alloc((size_t)2 * grid.x * N * sizeof(float)));
        int* counter = univl_det_counter();
        if (!part || !counter) return UNIVL_EINVAL;
        if (!q.drows) {
            q.drows = static_cast<float*>(univl_det_alloc((size_t)T * N * sizeof(float)));
            if (!q.drows) return UNIVL_EINVAL;
        }
        hipLaunchKernelGGL(embed_bwd_kernel<true>, grid, block, 0, stream, q, part, counter);
        if (!d->drows)       // the caller wants the table gradient itself, not the per-token rows
            hipLaunchKernelGGL(embed_scatter_det_kernel, dim3((unsigned)T), block, 0, stream, d->ids, q.drows, T, 1.0f, d->dword);
        if (d->dpos)
            hipLaunchKernelGGL(embed_dpos_gather_kernel, dim3((unsigned)(((long)d->S * N + 255) / 256)), block, 0, stream, q.drows, d->B, d->S, d->dpos);
        if (d->dtype_emb)
            hipLaunchKernelGGL(embed_dtype_gather_kernel, dim3(N / 256, 2), block, 0, stream, d->type_ids, q.drows, T, d->dtype_emb);
        UNIVL_LAUNCH_CHECK();
        return UNIVL_OK;
    }
    hipLaunchKernelGGL(embed_bwd_kernel<false>, grid, block, 0, stream, *d);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
