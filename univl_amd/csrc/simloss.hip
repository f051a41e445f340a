// Pooling for the video-text similarity and the three similarity losses of the reference.
//   univl_pool_*          _mean_pooling_for_similarity + F.normalize        modules/modeling.py:327-339, 385-388
//   univl_maxmargin_loss  MaxMarginRankingLoss.forward                        modules/until_module.py:245-251
//   univl_crossen_loss    CrossEn.forward                                     modules/until_module.py:186-191
//   univl_milnce_loss     MILNCELoss.forward                                  modules/until_module.py:201-221
// The (B,768)x(768,B) similarity product itself goes through univl_gemm.  All of this is bandwidth-trivial
// (a few hundred KB); the kernels exist so that the whole step stays on the stream with no host round trip.
#include "common.h"
#include "univl_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

constexpr int PN = 768, PC = PN / 256;

__device__ __forceinline__ float pool_weight(const UnivlPool& p, int b, int s) {
    float w = p.mask ? (float)p.mask[(long)b * p.S + s] : 1.0f;
    if (p.skip_first && s == 0) w = 0.0f;          // attention_mask_un[:, 0, :] = 0   (modeling.py:329)
    return w;
}

// One workgroup per pooled row.  The S mask weights are staged in LDS once (a serial "read mask, maybe read row" loop
// is S dependent L2 round trips); the row loop then issues independent, unconditional loads, 8 sequence positions in
// flight per thread.  Masked positions contribute exactly 0 (select, not multiply: padded rows may hold anything).
constexpr int PCHUNK = 256;

__device__ __forceinline__ float stage_weights(const UnivlPool& p, int b, int s0, float* wts, float* red) {
    const int s = s0 + threadIdx.x;
    const float w = (s < p.S) ? pool_weight(p, b, s) : 0.0f;
    wts[threadIdx.x] = w;
    return block_sum(w, red);          // includes the barrier that publishes wts
}

__device__ __forceinline__ void pool_fwd_body(const UnivlPool& p, const int b, float* red, float* wts) {
    const int t = threadIdx.x;
    float acc[PC] = {0.f, 0.f, 0.f};
    float cnt = 0.f;
    for (int s0 = 0; s0 < p.S; s0 += PCHUNK) {
        cnt += stage_weights(p, b, s0, wts, red);
        const int n = min(PCHUNK, p.S - s0);
        const float* x = p.x + ((long)b * p.S + s0) * p.ldx_row + t;
#pragma unroll 8
        for (int s = 0; s < n; ++s) {
            const float w = wts[s];
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                const float v = x[(long)s * p.ldx_row + 256 * j];
                acc[j] += (w != 0.0f) ? v * w : 0.0f;
            }
        }
        __syncthreads();
    }
    if (!p.skip_first && cnt == 0.0f) cnt = 1.0f;   // video_mask_un_sum[== 0] = 1   (modeling.py:336)
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j) { acc[j] /= cnt; sq += acc[j] * acc[j]; }
    float inv = 1.0f;
    if (p.normalize) inv = 1.0f / fmaxf(sqrtf(block_sum(sq, red)), 1e-12f);   // F.normalize eps
#pragma unroll
    for (int j = 0; j < PC; ++j) {
        if (p.mean) p.mean[(long)b * PN + t + 256 * j] = acc[j];
        p.out[(long)b * PN + t + 256 * j] = acc[j] * inv;
    }
}

__global__ __launch_bounds__(256) void pool_fwd_kernel(UnivlPool p) {
    __shared__ float red[4];
    __shared__ float wts[PCHUNK];
    pool_fwd_body(p, blockIdx.x, red, wts);
}

// two descriptors in one launch: blocks [0, a.B) pool a's rows, the rest b's (block-uniform choice)
__global__ __launch_bounds__(256) void pool_pair_fwd_kernel(UnivlPool a, UnivlPool b) {
    __shared__ float red[4];
    __shared__ float wts[PCHUNK];
    if ((int)blockIdx.x < a.B) pool_fwd_body(a, blockIdx.x, red, wts);
    else pool_fwd_body(b, (int)blockIdx.x - a.B, red, wts);
}

// grid (B, ceil(S / PCHUNK)): every block recomputes the row's count (cheap, from LDS-staged weights) and writes its
// own chunk of sequence positions
__device__ __forceinline__ void pool_bwd_body(const UnivlPool& p, const int b, const int chunk, float* red, float* wts, float* mine) {
    const int t = threadIdx.x;
    if (chunk * PCHUNK >= p.S) return;              // (pair launch: the grid is sized for the longer sequence) -- block-uniform
    float cnt = 0.f;
    for (int s0 = 0; s0 < p.S; s0 += PCHUNK) {
        cnt += stage_weights(p, b, s0, wts, red);
        if (s0 == chunk * PCHUNK) mine[t] = wts[t];
        __syncthreads();
    }
    if (!p.skip_first && cnt == 0.0f) cnt = 1.0f;
    float m[PC], d[PC];
    float sq = 0.f, dot = 0.f;
#pragma unroll
    for (int j = 0; j < PC; ++j) {
        m[j] = p.mean[(long)b * PN + t + 256 * j];
        d[j] = p.dsim ? 0.0f : p.dout[(long)b * PN + t + 256 * j];
    }
    if (p.dsim) {
        // upstream gradient straight from d loss / d sim (UnivlPool.dsim): row b (or column b) of dsim times the other side's matrix
        const float gs = p.gscale ? p.gscale[0] : 1.0f;
        for (int k = 0; k < p.n_other; ++k) {
            const float w = p.transpose ? p.dsim[(long)k * p.ldsim + b] : p.dsim[(long)b * p.ldsim + k];
#pragma unroll
            for (int j = 0; j < PC; ++j) d[j] += w * p.other[(long)k * PN + t + 256 * j];
        }
#pragma unroll
        for (int j = 0; j < PC; ++j) d[j] *= gs;
    }
#pragma unroll
    for (int j = 0; j < PC; ++j) {
        sq += m[j] * m[j];
        dot += m[j] * d[j];
    }
    if (p.normalize) {
        const float nrm = fmaxf(sqrtf(block_sum(sq, red)), 1e-12f);
        const float dd = block_sum(dot, red) / (nrm * nrm);     // sum(dout * out) / nrm
#pragma unroll
        for (int j = 0; j < PC; ++j) d[j] = (d[j] - m[j] * dd) / nrm;
    }
    const float inv_cnt = 1.0f / cnt;
    const int s0 = chunk * PCHUNK, n = min(PCHUNK, p.S - s0);
    float* dx = p.dx + ((long)b * p.S + s0) * PN + t;
    if (p.accumulate) {
        // read-modify-write of dx: the old values of 8 sequence positions are fetched before the first store (the loads
        // of position s+1 may not be hoisted above the stores of position s by the compiler: same array)
        constexpr int UB = 8;
        for (int s0b = 0; s0b < n; s0b += UB) {
            float old[UB][PC];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int s = min(s0b + u, n - 1);
#pragma unroll
                for (int j = 0; j < PC; ++j) old[u][j] = dx[(long)s * PN + 256 * j];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int s = s0b + u;
                if (s < n) {
                    const float w = mine[s] * inv_cnt;
#pragma unroll
                    for (int j = 0; j < PC; ++j) dx[(long)s * PN + 256 * j] = old[u][j] + d[j] * w;
                }
            }
        }
    } else {
#pragma unroll 4
        for (int s = 0; s < n; ++s) {
            const float w = mine[s] * inv_cnt;
#pragma unroll
            for (int j = 0; j < PC; ++j) dx[(long)s * PN + 256 * j] = d[j] * w;
        }
    }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(UnivlPool p) {
    __shared__ float red[4];
    __shared__ float wts[PCHUNK];
    __shared__ float mine[PCHUNK];
    pool_bwd_body(p, blockIdx.x, blockIdx.y, red, wts, mine);
}

__global__ __launch_bounds__(256) void pool_pair_bwd_kernel(UnivlPool a, UnivlPool b) {
    __shared__ float red[4];
    __shared__ float wts[PCHUNK];
    __shared__ float mine[PCHUNK];
    if ((int)blockIdx.x < a.B) pool_bwd_body(a, blockIdx.x, blockIdx.y, red, wts, mine);
    else pool_bwd_body(b, (int)blockIdx.x - a.B, blockIdx.y, red, wts, mine);
}

// ----------------------------------------------------------------------------------------------- losses
__global__ __launch_bounds__(256) void maxmargin_kernel(const float* x, int n, int ld, float margin, const float* w,
                                                        float* loss, float* dx) {
    extern __shared__ float diag_acc[];   // [n] gradient accumulated on the diagonal
    __shared__ float red[4];
    const float inv = 1.0f / ((float)n * (float)n);
    for (int i = threadIdx.x; i < n; i += 256) diag_acc[i] = 0.f;
    __syncthreads();
    float acc = 0.f;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        const float v = x[(long)i * ld + j], wi = w ? w[e] : 1.0f;
        const float a = margin + v - x[(long)i * ld + i];     // relu(margin + x - d.view(-1,1))
        const float b = margin + v - x[(long)j * ld + j];     // relu(margin + x - d.view(1,-1))
        float g = 0.f;
        if (a > 0.f) { acc += wi * a; g += wi; atomicAdd(&diag_acc[i], -wi); }
        if (b > 0.f) { acc += wi * b; g += wi; atomicAdd(&diag_acc[j], -wi); }
        dx[(long)i * ld + j] = g * inv;
    }
    const float total = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = total * inv;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) dx[(long)i * ld + i] += diag_acc[i] * inv;
}

__global__ __launch_bounds__(256) void crossen_kernel(const float* x, int n, int ld, float* loss, float* dx) {
    __shared__ float red[4];
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
        float mx = -INFINITY;
        for (int j = threadIdx.x; j < n; j += 256) mx = fmaxf(mx, x[(long)i * ld + j]);
        mx = block_max(mx, red);
        float s = 0.f;
        for (int j = threadIdx.x; j < n; j += 256) s += expf(x[(long)i * ld + j] - mx);
        s = block_sum(s, red);
        const float lse = mx + logf(s);
        total += lse - x[(long)i * ld + i];
        for (int j = threadIdx.x; j < n; j += 256)
            dx[(long)i * ld + j] = (expf(x[(long)i * ld + j] - lse) - (i == j ? 1.0f : 0.0f)) / (float)n;
    }
    if (threadIdx.x == 0) loss[0] = total / (float)n;
}

// MIL-NCE.  Row i of the reference's [n, 2n] matrix is  [ sim[:, i]  |  sim[i, :] with the own block masked ].
// loss_i = logsumexp(row i) - logsumexp_{j in block(i)} sim[j, i]; only rows b*n_pair + n_pair/2 are selected.
__global__ __launch_bounds__(256) void milnce_kernel(const float* x, int bs, int np, int ld, float* loss, float* dx) {
    __shared__ float red[4];
    const int n = bs * np;
    for (int e = threadIdx.x; e < n * n; e += 256) dx[(long)(e / n) * ld + (e % n)] = 0.f;
    __syncthreads();
    float total = 0.f;
    for (int bb = 0; bb < bs; ++bb) {
        const int i = bb * np + np / 2;
        const int lo = bb * np, hi = lo + np;
        float mx = -INFINITY;
        for (int j = threadIdx.x; j < 2 * n; j += 256) {
            float v;
            if (j < n) v = x[(long)j * ld + i];
            else { const int jj = j - n; v = (jj >= lo && jj < hi) ? -INFINITY : x[(long)i * ld + jj]; }
            mx = fmaxf(mx, v);
        }
        mx = block_max(mx, red);
        float sden = 0.f, snum = 0.f;
        for (int j = threadIdx.x; j < 2 * n; j += 256) {
            if (j < n) {
                const float e = expf(x[(long)j * ld + i] - mx);
                sden += e;
                if (j >= lo && j < hi) snum += e;
            } else {
                const int jj = j - n;
                if (!(jj >= lo && jj < hi)) sden += expf(x[(long)i * ld + jj] - mx);
            }
        }
        sden = block_sum(sden, red);
        snum = block_sum(snum, red);
        total += logf(sden) - logf(snum);
        const float wgt = 1.0f / (float)bs;
        __syncthreads();
        for (int j = threadIdx.x; j < 2 * n; j += 256) {
            if (j < n) {
                const float e = expf(x[(long)j * ld + i] - mx);
                float g = e / sden;
                if (j >= lo && j < hi) g -= e / snum;
                dx[(long)j * ld + i] += wgt * g;
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) {
            if (!(j >= lo && j < hi)) dx[(long)i * ld + j] += wgt * expf(x[(long)i * ld + j] - mx) / sden;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = total / (float)bs;
}

__global__ void scale_dev_kernel(float* x, long n, const float* s) {
    const float k = s[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= k;
}

}  // namespace

extern "C" int univl_scale_by_device_scalar(float* x, int64_t n, const float* s, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(x && s && n > 0, UNIVL_EINVAL, "univl_scale_by_device_scalar: bad argument");
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(scale_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, (long)n, s);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

__global__ __launch_bounds__(256) void rank_counts_kernel(const float* x, int n, long ld, int* gt, int* eq) {
    __shared__ int red[8];
    const int i = blockIdx.x, t = threadIdx.x;
    const float* row = x + (long)i * ld;
    const float d = row[i];
    int g = 0, e = 0;
    for (int j = t; j < n; j += 256) {
        const float v = row[j];
        g += v > d ? 1 : 0;
        e += v == d ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_down(g, o, 64); e += __shfl_down(e, o, 64); }
    if ((t & 63) == 0) { red[t >> 6] = g; red[4 + (t >> 6)] = e; }
    __syncthreads();
    if (t == 0) {
        gt[i] = (red[0] + red[1]) + (red[2] + red[3]);
        eq[i] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

extern "C" int univl_rank_counts(const float* sim, int32_t n, int64_t ld, int32_t* gt, int32_t* eq, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(sim && gt && eq && n > 0 && ld >= n, UNIVL_EINVAL, "univl_rank_counts: bad argument");
    hipLaunchKernelGGL(rank_counts_kernel, dim3(n), dim3(256), 0, stream, sim, n, (long)ld, gt, eq);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_pool_fwd(const UnivlPool* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(d && d->N == 768 && d->B > 0 && d->S > 0 && d->x && d->out, UNIVL_EINVAL, "univl_pool_fwd: bad argument (N must be 768)");
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(d->B), dim3(256), 0, stream, *d);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

static int pool_bwd_check(const UnivlPool* d, const char* who) {
    UNIVL_CHECK_ARG(d && d->N == 768 && d->B > 0 && d->S > 0 && d->mean && d->dx && (d->dout || (d->dsim && d->other && d->n_other > 0)),
                    UNIVL_EINVAL, "%s: bad argument", who);
    return UNIVL_OK;
}

extern "C" int univl_pool_bwd(const UnivlPool* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    const int rc = pool_bwd_check(d, "univl_pool_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(d->B, (d->S + PCHUNK - 1) / PCHUNK), dim3(256), 0, stream, *d);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_pool_pair_fwd(const UnivlPool* a, const UnivlPool* b, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(a && b && a->N == 768 && b->N == 768 && a->B > 0 && b->B > 0 && a->S > 0 && b->S > 0 && a->x && b->x && a->out && b->out,
                    UNIVL_EINVAL, "univl_pool_pair_fwd: bad argument (N must be 768)");
    hipLaunchKernelGGL(pool_pair_fwd_kernel, dim3(a->B + b->B), dim3(256), 0, stream, *a, *b);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_pool_pair_bwd(const UnivlPool* a, const UnivlPool* b, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    int rc = pool_bwd_check(a, "univl_pool_pair_bwd");
    if (rc) return rc;
    rc = pool_bwd_check(b, "univl_pool_pair_bwd");
    if (rc) return rc;
    const int smax = a->S > b->S ? a->S : b->S;
    hipLaunchKernelGGL(pool_pair_bwd_kernel, dim3(a->B + b->B, (smax + PCHUNK - 1) / PCHUNK), dim3(256), 0, stream, *a, *b);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_maxmargin_loss(const float* sim, int32_t n, int32_t ld, float margin, const float* weight, float* loss,
                                    float* dsim, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(sim && loss && dsim && n > 0 && n <= 8192 && ld >= n, UNIVL_EINVAL, "univl_maxmargin_loss: bad argument");
    hipLaunchKernelGGL(maxmargin_kernel, dim3(1), dim3(256), n * sizeof(float), stream, sim, n, ld, margin, weight, loss, dsim);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_crossen_loss(const float* sim, int32_t n, int32_t ld, float* loss, float* dsim, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(sim && loss && dsim && n > 0 && ld >= n, UNIVL_EINVAL, "univl_crossen_loss: bad argument");
    hipLaunchKernelGGL(crossen_kernel, dim3(1), dim3(256), 0, stream, sim, n, ld, loss, dsim);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_milnce_loss(const float* sim, int32_t batch_size, int32_t n_pair, int32_t ld, float* loss, float* dsim,
                                 hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(sim && loss && dsim && batch_size > 0 && n_pair > 0 && ld >= batch_size * n_pair, UNIVL_EINVAL, "univl_milnce_loss: bad argument");
    hipLaunchKernelGGL(milnce_kernel, dim3(1), dim3(256), 0, stream, sim, batch_size, n_pair, ld, loss, dsim);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
