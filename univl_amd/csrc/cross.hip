// Small HBM-bound kernels around the cross encoder, the caption decoder's classifier and the pretraining heads.
//
//   univl_pair_concat_fwd/bwd   torch.cat((sequence_output, visual_output), 1) for a list of (text row, video row)
//                               pairs + concat_mask -- UniVL._get_cross_output modules/modeling.py:315-325 and the
//                               repeat/view pair expansion of _cross_similarity :341-375.  Backward scatter-adds.
//   univl_postype_fwd/bwd       position + token-type rows of CrossEmbeddings (module_cross.py:123-138) folded into one
//                               [S,768] table that the LayerNorm kernel adds with period S.
//   univl_tanh_fwd/bwd          CrossPooler's nn.Tanh (module_cross.py:281-287).
//   univl_ce_loss               CrossEntropyLoss(ignore_index=-1) over the vocabulary (modeling.py:168,252-254,275) and
//                               its gradient (written in the compute type as the operand of the classifier's dgrad/wgrad).
//   univl_mfm_nce_loss          the masked-frame NCE of UniVL._calculate_mfm_loss (modeling.py:278-297), on the logits
//                               matrix produced by univl_gemm.
#include "common.h"
#include "univl_hip.h"

namespace {

constexpr int N = 768;

// one wave per output row (pair p, position s); S = W + F
__global__ __launch_bounds__(256) void pair_concat_fwd_kernel(const float* seq, const float* vis, const int64_t* amask,
                                                              const int64_t* vmask, const int32_t* tidx, const int32_t* vidx,
                                                              int P, int W, int F, float* out, int64_t* omask) {
    const int S = W + F;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)P * S) return;
    const int p = (int)(row / S), s = (int)(row % S);
    const float* src;
    int64_t m;
    if (s < W) { const int b = tidx[p]; src = seq + ((long)b * W + s) * N; m = amask[(long)b * W + s]; }
    else       { const int b = vidx[p]; src = vis + ((long)b * F + (s - W)) * N; m = vmask[(long)b * F + (s - W)]; }
    float4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const float4*>(src + 4 * lane + 256 * j);
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(out + row * N + 4 * lane + 256 * j) = v[j];
    if (lane == 0 && omask) omask[row] = m;
}

__global__ __launch_bounds__(256) void pair_concat_bwd_kernel(const float* dout, const int32_t* tidx, const int32_t* vidx,
                                                              int P, int W, int F, float* dseq, float* dvis) {
    const int S = W + F;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)P * S) return;
    const int p = (int)(row / S), s = (int)(row % S);
    float* dst = (s < W) ? dseq + ((long)tidx[p] * W + s) * N : dvis + ((long)vidx[p] * F + (s - W)) * N;
    float4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const float4*>(dout + row * N + 4 * lane + 256 * j);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float* d = dst + 4 * lane + 256 * j;
        unsafeAtomicAdd(d, v[j].x); unsafeAtomicAdd(d + 1, v[j].y); unsafeAtomicAdd(d + 2, v[j].z); unsafeAtomicAdd(d + 3, v[j].w);
    }
}

// Deterministic form (common.h): output row (p, s) owns its destination row iff no earlier pair p' < p points at the same source
// row; the owner adds the rows (p'', s) of ALL pairs with that source row in pair order, then adds the total to the destination.
__global__ __launch_bounds__(256) void pair_concat_bwd_det_kernel(const float* dout, const int32_t* tidx, const int32_t* vidx,
                                                                  int P, int W, int F, float* dseq, float* dvis) {
    const int S = W + F;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)P * S) return;
    const int p = (int)(row / S), s = (int)(row % S);
    const int32_t* idx = (s < W) ? tidx : vidx;
    const int mine = idx[p];
    int dup = 0;
    for (int j = lane; j < p; j += 64) dup |= (idx[j] == mine) ? 1 : 0;
    if (__any(dup)) return;                                    // wave-uniform
    float4 acc[3] = {float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f}};
    for (int u = p; u < P; ++u) {
        if (idx[u] != mine) continue;
        const float* src = dout + ((long)u * S + s) * N;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * lane + 256 * j);
            acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
        }
    }
    float* dst = (s < W) ? dseq + ((long)mine * W + s) * N : dvis + ((long)mine * F + (s - W)) * N;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float4* d = reinterpret_cast<float4*>(dst + 4 * lane + 256 * j);
        float4 o = *d;
        o.x += acc[j].x; o.y += acc[j].y; o.z += acc[j].z; o.w += acc[j].w;
        *d = o;
    }
}

__global__ __launch_bounds__(256) void postype_fwd_kernel(const float* pos, const float* type, int W, int S, float* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)S * N) return;
    const int s = (int)(i / N), c = (int)(i % N);
    out[i] = pos[i] + type[(s >= W ? N : 0) + c];
}

// dpos[s] += dpt[s] (one writer per element); dtype[0] += sum_{s < W} dpt[s], dtype[1] += sum_{s >= W} dpt[s]: the threads of
// positions 0 and W walk their half in position order (a fixed order: no atomics, the same bits every run)
__global__ __launch_bounds__(256) void postype_bwd_kernel(const float* dpt, int W, int S, float* dpos, float* dtype) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)S * N) return;
    const int s = (int)(i / N), c = (int)(i % N);
    dpos[i] += dpt[i];
    if (s == 0 || s == W) {
        const int s1 = (s == 0 && W > 0) ? W : S;            // W == 0: position 0 is already the second half
        // eight independent partial sums (positions u, u + 1, ... u + 7 of every group of eight), folded in a fixed order: the
        // loads of a group are in flight together instead of one dependent round trip per position
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int u = s;
        for (; u + 8 <= s1; u += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += dpt[(long)(u + k) * N + c];
        }
        for (int k = 0; u < s1; ++u, ++k) a[k] += dpt[(long)u * N + c];
        dtype[((s >= W) ? N : 0) + c] += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}

__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* x, float* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = tanhf(x[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* dy, const float* y, T* dx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dx[i] = from_f32<T>(dy[i] * (1.0f - y[i] * y[i]));
}

// dU = dG * gelu'(U)  (BertPredictionHeadTransform backward: the GELU sits between a GEMM and a LayerNorm there,
// module_bert.py:299-311, so it cannot ride on a GEMM epilogue)
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* dg, const T* u, T* du, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        du[i] = from_f32<T>(dg[i] * gelu_grad_f(to_f32<T>(u[i])));
}

// out[c] += sum_r x[r, c]   (bias gradient of a projection whose weight gradient is produced transposed)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long ld, int rows, int n, float* out, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float acc = 0.f;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int r = r0; r < r1; ++r) acc += to_f32<T>(x[(long)r * ld + c]);
    if (gridDim.y == 1) out[c] += acc;                       // deterministic mode: one block walks every row in order
    else unsafeAtomicAdd(out + c, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_ct_kernel(T* x, long n, const float* s) {
    const float k = s[0];
    if (k == 1.0f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = from_f32<T>(to_f32<T>(x[i]) * k);
}

__device__ __forceinline__ float block_sum_fwd(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// similarity_dense = nn.Linear(768, 1) (modeling.py:167,371): one wave per row
__global__ __launch_bounds__(256) void simdense_fwd_kernel(const float* x, const float* w, const float* b, int rows, float* out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(x + (long)row * N + 4 * lane + 256 * j);
        const float4 c = *reinterpret_cast<const float4*>(w + 4 * lane + 256 * j);
        acc += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc + b[0];
}

// dx[r,:] = ds[r] * w ; dw += sum_r ds[r] x[r,:] ; db += sum_r ds[r]     (single workgroup: rows is a few hundred)
__global__ __launch_bounds__(256) void simdense_bwd_kernel(const float* ds, const float* x, const float* w, int rows, float* dx,
                                                           float* dw, float* db) {
    __shared__ float red[4];
    float aw[3] = {0.f, 0.f, 0.f};
    float ab = 0.f;
    for (int r = 0; r < rows; ++r) {
        const float g = ds[r];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = threadIdx.x + 256 * j;
            dx[(long)r * N + c] = g * w[c];
            aw[j] += g * x[(long)r * N + c];
        }
    }
    for (int r = threadIdx.x; r < rows; r += 256) ab += ds[r];
    ab = block_sum_fwd(ab, red);
#pragma unroll
    for (int j = 0; j < 3; ++j) unsafeAtomicAdd(dw + threadIdx.x + 256 * j, aw[j]);
    if (threadIdx.x == 0) unsafeAtomicAdd(db, ab);
}

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

// scal[0] <- number of rows whose label != ignore ; scal[1] <- 0 (loss accumulator)
__global__ __launch_bounds__(256) void ce_count_kernel(const int64_t* labels, int rows, int ignore, float* scal) {
    __shared__ float red[4];
    float c = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) c += (labels[i] != ignore) ? 1.0f : 0.0f;
    c = block_sum(c, red);
    if (threadIdx.x == 0) { scal[0] = c; scal[1] = 0.0f; }
}

// one workgroup per row: log-softmax over V, loss += (lse - x[label]) / n_valid, d = (softmax - onehot) / n_valid
// rowloss != nullptr (deterministic mode): the row's loss term goes to rowloss[row] and ce_finish_kernel adds the slots in row
// order; otherwise it is added to scal[1] with an fp32 atomic.
template <typename T>
__global__ __launch_bounds__(256) void ce_row_kernel(const float* logits, long ld, const int64_t* labels, int V, int ignore,
                                                     float* scal, T* dl, long lddl, float* rowloss) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* x = logits + (long)row * ld;
    T* d = dl + (long)row * lddl;
    const long lab = labels[row];
    const float nvalid = scal[0];
    if (lab == ignore || nvalid == 0.0f) {
        for (int j = threadIdx.x; j < V; j += 256) d[j] = from_f32<T>(0.0f);
        if (rowloss && threadIdx.x == 0) rowloss[row] = 0.0f;
        return;
    }
    // Two passes over the 122 KB row in 16-byte vectors (the row is a multiple of 8 floats apart from its neighbours: 16-byte
    // aligned): pass 1 keeps a running (max, sum of exp) per thread -- the online form of logsumexp -- and combines them across the
    // block; pass 2 writes (softmax - onehot) / n_valid.  (Round 2 made three scalar passes: 92 us per launch at 512 x 30522.)
    const bool vec = ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)d) & (sizeof(T) == 2 ? 7 : 15)) == 0);
    const int nv = vec ? V / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float m = -INFINITY, s = 0.f;
    for (int j = threadIdx.x; j < nv; j += 256) {
        const float4 v = x4[j];
        const float mn = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        s = s * expf(m - mn) + ((expf(v.x - mn) + expf(v.y - mn)) + (expf(v.z - mn) + expf(v.w - mn)));
        m = mn;
    }
    for (int j = nv * 4 + threadIdx.x; j < V; j += 256) {
        const float mn = fmaxf(m, x[j]);
        s = s * expf(m - mn) + expf(x[j] - mn);
        m = mn;
    }
    const float mx = block_max(m, red);
    s = block_sum(m == -INFINITY ? 0.f : s * expf(m - mx), red);
    const float lse = mx + logf(s);
    const float inv = 1.0f / nvalid;
    for (int j = threadIdx.x; j < nv; j += 256) {
        const float4 v = x4[j];
        const int j0 = 4 * j;
        float r[4] = {expf(v.x - lse), expf(v.y - lse), expf(v.z - lse), expf(v.w - lse)};
        if (lab >= j0 && lab < j0 + 4) r[lab - j0] -= 1.0f;
        if (sizeof(T) == 2) {
            bf16x4_t w;
            w[0] = (__bf16)(r[0] * inv); w[1] = (__bf16)(r[1] * inv); w[2] = (__bf16)(r[2] * inv); w[3] = (__bf16)(r[3] * inv);
            *reinterpret_cast<bf16x4_t*>(d + j0) = w;
        } else {
            *reinterpret_cast<float4*>(d + j0) = make_float4(r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv);
        }
    }
    for (int j = nv * 4 + threadIdx.x; j < V; j += 256)
        d[j] = from_f32<T>((expf(x[j] - lse) - (j == lab ? 1.0f : 0.0f)) * inv);
    if (threadIdx.x == 0) {
        if (rowloss) rowloss[row] = (lse - x[lab]) * inv;
        else unsafeAtomicAdd(scal + 1, (lse - x[lab]) * inv);
    }
}

__global__ __launch_bounds__(256) void ce_finish_kernel(const float* scal, float* loss, const float* rowloss, int rows) {
    __shared__ float red[4];
    float total = scal[1];
    if (rowloss) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < rows; i += 256) acc += rowloss[i];
        total = block_sum(acc, red);
    }
    if (threadIdx.x == 0) loss[0] = scal[0] > 0.f ? total : NAN;
}

// MFM NCE (modeling.py:285-297): row i of the [n,n] logits is masked with (1 - m_i m_j) * -1e8, the loss is the mean over
// rows with label != -1 of  -(log_softmax(row)[i]).  Writes d logits (fp32, in place allowed) for an upstream grad of 1.
__global__ __launch_bounds__(256) void mfm_kernel(const float* logits, long ld, const int64_t* vmask, const int64_t* labels,
                                                  int n, float* scal, float* dl, long lddl, float* rowloss) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    const float* x = logits + (long)i * ld;
    float* d = dl + (long)i * lddl;
    const float nvalid = scal[0];
    const bool sel = labels[i] != -1;
    if (!sel || nvalid == 0.0f) {
        for (int j = threadIdx.x; j < n; j += 256) d[j] = 0.0f;
        if (rowloss && threadIdx.x == 0) rowloss[i] = 0.0f;
        return;
    }
    const float mi = (float)vmask[i];
    const float xii = x[i];            // read before the (possibly in-place) gradient overwrites the row
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < n; j += 256) mx = fmaxf(mx, x[j] + (1.0f - mi * (float)vmask[j]) * -1e8f);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) s += expf(x[j] + (1.0f - mi * (float)vmask[j]) * -1e8f - mx);
    s = block_sum(s, red);
    const float lse = mx + logf(s);
    const float inv = 1.0f / nvalid;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float v = x[j] + (1.0f - mi * (float)vmask[j]) * -1e8f;
        d[j] = (expf(v - lse) - (j == i ? 1.0f : 0.0f)) * inv;
    }
    if (threadIdx.x == 0) {
        const float vii = xii + (1.0f - mi * mi) * -1e8f;
        if (rowloss) rowloss[i] = (lse - vii) * inv;
        else unsafeAtomicAdd(scal + 1, (lse - vii) * inv);
    }
}

}  // namespace

extern "C" int univl_pair_concat_fwd(const float* seq, const float* vis, const int64_t* amask, const int64_t* vmask,
                                     const int32_t* tidx, const int32_t* vidx, int32_t P, int32_t W, int32_t F, float* out,
                                     int64_t* out_mask, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(seq && vis && amask && vmask && tidx && vidx && out && P > 0 && W > 0 && F > 0, UNIVL_EINVAL,
                    "univl_pair_concat_fwd: bad argument");
    const long rows = (long)P * (W + F);
    hipLaunchKernelGGL(pair_concat_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, seq, vis, amask, vmask,
                       tidx, vidx, P, W, F, out, out_mask);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_pair_concat_bwd(const float* dout, const int32_t* tidx, const int32_t* vidx, int32_t P, int32_t W,
                                     int32_t F, float* dseq, float* dvis, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(dout && tidx && vidx && dseq && dvis && P > 0 && W > 0 && F > 0, UNIVL_EINVAL, "univl_pair_concat_bwd: bad argument");
    const long rows = (long)P * (W + F);
    if (univl_deterministic())
        hipLaunchKernelGGL(pair_concat_bwd_det_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, dout, tidx, vidx, P, W, F,
                           dseq, dvis);
    else
        hipLaunchKernelGGL(pair_concat_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, dout, tidx, vidx, P, W, F,
                           dseq, dvis);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_postype_fwd(const float* pos, const float* type, int32_t W, int32_t S, float* out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(pos && type && out && S > 0 && W >= 0, UNIVL_EINVAL, "univl_postype_fwd: bad argument");
    hipLaunchKernelGGL(postype_fwd_kernel, dim3((unsigned)(((long)S * N + 255) / 256)), dim3(256), 0, stream, pos, type, W, S, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_postype_bwd(const float* dpt, int32_t W, int32_t S, float* dpos, float* dtype, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(dpt && dpos && dtype && S > 0, UNIVL_EINVAL, "univl_postype_bwd: bad argument");
    hipLaunchKernelGGL(postype_bwd_kernel, dim3((unsigned)(((long)S * N + 255) / 256)), dim3(256), 0, stream, dpt, W, S, dpos, dtype);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_tanh_fwd(const float* x, float* y, int64_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(x && y && n > 0, UNIVL_EINVAL, "univl_tanh_fwd: bad argument");
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, (long)n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_tanh_bwd(int32_t dtype, const float* dy, const float* y, void* dx, int64_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(dy && y && dx && n > 0, UNIVL_EINVAL, "univl_tanh_bwd: bad argument");
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    if (dtype == UNIVL_DT_BF16)
        hipLaunchKernelGGL((tanh_bwd_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, dy, y, reinterpret_cast<__bf16*>(dx), (long)n);
    else
        hipLaunchKernelGGL((tanh_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, dy, y, reinterpret_cast<float*>(dx), (long)n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_gelu_bwd(int32_t dtype, const float* dg, const void* u, void* du, int64_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(dg && u && du && n > 0, UNIVL_EINVAL, "univl_gelu_bwd: bad argument");
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    if (dtype == UNIVL_DT_BF16)
        hipLaunchKernelGGL((gelu_bwd_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, dg,
                           reinterpret_cast<const __bf16*>(u), reinterpret_cast<__bf16*>(du), (long)n);
    else
        hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, dg,
                           reinterpret_cast<const float*>(u), reinterpret_cast<float*>(du), (long)n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_colsum(int32_t dtype, const void* x, int64_t ld, int32_t rows, int32_t n, float* out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(x && out && rows > 0 && n > 0, UNIVL_EINVAL, "univl_colsum: bad argument");
    const int rpb = univl_deterministic() ? rows : 64;
    // (a 16-byte-vector form of this kernel was measured at 128 pairs and removed: 12.70 vs 12.59 ms per step,
    // profiles/r03z3_ab_colsum_b128.txt -- the column sums are not what the chain waits for)
    dim3 grid((n + 255) / 256, (rows + rpb - 1) / rpb);
    if (dtype == UNIVL_DT_BF16)
        hipLaunchKernelGGL((colsum_kernel<__bf16>), grid, dim3(256), 0, stream, reinterpret_cast<const __bf16*>(x), (long)ld, rows, n, out, rpb);
    else
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, stream, reinterpret_cast<const float*>(x), (long)ld, rows, n, out, rpb);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_scale_ct_by_device_scalar(int32_t dtype, void* x, int64_t n, const float* s, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(x && s && n > 0, UNIVL_EINVAL, "univl_scale_ct_by_device_scalar: bad argument");
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    if (dtype == UNIVL_DT_BF16)
        hipLaunchKernelGGL((scale_ct_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<__bf16*>(x), (long)n, s);
    else
        hipLaunchKernelGGL((scale_ct_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<float*>(x), (long)n, s);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* src, uint4* dst, const int32_t* idx, long stride16, long n16) {
    const int r = blockIdx.y;
    const uint4* s = src + (long)idx[r] * stride16;
    uint4* d = dst + (long)r * stride16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) d[i] = s[i];
}

__global__ __launch_bounds__(256) void log_softmax_rows_kernel(float* x, int n, long ld) {
    __shared__ float red[4];
    float* row = x + (long)blockIdx.x * ld;
    const int t = threadIdx.x;
    float mx = -INFINITY;
    for (int i = t; i < n; i += 256) mx = fmaxf(mx, row[i]);
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = t; i < n; i += 256) sum += expf(row[i] - mx);
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[t >> 6] = sum;
    __syncthreads();
    const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = t; i < n; i += 256) row[i] -= lse;
}
}  // namespace

extern "C" int univl_gather_rows(const void* src, void* dst, const int32_t* idx, int32_t rows, int64_t row_stride, int64_t copy_bytes,
                                 hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(src && dst && idx && rows > 0 && copy_bytes >= 0 && copy_bytes <= row_stride, UNIVL_EINVAL,
                    "univl_gather_rows: bad argument");
    UNIVL_CHECK_ARG(aligned16(src) && aligned16(dst) && row_stride % 16 == 0 && copy_bytes % 16 == 0, UNIVL_EALIGN,
                    "univl_gather_rows: pointers, row stride and copy size must be multiples of 16 bytes");
    if (copy_bytes == 0) return UNIVL_OK;
    const long n16 = copy_bytes / 16;
    long bx = (n16 + 255) / 256; if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)bx, rows), dim3(256), 0, stream, reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<uint4*>(dst), idx, (long)(row_stride / 16), n16);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_log_softmax_rows(float* x, int32_t rows, int32_t n, int64_t ld, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(x && rows > 0 && n > 0 && ld >= n, UNIVL_EINVAL, "univl_log_softmax_rows: bad argument");
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, n, (long)ld);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_simdense_fwd(const float* x, const float* w, const float* b, int32_t rows, float* out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(x && w && b && out && rows > 0, UNIVL_EINVAL, "univl_simdense_fwd: bad argument");
    hipLaunchKernelGGL(simdense_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, w, b, rows, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_simdense_bwd(const float* ds, const float* x, const float* w, int32_t rows, float* dx, float* dw, float* db,
                                  hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(ds && x && w && dx && dw && db && rows > 0, UNIVL_EINVAL, "univl_simdense_bwd: bad argument");
    hipLaunchKernelGGL(simdense_bwd_kernel, dim3(1), dim3(256), 0, stream, ds, x, w, rows, dx, dw, db);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_ce_loss(int32_t dtype, const float* logits, int64_t ld, const int64_t* labels, int32_t rows, int32_t V,
                             int32_t ignore_index, float* scratch2, float* loss, void* dlogits, int64_t lddl, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(logits && labels && scratch2 && loss && dlogits && rows > 0 && V > 0 && ld >= V && lddl >= V, UNIVL_EINVAL,
                    "univl_ce_loss: bad argument");
    UNIVL_CHECK_ARG(dtype == UNIVL_DT_F32 || dtype == UNIVL_DT_BF16, UNIVL_EUNSUPPORTED, "univl_ce_loss: dtype %d", dtype);
    float* rowloss = nullptr;
    if (univl_deterministic()) {
        rowloss = static_cast<float*>(univl_det_alloc((size_t)rows * sizeof(float)));
        if (!rowloss) return UNIVL_EINVAL;
    }
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, stream, labels, rows, ignore_index, scratch2);
    if (dtype == UNIVL_DT_BF16)
        hipLaunchKernelGGL((ce_row_kernel<__bf16>), dim3(rows), dim3(256), 0, stream, logits, (long)ld, labels, V, ignore_index,
                           scratch2, reinterpret_cast<__bf16*>(dlogits), (long)lddl, rowloss);
    else
        hipLaunchKernelGGL((ce_row_kernel<float>), dim3(rows), dim3(256), 0, stream, logits, (long)ld, labels, V, ignore_index,
                           scratch2, reinterpret_cast<float*>(dlogits), (long)lddl, rowloss);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, stream, scratch2, loss, rowloss, rows);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_mfm_nce_loss(const float* logits, int64_t ld, const int64_t* vmask, const int64_t* labels, int32_t n,
                                  float* scratch2, float* loss, float* dlogits, int64_t lddl, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(logits && vmask && labels && scratch2 && loss && dlogits && n > 0 && ld >= n && lddl >= n, UNIVL_EINVAL,
                    "univl_mfm_nce_loss: bad argument");
    float* rowloss = nullptr;
    if (univl_deterministic()) {
        rowloss = static_cast<float*>(univl_det_alloc((size_t)n * sizeof(float)));
        if (!rowloss) return UNIVL_EINVAL;
    }
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, stream, labels, n, -1, scratch2);
    hipLaunchKernelGGL(mfm_kernel, dim3(n), dim3(256), 0, stream, logits, (long)ld, vmask, labels, n, scratch2, dlogits, (long)lddl, rowloss);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, stream, scratch2, loss, rowloss, n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
