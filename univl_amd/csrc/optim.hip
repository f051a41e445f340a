// Fused multi-tensor BertAdam + gradient clipping over FLAT fp32 buffers.
//
// Reference semantics (per parameter tensor, modules/optimization.py:103-168 and main_task_retrieval.py:347-353):
//   1. torch.nn.utils.clip_grad_norm_(all params, 1.0): coef = min(1, 1/(||g||_2 + 1e-6)), g *= coef
//   2. per parameter: clip_grad_norm_(p, max_grad_norm) again                       optimization.py:135-136
//   3. m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; update = m / (sqrt(v) + e)     :141-144 (no bias correction)
//   4. update += weight_decay * p                                                   :153-154
//   5. p -= lr * warmup_linear(step / t_total, warmup) * update ; step += 1         :156-166
// The reference runs this as a Python loop of ~12 tiny kernels per tensor (3-4 k launches per step); here it is
// three launches: per-tensor sum of squares (one streaming read of g), a 1-block scalar kernel, and one streaming
// update kernel (reads p,g,m,v; writes p,m,v and the bf16 shadow of p used by the GEMMs): 30 B/param, HBM-bound.
//
// Work decomposition: the host splits every tensor into chunks of <= CHUNK elements (chunk_seg/off/len arrays);
// one workgroup per chunk, so no chunk straddles two tensors and per-tensor scalars are workgroup-uniform.
#include <stdlib.h>
#include "common.h"
#include "univl_hip.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// det_part != nullptr (deterministic mode, common.h): the chunk's sum goes to det_part[chunk] and sumsq_seg_finish_kernel adds a
// tensor's chunks in chunk order; otherwise one fp32 atomic per chunk into the tensor's slot.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, const UnivlSeg* segs, const int32_t* chunk_seg,
                                                    const int64_t* chunk_off, const int32_t* chunk_len, float* sumsq, float* det_part) {
    __shared__ float red[4];
    const int c = blockIdx.x, seg = chunk_seg[c];
    if (!segs[seg].active) { if (det_part && threadIdx.x == 0) det_part[c] = 0.f; return; }
    const float* p = g + chunk_off[c];
    const int len = chunk_len[c];
    float acc = 0.f;
    const int nv = ((((uintptr_t)p) & 15) == 0) ? len / 4 : 0;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    int i = threadIdx.x;
    // full chunks (8192 floats = 8 float4 per thread): all 8 loads in flight before the first use
    for (; i + 7 * 256 < nv; i += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p4[i + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
    }
    for (; i < nv; i += 256) {
        const float4 v = p4[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (int j = nv * 4 + threadIdx.x; j < len; j += 256) acc += p[j] * p[j];
    const float s = block_sum256(acc, red);
    if (threadIdx.x == 0) {
        if (det_part) det_part[c] = s;
        else unsafeAtomicAdd(sumsq + seg, s);
    }
}

// one thread per chunk; the FIRST chunk of a tensor walks the tensor's run of chunks (the host lists them contiguously)
__global__ __launch_bounds__(256) void sumsq_seg_finish_kernel(const float* part, const UnivlSeg* segs, const int32_t* chunk_seg, int nchunk,
                                                               float* sumsq) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunk) return;
    const int seg = chunk_seg[c];
    if (c > 0 && chunk_seg[c - 1] == seg) return;
    if (!segs[seg].active) return;
    float acc = 0.f;
    for (int u = c; u < nchunk && chunk_seg[u] == seg; ++u) acc += part[u];
    sumsq[seg] += acc;
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const float* sumsq, const UnivlSeg* segs, int nseg, float max_norm,
                                                        float* coef) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int s = threadIdx.x; s < nseg; s += 256) if (segs[s].active) acc += sumsq[s];
    const float total = sqrtf(block_sum256(acc, red));
    if (threadIdx.x == 0) {
        coef[0] = fminf(1.0f, max_norm / (total + 1e-6f));
        coef[1] = total;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* g, const UnivlSeg* segs, const int32_t* chunk_seg,
                                                    const int64_t* chunk_off, const int32_t* chunk_len, const float* coef) {
    const int c = blockIdx.x;
    if (!segs[chunk_seg[c]].active) return;
    const float k = coef[0];
    if (k == 1.0f) return;
    float* p = g + chunk_off[c];
    const int len = chunk_len[c];
    for (int i = threadIdx.x; i < len; i += 256) p[i] *= k;
}

__device__ __forceinline__ float warmup_linear_f(float x, float warmup) {
    // optimization.py:38-43
    if (x < warmup) return x / warmup;
    return fmaxf((x - 1.0f) / (warmup - 1.0f), 0.0f);
}

// per-tensor scalars: scal[2s] = total gradient scale (global clip x per-parameter clip), scal[2s+1] = scheduled lr
__global__ void adam_prep_kernel(UnivlAdam a) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.nseg) return;
    const UnivlSeg sg = a.segs[s];
    if (!sg.active) return;
    const float gc = a.coef ? a.coef[0] : 1.0f;
    float scale = gc;
    if (sg.max_grad_norm > 0.f) {
        const float nrm = sqrtf(a.sumsq[s]) * gc;
        scale *= fminf(1.0f, sg.max_grad_norm / (nrm + 1e-6f));
    }
    const int st = a.step[s];
    float lr = sg.lr;
    if (a.t_total != -1) {
        const float x = (float)st / (float)a.t_total;
        float f = warmup_linear_f(x, a.warmup);
        if (a.schedule == 1) f = x < a.warmup ? x / a.warmup : 0.5f * (1.0f + cosf(3.14159265358979323846f * x));   // warmup_cosine :26-29
        if (a.schedule == 2) f = x < a.warmup ? x / a.warmup : 1.0f;                                                // warmup_constant :31-36
        lr *= f;
    }
    a.seg_scalars[2 * s] = scale;
    a.seg_scalars[2 * s + 1] = lr;
    a.step[s] = st + 1;
}

// chunks [c0, c1) of the chunk table; a grid smaller than the range walks it with stride gridDim.x (a throttled launch
// that leaves CUs to concurrently running kernels: the pipelined step overlaps this update with the next forward).
// NT: the fp32 streams (p, g, m, v: 28 of the 30 bytes per parameter, touched exactly once per step) go through
// non-temporal loads / stores so that they do not push the bf16 shadow -- the only thing the next forward reads -- out of
// the caches (UNIVL_ADAM_NT, A/B in profiles/README.md).
template <bool NT> __device__ __forceinline__ f32x4_t ld4(const float* p, int i) {
    const f32x4_t* q = reinterpret_cast<const f32x4_t*>(p) + i;
    if (NT) return __builtin_nontemporal_load(q);
    return *q;
}
template <bool NT> __device__ __forceinline__ void st4(float* p, int i, f32x4_t v) {
    f32x4_t* q = reinterpret_cast<f32x4_t*>(p) + i;
    if (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void adam_apply_kernel(UnivlAdam a, int c0, int c1) {
  for (int c = c0 + blockIdx.x; c < c1; c += gridDim.x) {
    const int seg = a.chunk_seg[c];
    const UnivlSeg sg = a.segs[seg];
    if (!sg.active) continue;
    const float gs = a.seg_scalars[2 * seg], lr = a.seg_scalars[2 * seg + 1], wd = sg.weight_decay;
    const float b1 = a.b1, b2 = a.b2, eps = a.eps;
    const int64_t off = a.chunk_off[c];
    const int len = a.chunk_len[c];
    float* p = a.p + off; const float* g = a.g + off; float* m = a.m + off; float* v = a.v + off;
    __bf16* p16 = a.p16 ? reinterpret_cast<__bf16*>(a.p16) + off : nullptr;
    __bf16* p16lo = (a.p16 && a.p16_lo) ? reinterpret_cast<__bf16*>(a.p16_lo) + off : nullptr;      // lo half of the shadow pair
    const int nv = ((off & 3) == 0) ? len / 4 : 0;
    if (a.row_flags != nullptr && seg == a.flag_seg) {
        // rows nobody ever touched (UnivlAdam.row_flags): g = m = v = 0 exactly, the update is the weight decay alone
        const long r0 = (off - sg.offset) / a.row_len, r1 = (off - sg.offset + len + a.row_len - 1) / a.row_len;
        int touched = 0;
        for (long r = r0 + threadIdx.x; r < r1; r += 256) touched |= a.row_flags[r];
        if (!__syncthreads_or(touched)) {
            for (int i = threadIdx.x; i < nv; i += 256) {
                f32x4_t pp = ld4<NT>(p, i);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float upd = wd * pp[e]; pp[e] -= lr * upd; }
                st4<NT>(p, i, pp);
                if (p16) {
                    bf16x4_t w;
                    w[0] = (__bf16)pp[0]; w[1] = (__bf16)pp[1]; w[2] = (__bf16)pp[2]; w[3] = (__bf16)pp[3];
                    reinterpret_cast<bf16x4_t*>(p16)[i] = w;
                    if (p16lo) {
                        bf16x4_t l;
                        l[0] = (__bf16)(pp[0] - (float)w[0]); l[1] = (__bf16)(pp[1] - (float)w[1]);
                        l[2] = (__bf16)(pp[2] - (float)w[2]); l[3] = (__bf16)(pp[3] - (float)w[3]);
                        reinterpret_cast<bf16x4_t*>(p16lo)[i] = l;
                    }
                }
            }
            for (int i = nv * 4 + threadIdx.x; i < len; i += 256) {
                const float pi = p[i] - lr * (wd * p[i]);
                p[i] = pi;
                if (p16) p16[i] = (__bf16)pi;
                if (p16lo) p16lo[i] = (__bf16)(pi - (float)(__bf16)pi);
            }
            continue;
        }
    }
    auto update = [&](int i, f32x4_t pp, const f32x4_t gg, f32x4_t mm, f32x4_t vv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gs;
            mm[e] = mm[e] * b1 + (1.0f - b1) * gr;
            vv[e] = vv[e] * b2 + (1.0f - b2) * gr * gr;
            const float upd = mm[e] / (sqrtf(vv[e]) + eps) + wd * pp[e];
            pp[e] -= lr * upd;
        }
        st4<NT>(p, i, pp);
        st4<NT>(m, i, mm);
        st4<NT>(v, i, vv);
        if (p16) {
            bf16x4_t w;
            w[0] = (__bf16)pp[0]; w[1] = (__bf16)pp[1]; w[2] = (__bf16)pp[2]; w[3] = (__bf16)pp[3];
            reinterpret_cast<bf16x4_t*>(p16)[i] = w;
            if (p16lo) {
                bf16x4_t l;
                l[0] = (__bf16)(pp[0] - (float)w[0]); l[1] = (__bf16)(pp[1] - (float)w[1]);
                l[2] = (__bf16)(pp[2] - (float)w[2]); l[3] = (__bf16)(pp[3] - (float)w[3]);
                reinterpret_cast<bf16x4_t*>(p16lo)[i] = l;
            }
        }
    };
    int i = threadIdx.x;
    // two vectors per thread per trip: all eight 16-byte loads are issued before the first store (p, m, v are read and
    // written through the same pointers, so the compiler cannot hoist the next trip's loads above this trip's stores)
    for (; i + 256 < nv; i += 512) {
        const f32x4_t p0 = ld4<NT>(p, i), p1 = ld4<NT>(p, i + 256);
        const f32x4_t g0 = ld4<NT>(g, i), g1 = ld4<NT>(g, i + 256);
        const f32x4_t m0 = ld4<NT>(m, i), m1 = ld4<NT>(m, i + 256);
        const f32x4_t v0 = ld4<NT>(v, i), v1 = ld4<NT>(v, i + 256);
        update(i, p0, g0, m0, v0);
        update(i + 256, p1, g1, m1, v1);
    }
    for (; i < nv; i += 256) update(i, ld4<NT>(p, i), ld4<NT>(g, i), ld4<NT>(m, i), ld4<NT>(v, i));
    for (int i = nv * 4 + threadIdx.x; i < len; i += 256) {
        const float gr = g[i] * gs;
        const float mi = m[i] * b1 + (1.0f - b1) * gr;
        const float vi = v[i] * b2 + (1.0f - b2) * gr * gr;
        const float upd = mi / (sqrtf(vi) + eps) + wd * p[i];
        const float pi = p[i] - lr * upd;
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p16) p16[i] = (__bf16)pi;
        if (p16lo) p16lo[i] = (__bf16)(pi - (float)(__bf16)pi);
    }
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* p, __bf16* o, int64_t n) {
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        bf16x4_t w;
        w[0] = (__bf16)v.x; w[1] = (__bf16)v.y; w[2] = (__bf16)v.z; w[3] = (__bf16)v.w;
        reinterpret_cast<bf16x4_t*>(o)[i] = w;
    }
    if (blockIdx.x == 0) for (int64_t i = nv * 4 + threadIdx.x; i < n; i += 256) o[i] = (__bf16)p[i];
}

__global__ __launch_bounds__(256) void cast_pair_kernel(const float* p, __bf16* hi, __bf16* lo, int64_t n) {
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        bf16x4_t w, l;
        w[0] = (__bf16)v.x; w[1] = (__bf16)v.y; w[2] = (__bf16)v.z; w[3] = (__bf16)v.w;
        l[0] = (__bf16)(v.x - (float)w[0]); l[1] = (__bf16)(v.y - (float)w[1]); l[2] = (__bf16)(v.z - (float)w[2]); l[3] = (__bf16)(v.w - (float)w[3]);
        if (hi) reinterpret_cast<bf16x4_t*>(hi)[i] = w;
        reinterpret_cast<bf16x4_t*>(lo)[i] = l;
    }
    if (blockIdx.x == 0) for (int64_t i = nv * 4 + threadIdx.x; i < n; i += 256) {
        const __bf16 h = (__bf16)p[i];
        if (hi) hi[i] = h;
        lo[i] = (__bf16)(p[i] - (float)h);
    }
}

__global__ __launch_bounds__(256) void uncast_kernel(const __bf16* p, float* o, int64_t n) {
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const bf16x4_t w = reinterpret_cast<const bf16x4_t*>(p)[i];
        reinterpret_cast<float4*>(o)[i] = make_float4((float)w[0], (float)w[1], (float)w[2], (float)w[3]);
    }
    if (blockIdx.x == 0) for (int64_t i = nv * 4 + threadIdx.x; i < n; i += 256) o[i] = (float)p[i];
}

}  // namespace

bool univl_adam_nt() {
    static const int v = [] { const char* e = getenv("UNIVL_ADAM_NT"); return e ? atoi(e) : 1; }();   // measured: 3.16 -> 3.06 ms per step
    return v != 0;
}


extern "C" int univl_grad_sumsq(const float* g, const UnivlSeg* segs, int32_t nseg, const int32_t* chunk_seg,
                                const int64_t* chunk_off, const int32_t* chunk_len, int32_t nchunk, float* sumsq,
                                hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(g && segs && chunk_seg && chunk_off && chunk_len && sumsq && nseg > 0 && nchunk > 0, UNIVL_EINVAL,
                    "univl_grad_sumsq: bad argument");
    float* part = nullptr;
    if (univl_deterministic()) {
        part = static_cast<float*>(univl_det_alloc((size_t)nchunk * sizeof(float)));
        if (!part) return UNIVL_EINVAL;
    }
    hipLaunchKernelGGL(sumsq_kernel, dim3(nchunk), dim3(256), 0, stream, g, segs, chunk_seg, chunk_off, chunk_len, sumsq, part);
    if (part) hipLaunchKernelGGL(sumsq_seg_finish_kernel, dim3((nchunk + 255) / 256), dim3(256), 0, stream, part, segs, chunk_seg, nchunk, sumsq);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* partials, const int32_t* seg, const int32_t* start,
                                                           const int32_t* count, float* out) {
    __shared__ float red[4];
    const int e = blockIdx.x;
    const float* p = partials + start[e];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count[e]; i += 256) acc += p[i];
    const float s = block_sum256(acc, red);
    if (threadIdx.x == 0) out[seg[e]] = s;
}

extern "C" int univl_sumsq_finish(const float* partials, const int32_t* seg, const int32_t* start, const int32_t* count,
                                  int32_t n, float* out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(partials && seg && start && count && out && n > 0, UNIVL_EINVAL, "univl_sumsq_finish: bad argument");
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(n), dim3(256), 0, stream, partials, seg, start, count, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_clip_coef(const float* sumsq, const UnivlSeg* segs, int32_t nseg, float max_norm, float* coef,
                               hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(sumsq && segs && coef && nseg > 0, UNIVL_EINVAL, "univl_clip_coef: bad argument");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, sumsq, segs, nseg, max_norm, coef);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_scale_grads(float* g, const UnivlSeg* segs, const int32_t* chunk_seg, const int64_t* chunk_off,
                                 const int32_t* chunk_len, int32_t nchunk, const float* coef, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(g && segs && chunk_seg && chunk_off && chunk_len && coef && nchunk > 0, UNIVL_EINVAL,
                    "univl_scale_grads: bad argument");
    hipLaunchKernelGGL(scale_kernel, dim3(nchunk), dim3(256), 0, stream, g, segs, chunk_seg, chunk_off, chunk_len, coef);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_bert_adam(const UnivlAdam* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(d && d->p && d->g && d->m && d->v && d->segs && d->chunk_seg && d->chunk_off && d->chunk_len &&
                        d->sumsq && d->step && d->seg_scalars && d->nseg > 0 && d->nchunk > 0,
                    UNIVL_EINVAL, "univl_bert_adam: bad argument");
    hipLaunchKernelGGL(adam_prep_kernel, dim3((d->nseg + 255) / 256), dim3(256), 0, stream, *d);
    if (univl_adam_nt()) hipLaunchKernelGGL(adam_apply_kernel<true>, dim3(d->nchunk), dim3(256), 0, stream, *d, 0, d->nchunk);
    else hipLaunchKernelGGL(adam_apply_kernel<false>, dim3(d->nchunk), dim3(256), 0, stream, *d, 0, d->nchunk);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_bert_adam_range(const UnivlAdam* d, int32_t chunk_begin, int32_t chunk_count, int32_t do_prep,
                                     int32_t max_blocks, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(d && d->p && d->g && d->m && d->v && d->segs && d->chunk_seg && d->chunk_off && d->chunk_len &&
                        d->sumsq && d->step && d->seg_scalars && d->nseg > 0 && d->nchunk > 0,
                    UNIVL_EINVAL, "univl_bert_adam_range: bad argument");
    UNIVL_CHECK_ARG(chunk_begin >= 0 && chunk_count >= 0 && chunk_begin + chunk_count <= d->nchunk, UNIVL_EINVAL,
                    "univl_bert_adam_range: chunks [%d, +%d) of %d", chunk_begin, chunk_count, d->nchunk);
    if (do_prep) hipLaunchKernelGGL(adam_prep_kernel, dim3((d->nseg + 255) / 256), dim3(256), 0, stream, *d);
    if (chunk_count > 0) {
        const int grid = (max_blocks > 0 && max_blocks < chunk_count) ? max_blocks : chunk_count;
        if (univl_adam_nt()) hipLaunchKernelGGL(adam_apply_kernel<true>, dim3(grid), dim3(256), 0, stream, *d, chunk_begin, chunk_begin + chunk_count);
        else hipLaunchKernelGGL(adam_apply_kernel<false>, dim3(grid), dim3(256), 0, stream, *d, chunk_begin, chunk_begin + chunk_count);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

__global__ void bump_kernel(uint64_t* c) { c[0] += 1; }

extern "C" int univl_bump_counter(uint64_t* ctr, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(ctr != nullptr, UNIVL_EINVAL, "univl_bump_counter: null");
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, stream, ctr);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

__global__ void stamp_kernel(uint64_t* out) { out[0] = wall_clock64(); }

extern "C" int univl_stamp(uint64_t* out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(out != nullptr, UNIVL_EINVAL, "univl_stamp: null");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, stream, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_cast_bf16(const float* p, void* p16, int64_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(p && p16 && n > 0 && aligned16(p) && ((((uintptr_t)p16) & 7) == 0), UNIVL_EINVAL, "univl_cast_bf16: bad argument");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, reinterpret_cast<__bf16*>(p16), n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_cast_bf16_pair(const float* p, void* p16, void* p16_lo, int64_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(p && p16_lo && n > 0 && aligned16(p) && ((((uintptr_t)p16 | (uintptr_t)p16_lo) & 7) == 0), UNIVL_EINVAL,
                    "univl_cast_bf16_pair: bad argument");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, reinterpret_cast<__bf16*>(p16),
                       reinterpret_cast<__bf16*>(p16_lo), n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_cast_f32(const void* p16, float* p, int64_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(p && p16 && n > 0 && aligned16(p) && ((((uintptr_t)p16) & 7) == 0), UNIVL_EINVAL, "univl_cast_f32: bad argument");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(uncast_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const __bf16*>(p16), p, n);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
