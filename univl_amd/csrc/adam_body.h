// One chunk of the fused BertAdam update (modules/optimization.py:103-168) as a device function, for kernels that carry optimizer
// work beside their own (gemm.hip: gemm_adam_kernel, EXPERIMENTAL).  Same arithmetic, operation order and non-temporal policy as
// adam_apply_kernel<NT> in optim.hip (which stays the product path's kernel); NTH = threads of the calling workgroup.
#pragma once
#include "common.h"
#include "univl_hip.h"

bool univl_adam_nt();      // optim.hip: UNIVL_ADAM_NT (default 1): non-temporal loads / stores of the 28 fp32 bytes per parameter

template <bool NT> __device__ __forceinline__ f32x4_t adam_ld4(const float* p, int i) {
    const f32x4_t* q = reinterpret_cast<const f32x4_t*>(p) + i;
    if (NT) return __builtin_nontemporal_load(q);
    return *q;
}
template <bool NT> __device__ __forceinline__ void adam_st4(float* p, int i, f32x4_t v) {
    f32x4_t* q = reinterpret_cast<f32x4_t*>(p) + i;
    if (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}

// LO: the instantiation also keeps the lo half of the shadow pair (UnivlAdam.p16_lo).  The rider kernel of the 64 x 128 tile is built for
// <= 80 VGPRs (three workgroups per unit) and spilled two more with it: it instantiates LO = false and the host keeps an update that carries
// p16_lo out of that kernel (univl_gemm_rider).
template <bool NT, int NTH, bool LO = true>
__device__ __forceinline__ void adam_chunk(const UnivlAdam& a, int c) {
    const int seg = a.chunk_seg[c];
    const UnivlSeg sg = a.segs[seg];
    if (!sg.active) return;
    const float gs = a.seg_scalars[2 * seg], lr = a.seg_scalars[2 * seg + 1], wd = sg.weight_decay;
    const float b1 = a.b1, b2 = a.b2, eps = a.eps;
    const int64_t off = a.chunk_off[c];
    const int len = a.chunk_len[c];
    float* p = a.p + off; const float* g = a.g + off; float* m = a.m + off; float* v = a.v + off;
    __bf16* p16 = a.p16 ? reinterpret_cast<__bf16*>(a.p16) + off : nullptr;
    __bf16* p16lo = (LO && a.p16 && a.p16_lo) ? reinterpret_cast<__bf16*>(a.p16_lo) + off : nullptr;      // lo half of the shadow pair
    const int nv = ((off & 3) == 0) ? len / 4 : 0;
    auto update = [&](int i, f32x4_t pp, const f32x4_t gg, f32x4_t mm, f32x4_t vv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gs;
            mm[e] = mm[e] * b1 + (1.0f - b1) * gr;
            vv[e] = vv[e] * b2 + (1.0f - b2) * gr * gr;
            const float upd = mm[e] / (sqrtf(vv[e]) + eps) + wd * pp[e];
            pp[e] -= lr * upd;
        }
        adam_st4<NT>(p, i, pp);
        adam_st4<NT>(m, i, mm);
        adam_st4<NT>(v, i, vv);
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
    for (; i + NTH < nv; i += 2 * NTH) {
        const f32x4_t p0 = adam_ld4<NT>(p, i), p1 = adam_ld4<NT>(p, i + NTH);
        const f32x4_t g0 = adam_ld4<NT>(g, i), g1 = adam_ld4<NT>(g, i + NTH);
        const f32x4_t m0 = adam_ld4<NT>(m, i), m1 = adam_ld4<NT>(m, i + NTH);
        const f32x4_t v0 = adam_ld4<NT>(v, i), v1 = adam_ld4<NT>(v, i + NTH);
        update(i, p0, g0, m0, v0);
        update(i + NTH, p1, g1, m1, v1);
    }
    for (; i < nv; i += NTH) update(i, adam_ld4<NT>(p, i), adam_ld4<NT>(g, i), adam_ld4<NT>(m, i), adam_ld4<NT>(v, i));
    for (int j = nv * 4 + threadIdx.x; j < len; j += NTH) {
        const float gr = g[j] * gs;
        const float mi = m[j] * b1 + (1.0f - b1) * gr;
        const float vi = v[j] * b2 + (1.0f - b2) * gr * gr;
        const float upd = mi / (sqrtf(vi) + eps) + wd * p[j];
        const float pi = p[j] - lr * upd;
        p[j] = pi; m[j] = mi; v[j] = vi;
        if (p16) p16[j] = (__bf16)pi;
        if (p16lo) p16lo[j] = (__bf16)(pi - (float)(__bf16)pi);
    }
}
