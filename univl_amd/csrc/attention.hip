// Attention core of BertSelfAttention / VisualSelfAttention / CrossSelfAttention / decoder MultiHeadAttention
// (reference: modules/module_bert.py:181-197 and the identical copies module_visual.py:165-181,
// module_cross.py:172-188, module_decoder.py:230-247):
//
//      P = softmax(Q K^T / sqrt(64) + mask)   (mask added AFTER the scaling; 0 / -10000, never -inf)
//      O = dropout(P) V
//
// One workgroup = one (batch row, head) x one block of 64 queries (forward, dQ) or 64 keys (dK/dV); each of its
// 4 waves owns 16 of them.  Sequences here are <= 224, so the whole K and V (or Q and dO) of a head are staged
// once in LDS; QK^T and PV run on MFMA; the scores never leave registers: the product is computed TRANSPOSED
// (S^T = K Q^T) so that every lane owns one query column, the softmax row-reduce is an in-lane reduction plus two
// wave shuffles (lane^16, lane^32), and the probability tile in accumulator layout is already the B operand of
// the following P^T-contraction (see the slot map in common.h).  V (and K, Q, dO in the backward) are consumed
// as T-major operands straight from their row-major LDS image with the gfx950 transpose read.
//
// The backward recomputes P from the saved row log-sum-exp (module: nothing but lse is kept from the forward).
#include <stdlib.h>
#include "common.h"
#include "univl_hip.h"

namespace {

#include "attn_body.h"

// ------------------------------------------------------------------------------------------------ forward (body: attn_body.h)
template <typename T, int MAXKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(UnivlAttention p, int Sk_pad, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    attn_fwd_body<T, MAXKT, false>(p, Sk_pad, scale, (int)blockIdx.x, (int)blockIdx.y, smem_raw);
}

// ------------------------------------------------------------------------------------------------ backward (body: attn_body.h)
template <typename T, int TRIPS, bool DUAL>
__global__ __launch_bounds__(256) void attn_bwd_kernel(UnivlAttention p, int Sk_pad, int Sq_pad, int nqb, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    attn_bwd_body<T, TRIPS, DUAL, false>(p, Sk_pad, Sq_pad, nqb, scale, (int)blockIdx.x, (int)blockIdx.y, smem_raw, AttnNoDo{});
}

template <typename T, int MAXKT>
int launch_fwd(const UnivlAttention* d, int Sk_pad, hipStream_t stream) {
    const size_t smem = (size_t)Sk_pad * (AttnCfg<T>::PK + AttnCfg<T>::PT) * sizeof(T) + Sk_pad * sizeof(float);
    static bool attr_done[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(attn_fwd_kernel<T, MAXKT>, 160 * 1024, attr_done);
    dim3 grid(d->B * d->H, (d->Sq + 63) / 64);
    hipLaunchKernelGGL((attn_fwd_kernel<T, MAXKT>), grid, dim3(256), smem, stream, *d, Sk_pad, 0.125f);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

template <typename T>
int dispatch_fwd(const UnivlAttention* d, hipStream_t stream) {
    const int CH = Mma<T>::CH;
    const int Sk_pad = (d->Sk + CH - 1) / CH * CH;
    const int nkt = Sk_pad / 16;
    if (nkt <= 4) return launch_fwd<T, 4>(d, Sk_pad, stream);
    if (nkt <= 8) return launch_fwd<T, 8>(d, Sk_pad, stream);
    if (nkt <= 16) return launch_fwd<T, 16>(d, Sk_pad, stream);
    return launch_fwd<T, 24>(d, Sk_pad, stream);
}

template <typename T, int TR, bool DUAL>
int launch_bwd(const UnivlAttention* d, int Sk_pad, int Sq_pad, size_t smem, hipStream_t stream) {
    const int nqb = (d->Sq + 63) / 64, nkb = (d->Sk + 63) / 64;
    static bool attr_done[UNIVL_MAX_DEVICES] = {};
    univl_allow_lds(attn_bwd_kernel<T, TR, DUAL>, 160 * 1024, attr_done);
    dim3 grid(d->B * d->H, nqb + nkb);
    hipLaunchKernelGGL((attn_bwd_kernel<T, TR, DUAL>), grid, dim3(256), smem, stream, *d, Sk_pad, Sq_pad, nqb, 0.125f);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// Sequences up to this many (padded) positions get the two-image staging in the backward (bf16 only): 64 rows x (144 + 160) B x 2
// matrices = 38 KB per workgroup, four workgroups per compute unit.  Longer sequences keep one image per matrix: at S = 224 the two
// images of Q and dO are 136 KB -- one 4-wave workgroup per compute unit.
constexpr int ATTN_DUAL_MAX = 64;

template <typename T>
int dispatch_bwd(const UnivlAttention* d, hipStream_t stream) {
    using C = AttnCfg<T>;
    const int CH = Mma<T>::CH;
    const int Sk_pad = (d->Sk + CH - 1) / CH * CH, Sq_pad = (d->Sq + CH - 1) / CH * CH;
    int dual_max = ATTN_DUAL_MAX;
#ifdef UNIVL_TRACE
    if (const char* e = getenv("UNIVL_ATTN_DUAL_MAX")) dual_max = atoi(e);      // measurement build only (scripts/mb_attention.py)
#endif
    const bool dual = sizeof(T) == 2 && Sk_pad <= dual_max && Sq_pad <= dual_max;
    const size_t smemA = (size_t)Sk_pad * (C::PT + C::PK + (dual ? C::PK : 0)) * sizeof(T) + Sk_pad * sizeof(float);
    const size_t smemB = (size_t)Sq_pad * (2 * C::PT + (dual ? 2 * C::PK : 0)) * sizeof(T) + (Sk_pad + 2 * Sq_pad) * sizeof(float);
    const size_t smem = smemA > smemB ? smemA : smemB;
    UNIVL_CHECK_ARG(smem <= 160 * 1024, UNIVL_EUNSUPPORTED, "univl_attention_bwd: %zu bytes of LDS", smem);
    constexpr int PIECES64 = 64 * (HD / C::EPC);                   // 16-byte pieces of a 64-row tile
    const bool small = Sk_pad <= 64 && Sq_pad <= 64;
    constexpr int TR = (PIECES64 + 511) / 512;
    if constexpr (sizeof(T) == 2) {
        if (dual) return small ? launch_bwd<T, TR, true>(d, Sk_pad, Sq_pad, smem, stream) : launch_bwd<T, 0, true>(d, Sk_pad, Sq_pad, smem, stream);
    }
    return small ? launch_bwd<T, TR, false>(d, Sk_pad, Sq_pad, smem, stream) : launch_bwd<T, 0, false>(d, Sk_pad, Sq_pad, smem, stream);
}

}  // namespace

extern "C" int univl_attention_fwd(const UnivlAttention* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    int rc = attn_check(d, "univl_attention_fwd", false);
    if (rc) return rc;
    return d->dtype == UNIVL_DT_BF16 ? dispatch_fwd<__bf16>(d, stream) : dispatch_fwd<float>(d, stream);
}

extern "C" int univl_attention_bwd(const UnivlAttention* d, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    int rc = attn_check(d, "univl_attention_bwd", true);
    if (rc) return rc;
    return d->dtype == UNIVL_DT_BF16 ? dispatch_bwd<__bf16>(d, stream) : dispatch_bwd<float>(d, stream);
}
