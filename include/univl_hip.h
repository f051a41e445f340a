/* libunivl_hip.so -- C ABI of the MI355X (gfx950) kernels behind the UniVL hot path.
 *
 * The reference (microsoft/UniVL) has NO native layer: its "operator API" is the Python class
 * modules/modeling.py::UniVL and all arithmetic is delegated to PyTorch ATen.  Each entry
 * point below therefore replaces an ATen op SEQUENCE at a cited call site of the reference; the Python host
 * (univl_amd/) mirrors the reference's module surface and calls these through ctypes.
 *
 * Contract (SURVEY.md section 8b "C-ABI"):
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch's caching
 *     allocator on the Python side).  The library never allocates, frees or synchronises on the hot path.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t) and returns 0 on success, <0 for an argument error (UNIVL_E*), >0 for a hipError_t; univl_last_error() describes it.
 *   - the DEVICE is the one `stream` belongs to (hipStreamGetDevice), not the calling thread's current device: the
 *     reference's evaluation fan-out calls the model from one thread per GPU (util.py:21-60) and autograd's backward
 *     threads carry their own current device.  The null stream means the current device.
 *   - re-entrant; no global mutable state besides the thread-local error string, per-device "large LDS enabled" flags and the
 *     deterministic-mode switch below (with its per-device scratch ring -- the one thing the library allocates, and only when asked).
 *   - dtype: UNIVL_F32 (parity mode, exact-fp32 MFMA) or UNIVL_BF16 (production: bf16 operands, fp32
 *     accumulate, fp32 LayerNorm/softmax/residual stream).
 *   - RNG for dropout is (seed, offset) supplied by the caller; offsets distinguish call sites.  The effective
 *     seed is seed + *seed_dev when seed_dev (a device pointer) is given, so that a captured hipGraph draws a
 *     fresh mask on every replay (univl_bump_counter advances the device word inside the graph).
 */
#ifndef UNIVL_HIP_H
#define UNIVL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define UNIVL_DT_F32 0
#define UNIVL_DT_BF16 1

const char* univl_last_error(void);
int univl_version(void);
/* sizeof of ABI struct #which (0 Gemm, 1 LayerNorm, 2 Attention, 3 EmbedText, 4 Pool, 5 Seg, 6 Adam, 7 VocabCE) -- lets a
 * foreign-language binding verify its struct mirrors at load time */
int univl_struct_size(int which);
/* number of CUs / name of the current device, for host-side launch heuristics; returns 0 or hipError_t */
int univl_device_info(int* cu_count, char* name, int name_len);
/* univl_init(device): optional, idempotent.  Checks that `device` exists and is a gfx950 part (the only code object in the
 * library), returns 0 / UNIVL_EUNSUPPORTED / hipError_t.  Nothing needs to be initialised before the first call into the
 * library; a host that wants the failure at start-up instead of at the first launch calls this once per GPU
 * (where the reference calls torch.cuda.set_device, main_task_retrieval.py:121).  univl_destroy() releases what the
 * library opened lazily (the RCCL handle of univl_allreduce_bucket); device memory is never owned by the library. */
int univl_init(int device);
/* Clears n <= UNIVL_ZERO_MAX device buffers (16-byte aligned) with one launch; ptrs / bytes are HOST arrays read at call time. */
#define UNIVL_ZERO_MAX 16
int univl_zero_many(void* const* ptrs, const int64_t* bytes, int32_t n, hipStream_t stream);
/* n <= UNIVL_ZERO_MAX device-to-device copies (dst[i] <- src[i], bytes[i] each; non-overlapping) with one launch: the input
 * staging of UniVL.forward (modeling.py:192-202 hands over input_ids / token_type_ids / attention_mask / video / video_mask) and
 * the loss hand-off as ONE kernel node of the captured step instead of one memcpy node each.  16-byte aligned pairs move as
 * 16-byte words, others bytewise.  srcs / dsts / bytes are HOST arrays read at call time. */
int univl_copy_many(const void* const* srcs, void* const* dsts, const int64_t* bytes, int32_t n, hipStream_t stream);
int univl_destroy(void);
/* Deterministic mode.  By default every floating-point sum that several workgroups contribute to -- split-K slices, the column sums
 * behind LayerNorm / bias gradients, embedding and pair-concat scatter-adds, gradient-norm partials, loss accumulators -- is taken
 * with fp32 atomics, whose order (and therefore the last bits of the result) differs from run to run, as it does for the reference's
 * own CUDA kernels (index_add_ / embedding backward).  univl_set_deterministic(1) makes every later launch take these sums in a
 * FIXED order: no split-K, "last workgroup to arrive adds the partials in workgroup order" for column sums, gather-style kernels
 * ("the first source of a destination row sums all its sources in source order") for the scatters.  Two runs on the same inputs
 * are then bit-identical, at a cost of a few extra launches -- the parity tests run in this mode, production does not.  The call
 * allocates a scratch ring on the CURRENT device (UNIVL_DET_ARENA_MB, default 1024) and must therefore be made outside any stream
 * capture, once per device; plans / graphs built before a switch keep the mode they were built in.  Returns 0 or UNIVL_EINVAL. */
int univl_set_deterministic(int on);
int univl_get_deterministic(void);
/* Data-parallel gradient exchange for hosts that own an RCCL communicator themselves (the Python host goes through
 * torch.distributed's "nccl" backend = RCCL, main_task_retrieval.py:23,197-198, instead): in-place all-reduce of
 * buf[0..n) (dtype UNIVL_DT_F32 / UNIVL_DT_BF16) over `comm` (an ncclComm_t) on the side stream `side`,
 * op = average if `average` else sum.  Only enqueues.  RCCL is resolved at first use (the copy already loaded into
 * the process, else librccl.so); UNIVL_EUNSUPPORTED if it cannot be found, >0 = ncclResult_t + 1000. */
int univl_allreduce_bucket(void* buf, size_t n, int dtype, int average, void* comm, hipStream_t side);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = epi( alpha * A_op[M,K] . B_op[N,K]^T ).   trans_x = 0: operand stored [rows][K] (row-major, ld);
 * trans_x = 1: operand stored [K][rows].  Replaces nn.Linear forward (module_bert.py:172-174,207,233,246 and
 * the Visual/Cross/Decoder copies), its autograd dgrad (trans_b=1) and wgrad (trans_a=trans_b=1), the tied
 * vocabulary classifier (module_bert.py:327-330), torch.matmul of the similarity (modeling.py:389) and torch.mm
 * of the MFM logits (modeling.py:285). */
#define UNIVL_GEMM_ACCUM 1        /* C32 (and dbias) += result                                             */
#define UNIVL_GEMM_GELU_FWD 2     /* aux <- pre-activation (T), result <- gelu(result)  (until_module.py:28) */
#define UNIVL_GEMM_GELU_BWD 4     /* result *= gelu'(aux)                                                  */
#define UNIVL_GEMM_ATOMIC 8       /* internal: split-K atomics                                             */
#define UNIVL_GEMM_DBIAS_ATOMIC 16 /* dbias accumulated with atomics (several row tiles share a bias)       */
#define UNIVL_GEMM_XCD_MAP 64      /* internal: XCD-aware workgroup -> tile map (always on since round 4) */
#define UNIVL_GEMM_PROBE_NOSTORE 128 /* internal, only in the -DUNIVL_TRACE measurement build (UNIVL_GEMM_PROBE=1 there): the epilogue computes but does not store; the product library never sets or tests it */
#define UNIVL_GEMM_PROBE_NT_B 256   /* internal, only in the -DUNIVL_TRACE measurement build (UNIVL_GEMM_NT_B=1 there): non-temporal LDS-DMA of the B operand */
#define UNIVL_GEMM_AUX_F32 512     /* aux holds the GELU pre-activation in fp32 (ldaux in floats) instead of the compute type: removes the one
                                    * non-operand rounding between FFN1 and its GELU' (A/B measurement of DESIGN.md section 2; costs 2 x the
                                    * bytes of that buffer both ways) */
#define UNIVL_GEMM_NT_OUT 32       /* fp32 output written with non-temporal stores: a weight gradient is read next by the optimizer, a whole backward later */
typedef struct UnivlGemm {
    int32_t dtype, trans_a, trans_b;
    int32_t M, N, K;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    float* C32;            /* optional fp32 output                      */
    void* C16;             /* optional output in the compute type       */
    int64_t ldc;
    const float* bias;     /* optional [N]                              */
    const float* R;        /* optional fp32 residual [M,ldr]            */
    int64_t ldr;
    void* aux;             /* GELU pre-activation buffer (compute type) */
    int64_t ldaux;
    float* dbias;          /* optional, wgrad only: [M] sums of A_op rows over K */
    float alpha;
    int32_t flags;
    int32_t ksplit;        /* >1: split the contraction over gridDim.z, fp32 atomics into pre-zeroed C32 */
    int32_t tile;          /* 0 auto; 64 (64x64), 128 (128x128);
                            * 256 (256x256 on the 8-phase body, csrc/gemm256.h: bf16, M and N multiples of 256, K slices multiples of
                            *      128, not (T-major A, K-major B), no in-tile dbias -- else 128);
                            * 12864 / 64128 (128x64 / 64x128; bf16, K-major A, no sumsq / dbias, univl_gemm only -- else 128) */
    /* optional, no split-K: every wave stores the sum of squares of the FINAL values it wrote to
     *   sumsq[(m0 / sumsq_rows) * sumsq_stride + (((m0 % sumsq_rows) / tile) * tiles_x + tile_x) * waves + wave]
     * (sumsq_rows = 0: the whole output is one tensor).  Partial sums, no atomics: thousands of workgroups adding to one
     * address serialise in L2.  univl_sumsq_finish folds them into per-tensor sums.  Lets the weight-gradient GEMMs
     * produce the gradient norms clip_grad_norm_ (main_task_retrieval.py:347) and BertAdam's per-parameter clip
     * (optimization.py:135-136) need, instead of a separate 4 B/param pass.  sumsq_rows must be a multiple of 128.
     * At most rows x N / 1024 partial sums per tensor. */
    float* sumsq; int32_t sumsq_rows; int32_t sumsq_stride;
    int32_t stages;        /* LDS pipeline depth: 0 or 2 (double buffer, the only form since round 4; the field keeps the ABI layout) */
    int32_t waves;         /* waves per workgroup on the 64 / 128 tiles: 0 auto, 4, 8 (bf16, else 4) */
    /* Round 6, bf16 only -- operand PAIRS.  A bf16 operand carries 8 mantissa bits; x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
     * carries 16.  A_lo / B_lo (optional, same layout and leading dimension as A / B) hold the lo halves; the product then walks the
     * contraction once per term in a FIXED order --  A.B,  A.B_lo (if B_lo),  A_lo.B (if A_lo)  -- into the same fp32 accumulators
     * (the lo.lo term is below fp32 resolution of the sum and is dropped).  ksplit divides that concatenated contraction.  A paired
     * product runs on the 64 x 64 tile whatever `tile` says (the only body that carries the term walk), needs K-major A and K a
     * multiple of 128, and is carried by univl_gemm, univl_gemm_rider, univl_gemm_ln and univl_attention_fwd_fused; univl_gemm_pair /
     * univl_gemm_group / univl_attention_bwd_fused return UNIVL_EUNSUPPORTED for it.  The plans pair the forward products'
     * operands up to 768 tokens, where the matrix pipe is idle (what it buys in gradient error).
     * C16_lo (optional): the epilogue also stores lo = bf16(result - bf16(result)) there (same ldc) -- the A_lo of the next product. */
    const void* A_lo; const void* B_lo; void* C16_lo;
} UnivlGemm;
int univl_gemm(const UnivlGemm* desc, hipStream_t stream);
/* n (1..UNIVL_GEMM_GROUP_MAX) independent problems with the same dtype / trans_a / trans_b in ONE launch: the four
 * weight-gradient GEMMs autograd runs for one transformer layer (the wgrad halves of the nn.Linear backward of
 * module_bert.py:172-174,207,233,246).  Members must not alias each other's outputs. */
#define UNIVL_GEMM_GROUP_MAX 4
int univl_gemm_group(const UnivlGemm* descs, int32_t n, hipStream_t stream);
/* The same with at most max_blocks workgroups (0: one per tile): the kernel then walks its tiles.  A launch that is not on
 * the critical path -- a layer's weight gradients -- can run beside the next layer's latency-bound kernels on another
 * stream without taking the compute units away from them. */
int univl_gemm_group_limited(const UnivlGemm* d, int n, int max_blocks, hipStream_t stream);
/* One launch for a dgrad product (dY . W: A K-major, B T-major) and the weight-gradient product that consumes the
 * same upstream gradient (dY^T . X: both T-major) -- the two halves of one nn.Linear backward (e.g. module_bert.py:233, 246): the
 * weight-gradient tiles fill the compute units the latency-bound dgrad leaves idle.  bf16, 64 x 64 tiles only (returns
 * UNIVL_EUNSUPPORTED otherwise -- callers fall back to univl_gemm + univl_gemm_group); dry_run != 0 validates without launching. */
int univl_gemm_pair(const UnivlGemm* dgrad, const UnivlGemm* wgrad, int32_t dry_run, hipStream_t stream);
/* Host-side evaluation of the kernels' workgroup -> tile maps (no device work; lets a CPU test prove they are bijections):
 * what 0: plain-grid map, in out[0..2] = hardware block index, out = the tile that block computes;
 * what 1: out[0] <- position of linear workgroup out[0] in the XCD-grouped tile list of nx tiles;
 * what 2: out[0..2] <- tile number out[0] of an nx x ny x nz problem (a grouped launch's member);
 * what 3: out[0..2] <- the tile local workgroup out[0] of one half of a pair / rider launch computes;
 * what 4: the rider launch of the 64 x 128 tile (round 5), whose update workgroups are spread through the grid in groups of 8: workgroup
 *         out[0] of nx + ny (nx tile slots, ny update workgroups, both multiples of 8) -> out[1] = 1 and out[0] = its update index, or
 *         out[1] = 0 and out[0] = its tile slot (congruent to the workgroup id modulo 8). */
int univl_gemm_tile_map(int32_t what, int32_t nx, int32_t ny, int32_t nz, int32_t gm, int32_t* out);

/* Host-side evaluation of the LDS maps of the 256 x 256 product body (csrc/gemm256.h; no device work -- lets a CPU test prove that the
 * LDS-DMA image and the fragment reads agree and are bank-conflict free):
 * what 0: out[0..1] <- (row, k) of the first of the 8 elements at lane-linear LDS piece idx (0..1023) of a half-tile image;
 * what 1: out[0..7] <- LDS byte offsets of the fragment reads of lane idx (0..63) for the 32-row block at arg (0, 32, 64, 96):
 *         K-major (trans 0) out[0..3] per 16-deep k step, out[4..7] = -1; T-major (trans 1) two transpose reads per k step. */
int univl_gemm256_layout(int32_t what, int32_t trans, int32_t idx, int32_t arg, int32_t* out);

/* ------------------------------------------------------------------------------------------ LayerNorm
 * TF-style LayerNorm (until_module.py:40-53: biased variance, eps inside the sqrt) fused with what surrounds it
 * in the reference:  out = dropout_post( LN( dropout_pre(x) + residual + pos[row % period] ) ).
 *   BertSelfOutput/BertOutput (module_bert.py:207-211,246-250): x = dense output, residual, p_pre.
 *   Visual/Cross embeddings  (module_visual.py:118-131, module_cross.py:123-138): pos (+type via residual), p_post.
 *   NormalizeVideo (modeling.py:88-92): x is float64 (x_f64 = 1), no residual/dropout. */
typedef struct UnivlLayerNorm {
    int32_t dtype;         /* type of out16 / dxd16                                                       */
    int32_t rows, N;       /* N in {768, 1024}                                                             */
    int32_t x_f64;         /* forward input is float64                                                     */
    const void* x;         /* fwd: [rows,N] fp32 (or f64)                                                  */
    const float* residual; /* fwd: optional [rows,N]                                                       */
    const float* pos;      /* fwd: optional [period,N] added to row (row % period)                         */
    int32_t pos_period;
    const float* gamma; const float* beta;
    float eps;
    float* y;              /* fwd out / bwd in: pre-LN sum [rows,N] (may alias x)                          */
    float* stats;          /* fwd out / bwd in: [rows,2] = mean, rstd                                      */
    float* out32;          /* fwd: optional fp32 output                                                    */
    void* out16;           /* fwd: optional output in compute type                                         */
    float p_pre, p_post;
    uint64_t seed, off_pre, off_post;
    const uint64_t* seed_dev;
    /* backward */
    const float* dout;     /* [rows,N] grad wrt out                                                        */
    float* dx32;           /* optional: grad wrt the pre-LN sum (= grad of residual / pos inputs)           */
    float* dxd32;          /* optional: grad wrt x (dropout_pre backward applied), fp32                    */
    void* dxd16;           /* optional: same in compute type (operand of the following dgrad/wgrad)        */
    float* dgamma; float* dbeta;   /* [N], accumulated (atomics; fixed order in deterministic mode)        */
    float* dbias;          /* optional [N]: column sums of the grad wrt x (bias grad of the producing GEMM) */
    float* dpos;           /* optional [period,N], accumulated (atomics; fixed order in deterministic mode)*/
    void* out16_lo;        /* fwd, optional, bf16: lo half of the output pair (UnivlGemm.A_lo of the products that read out16) */
} UnivlLayerNorm;
int univl_layernorm_fwd(const UnivlLayerNorm* d, hipStream_t stream);
int univl_layernorm_bwd(const UnivlLayerNorm* d, hipStream_t stream);

/* ------------------------------------------------------------------------------------------- attention
 * softmax(Q K^T / sqrt(d) + mask) -> dropout -> .V   (module_bert.py:181-197 and its three copies).  The additive
 * mask is built in-kernel from the {0,1} key mask exactly as module_bert.py:429-437 ((1-m)*-10000, added AFTER
 * the scaling) and, with `causal`, as module_decoder.py:389-396 (((1-answer_mask)+triu(1))>0)*-10000.
 * Head h of row r lives at  ptr + r*ld + h*64.  d is fixed to 64.  Forward saves the row log-sum-exp. */
typedef struct UnivlAttention {
    int32_t dtype;
    int32_t B, H, Sq, Sk;
    const void* q; int64_t ldq;
    const void* k; int64_t ldk;
    const void* v; int64_t ldv;
    const int64_t* key_mask;  /* optional [B,Sk], 1 = attend */
    int32_t causal;
    void* out; int64_t ldo;   /* [B*Sq, ldo] compute type */
    float* lse;               /* [B,H,Sq] */
    float p_drop; uint64_t seed, offset;
    const uint64_t* seed_dev;
    /* backward */
    const void* dout; int64_t lddo;
    void* dq; int64_t lddq;
    void* dk; int64_t lddk;
    void* dv; int64_t lddv;
    /* optional batch strides (elements) of k and v: 0 = Sk*ld (densely packed rows).  A key/value cache of capacity
     * Tmax >= Sk per sequence (incremental caption decoding, main_task_caption.py:434-470) passes Tmax*ld. */
    int64_t bsk, bsv;
    void* out_lo;             /* fwd, optional, bf16: lo half of the output pair, [B*Sq, ldo] (UnivlGemm.A_lo of the output projection) */
} UnivlAttention;
int univl_attention_fwd(const UnivlAttention* d, hipStream_t stream);
int univl_attention_bwd(const UnivlAttention* d, hipStream_t stream);

/* univl_attention_bwd with the dgrad of the attention-output projection (dctx = dY . W_o: the dense of BertSelfOutput, module_bert.py:207,
 * differentiated) computed INSIDE the launch: every workgroup first multiplies the 64 x 64 block of dctx that belongs to its own (batch
 * row, head) -- sequences of at most 64 positions, head width 64 -- and runs the attention backward on it from LDS; dctx never goes
 * through global memory (at->dout is not written) and one launch leaves the backward chain.  dq / dk / dv are bit-identical to
 * univl_gemm(odgrad) + univl_attention_bwd(at).  owgrad (optional): the weight gradient of the same projection (dY^T . ctx), riding
 * as extra workgroups like in univl_gemm_pair.  bf16 only; UNIVL_EUNSUPPORTED where the launch does not carry the pair; dry_run != 0
 * validates without launching. */
int univl_attention_bwd_fused(const UnivlAttention* at, const UnivlGemm* odgrad, const UnivlGemm* owgrad, int32_t dry_run, hipStream_t stream);

/* univl_attention_fwd with the query / key / value projection (module_bert.py:172-174) computed INSIDE the launch: every workgroup
 * multiplies the 64 x 192 block of q | k | v that belongs to its (batch row, head), stores it to the qkv buffer (bit-identical to
 * univl_gemm(qkv)) and attends on it from LDS -- self-attention over at most 128 positions (65 .. 128: two workgroups per (batch row,
 * head), one per block of 64 queries, each multiplying the whole sequence's q | k | v), bf16; the projection may carry operand pairs (A_lo / B_lo).  adam / chunk_*: BertAdam chunks
 * riding in the launch like in univl_gemm_rider (NULL / 0: none).  UNIVL_EUNSUPPORTED where the launch does not carry the pair. */
int univl_attention_fwd_fused(const UnivlAttention* at, const UnivlGemm* qkv, const struct UnivlAdam* adam, int32_t chunk_begin,
                              int32_t chunk_count, int32_t max_blocks, int32_t dry_run, hipStream_t stream);

/* ------------------------------------------------------------------------------------- text embeddings
 * BertEmbeddings / DecoderEmbeddings (module_bert.py:132-146, module_decoder.py:309-320):
 * gather word + position (+ token type) -> LayerNorm -> dropout.  Backward scatter-adds into the tables. */
typedef struct UnivlEmbedText {
    int32_t dtype;
    int32_t B, S, N;
    const int64_t* ids; const int64_t* type_ids;   /* type_ids optional */
    const float* word; const float* pos; const float* type;   /* type optional */
    const float* gamma; const float* beta; float eps;
    float* y; float* stats; float* out32; void* out16;
    float p_post; uint64_t seed, off_post;
    const uint64_t* seed_dev;
    const float* dout;
    float* dword; float* dpos; float* dtype_emb; float* dgamma; float* dbeta;   /* dpos may be NULL when drows is given */
    /* optional [B*S, N]: store each token's word-table gradient row here INSTEAD of scatter-adding it into dword.  Under
     * data parallelism the dense table gradient (30522 x 768 fp32 = 94 MB, at most B*S non-zero rows) is then exchanged
     * as (ids, rows) and rebuilt by univl_embed_scatter on every rank. */
    float* drows;
    void* out16_lo;           /* fwd, optional, bf16: lo half of the output pair */
} UnivlEmbedText;
int univl_embed_text_fwd(const UnivlEmbedText* d, hipStream_t stream);
int univl_embed_text_bwd(const UnivlEmbedText* d, hipStream_t stream);
/* Sparse bookkeeping of the word-embedding gradient (retrieval configurations: the token gather is the table's only
 * gradient source, so at most B*W of its 30522 rows are non-zero).  list[0 .. meta[0]) holds the rows written since the
 * last clear; meta[1] != 0 means "treat every row as listed" (list overflow, or a dense writer).  univl_rows_zero clears
 * the listed rows (instead of a 94 MB memset), univl_rows_append records the rows of a backward (reset != 0 starts a new
 * list), univl_rows_sumsq adds the sum of squares of the listed rows (each row once) to *out (instead of streaming the
 * whole table for its gradient norm).  cap <= 8192.  `ever` (optional, [rows_total] bytes): sticky "this row has been written at
 * least once" flags for UnivlAdam.row_flags -- univl_rows_append sets the flag of every id it is given (all flags on overflow). */
int univl_rows_zero(float* table, int64_t rows_total, const int64_t* list, const int32_t* meta, hipStream_t stream);
int univl_rows_append(const int64_t* ids, int32_t n, int64_t* list, int32_t cap, int32_t* meta, int32_t reset, uint8_t* ever,
                      int64_t rows_total, hipStream_t stream);
int univl_rows_sumsq(const float* table, int64_t rows_total, const int64_t* list, const int32_t* meta, float* out, hipStream_t stream);
/* out[s, c] += sum of rows[r, c] over r = s, s + period, s + 2 period, ... < n_rows (ascending, fixed order): the gradient of a
 * [period, n] position table from per-token gradient rows (UnivlEmbedText.drows, UnivlLayerNorm.dx32) -- used instead of the fused
 * kernels' scatter-add (dpos) from 32 rows per position on, where B atomics per table element dominate those kernels. */
int univl_rows_gather_sum(const float* rows, int32_t n_rows, int32_t period, int32_t n, float* out, hipStream_t stream);
/* dword[ids[t]] += scale * rows[t] for t < n (fp32 atomics; rows [n, 768]): the second half of the sparse exchange */
int univl_embed_scatter(const int64_t* ids, const float* rows, int64_t n, float scale, float* dword, hipStream_t stream);

/* ------------------------------------------------------------------------- pooling / similarity / loss
 * _mean_pooling_for_similarity + F.normalize (modeling.py:327-339, 385-388): masked mean over tokens
 * (skip_first: position 0 excluded for text; zero-count guard), optional L2 normalisation (eps 1e-12). */
typedef struct UnivlPool {
    int32_t B, S, N;
    const float* x; int64_t ldx_row;   /* x[(b*S+s)*ldx_row + c] */
    const int64_t* mask;
    int32_t skip_first, normalize;
    float* mean;        /* [B,N] saved pre-normalisation mean */
    float* out;         /* [B,N] */
    const float* dout;  /* bwd: [B,N] */
    float* dx;          /* bwd: [B,S,N] */
    int32_t accumulate; /* bwd: dx += instead of dx = */
    /* bwd, optional: take the upstream gradient from the similarity matrix's instead of `dout` -- torch.matmul(text, video.t())
     * of modeling.py:389 backward folded in: dout[b, :] = gscale[0] * sum_k dsim[b, k] * other[k, :] (transpose = 0: this side
     * indexes the ROWS of dsim) or sum_k dsim[k, b] * other[k, :] (transpose = 1: the columns); `other` = the other modality's
     * pooled (normalised) [n_other, N] matrix, dsim rows ldsim floats apart, gscale a device scalar (NULL: 1). */
    const float* dsim; int64_t ldsim; const float* other; int32_t n_other; int32_t transpose; const float* gscale;
} UnivlPool;
int univl_pool_fwd(const UnivlPool* d, hipStream_t stream);
int univl_pool_bwd(const UnivlPool* d, hipStream_t stream);
/* The text and the video pooling of one similarity head in ONE launch each way (the two descriptors of modeling.py:327-339). */
int univl_pool_pair_fwd(const UnivlPool* a, const UnivlPool* b, hipStream_t stream);
int univl_pool_pair_bwd(const UnivlPool* a, const UnivlPool* b, hipStream_t stream);

/* MaxMarginRankingLoss (until_module.py:245-251): loss = mean(w * (relu(m + x - diag_col) + relu(m + x - diag_row))).
 * `sim` and `dsim` are [n, ld] row-major (ld >= n).  Writes the scalar loss and d loss / d x (for an upstream
 * gradient of 1). */
int univl_maxmargin_loss(const float* sim, int32_t n, int32_t ld, float margin, const float* weight, float* loss,
                         float* dsim, hipStream_t stream);
/* CrossEn (until_module.py:186-191): mean(-diag(log_softmax(x, -1))) and its gradient. */
int univl_crossen_loss(const float* sim, int32_t n, int32_t ld, float* loss, float* dsim, hipStream_t stream);
/* MILNCELoss (until_module.py:201-221) for batch_size x n_pair blocks; n = batch_size*n_pair. */
int univl_milnce_loss(const float* sim, int32_t batch_size, int32_t n_pair, int32_t ld, float* loss, float* dsim,
                      hipStream_t stream);
/* Retrieval metrics (metrics.py:8-20): for every row i of the [n, ld] similarity matrix, gt[i] = #{j : x[i][j] > x[i][i]}
 * and eq[i] = #{j : x[i][j] == x[i][i]} (>= 1).  The rank positions `ind` of compute_metrics are range(gt, gt + eq)
 * per row: R@k / median rank follow on the host from 2n integers instead of an n x n sort. */
int univl_rank_counts(const float* sim, int32_t n, int64_t ld, int32_t* gt, int32_t* eq, hipStream_t stream);
/* x[0..n) *= s[0] with s on the device (applies loss.backward()'s upstream gradient without a host sync) */
int univl_scale_by_device_scalar(float* x, int64_t n, const float* s, hipStream_t stream);

/* --------------------------------------------------------------------- cross encoder / classifier helpers
 * pair_concat: out[p, :, :] = cat(seq[tidx[p]], vis[vidx[p]]) and out_mask likewise -- torch.cat of
 * UniVL._get_cross_output (modeling.py:315-325) for an explicit list of (text row, video row) pairs (the repeat/view
 * expansion of _cross_similarity modeling.py:352-366, or tidx = vidx = arange for the caption / pretrain path).
 * Backward scatter-adds d out into dseq / dvis (fp32 atomics; callers zero or pre-fill them). */
int univl_pair_concat_fwd(const float* seq, const float* vis, const int64_t* amask, const int64_t* vmask,
                          const int32_t* tidx, const int32_t* vidx, int32_t P, int32_t W, int32_t F, float* out,
                          int64_t* out_mask, hipStream_t stream);
int univl_pair_concat_bwd(const float* dout, const int32_t* tidx, const int32_t* vidx, int32_t P, int32_t W, int32_t F,
                          float* dseq, float* dvis, hipStream_t stream);
/* CrossEmbeddings (module_cross.py:123-138): table[s] = position[s] + token_type[s >= W] for s < S = W + F; the
 * LayerNorm kernel then adds it with period S.  Backward accumulates into dpos / dtype with atomics. */
int univl_postype_fwd(const float* pos, const float* type, int32_t W, int32_t S, float* out, hipStream_t stream);
int univl_postype_bwd(const float* dtable, int32_t W, int32_t S, float* dpos, float* dtype, hipStream_t stream);
/* nn.Tanh of CrossPooler (module_cross.py:281-287) */
int univl_tanh_fwd(const float* x, float* y, int64_t n, hipStream_t stream);
int univl_tanh_bwd(int32_t dtype, const float* dy, const float* y, void* dx, int64_t n, hipStream_t stream);   /* dx in compute type */
/* du (compute type) = dg (fp32) * gelu'(u): BertPredictionHeadTransform backward (module_bert.py:299-311) */
int univl_gelu_bwd(int32_t dtype, const float* dg, const void* u, void* du, int64_t n, hipStream_t stream);
/* similarity_dense = nn.Linear(768, 1) on the pooled cross output (modeling.py:167,371) and its backward
 * (dx written, dw/db accumulated with atomics) */
/* out[c] += sum_r x[r, c] over a [rows, ld] matrix in the compute type (atomics) */
int univl_colsum(int32_t dtype, const void* x, int64_t ld, int32_t rows, int32_t n, float* out, hipStream_t stream);
/* dst[r, 0:copy_bytes) = src[idx[r], 0:copy_bytes) for r < rows; rows are row_stride bytes apart in both buffers, all
 * byte counts multiples of 16 (beam re-ordering of the decoder's key/value cache: Beam.get_current_origin, beam.py:54) */
int univl_gather_rows(const void* src, void* dst, const int32_t* idx, int32_t rows, int64_t row_stride, int64_t copy_bytes,
                      hipStream_t stream);
/* x[r, 0:n) <- log_softmax(x[r, 0:n)) in place, fp32 rows ld apart (torch.nn.functional.log_softmax of
 * main_task_caption.py:454 over the vocabulary) */
int univl_log_softmax_rows(float* x, int32_t rows, int32_t n, int64_t ld, hipStream_t stream);
/* out[seg[e]] = sum(partials[start[e] .. start[e] + count[e])) for e < n: folds the per-wave partial sums written by the
 * weight-gradient GEMMs (UnivlGemm.sumsq) into the per-tensor sums of squares */
int univl_sumsq_finish(const float* partials, const int32_t* seg, const int32_t* start, const int32_t* count, int32_t n,
                       float* out, hipStream_t stream);
/* x (compute type) *= s[0], s on the device */
int univl_scale_ct_by_device_scalar(int32_t dtype, void* x, int64_t n, const float* s, hipStream_t stream);
int univl_simdense_fwd(const float* x, const float* w, const float* b, int32_t rows, float* out, hipStream_t stream);
int univl_simdense_bwd(const float* ds, const float* x, const float* w, int32_t rows, float* dx, float* dw, float* db,
                       hipStream_t stream);
/* CrossEntropyLoss(ignore_index) over [rows, V] fp32 logits (modeling.py:168,252-254,275): loss = mean over rows with
 * label != ignore_index; dlogits (compute type, [rows, lddl]) = (softmax - onehot) / n_valid.  scratch2: 2 floats. */
int univl_ce_loss(int32_t dtype, const float* logits, int64_t ld, const int64_t* labels, int32_t rows, int32_t V,
                  int32_t ignore_index, float* scratch2, float* loss, void* dlogits, int64_t lddl, hipStream_t stream);
/* K16: the tied vocabulary classifier fused with an ONLINE log-softmax cross entropy -- BertLMPredictionHead.forward's last line
 * (module_bert.py:327-330: h . E_word^T + bias; decoder copy module_decoder.py:180-183) and CrossEntropyLoss(ignore_index) on it
 * (modeling.py:253, 275) without the [rows, V] logits ever existing in memory.
 *   univl_vocab_ce_fwd: walks the product in 128 x 128 tiles; a tile's epilogue leaves one (max, sum exp(logit - max)) pair per row in
 *     partial[row][column tile] and the label's logit in label_logit[row]; two small kernels fold them (fixed order: bit-reproducible)
 *     into lse[rows], rowloss[rows], scratch2[0] = number of rows whose label != ignore_index, loss = mean over those rows (NaN if none).
 *   univl_vocab_ce_bwd: the same product again; the epilogue stores dlogits[row, col] = (exp(logit - lse[row]) - [col == label]) *
 *     gout / n_valid in the compute type (0 on ignored rows) -- the operand of the two backward products (dh = dlogits . E,
 *     dE += dlogits^T . h), which are ordinary univl_gemm calls.  Call after univl_vocab_ce_fwd with the same descriptor.
 * x [rows, K] and table [V, K] in the compute type, K-major, K a multiple of 64 (bf16) / 32 (fp32); slots >= ceil(V / 128);
 * partial holds rows * slots * 2 floats; dlogits rows are lddl >= V elements apart (columns >= V are left untouched). */
typedef struct UnivlVocabCE {
    int32_t dtype, rows, V, K;
    const void* x; int64_t ldx;
    const void* table; int64_t ldt;
    const float* bias;             /* [V] or null */
    const int64_t* labels;         /* [rows] */
    int32_t ignore_index, slots;
    float* partial;                /* [rows, slots, 2] */
    float* label_logit;            /* [rows] */
    float* lse;                    /* [rows] */
    float* rowloss;                /* [rows] */
    float* scratch2;               /* [0] n_valid, [1] sum of the row losses */
    float* loss;                   /* [1] */
    const float* gout;             /* backward: upstream gradient of the loss, one float on the device; null = 1 */
    void* dlogits; int64_t lddl;   /* backward output */
} UnivlVocabCE;
int univl_vocab_ce_fwd(const UnivlVocabCE* desc, hipStream_t stream);
int univl_vocab_ce_bwd(const UnivlVocabCE* desc, hipStream_t stream);
/* masked-frame NCE of UniVL._calculate_mfm_loss (modeling.py:285-297) on the [n,n] logits matrix */
int univl_mfm_nce_loss(const float* logits, int64_t ld, const int64_t* vmask, const int64_t* labels, int32_t n,
                       float* scratch2, float* loss, float* dlogits, int64_t lddl, hipStream_t stream);

/* -------------------------------------------------------------------------------------------- optimizer
 * Fused multi-tensor BertAdam (modules/optimization.py:103-168) + clip_grad_norm_ (main_task_retrieval.py:347)
 * over FLAT parameter / gradient / moment buffers.  `segs` describes the parameter tensors (device array). */
typedef struct UnivlSeg {
    int64_t offset, numel;      /* element range in the flat buffers                                     */
    float lr, weight_decay;     /* group hyper-parameters (main_task_retrieval.py:185-190)               */
    float max_grad_norm;        /* per-parameter clip of optimization.py:135-136 (<=0: off)              */
    int32_t active;             /* 0: parameter has no gradient this step (skipped, as `p.grad is None`) */
} UnivlSeg;
/* sumsq[s] = sum of squares of segment s of g (fp32 atomics; sumsq must be zeroed by the caller). */
int univl_grad_sumsq(const float* g, const UnivlSeg* segs, int32_t nseg, const int32_t* chunk_seg,
                     const int64_t* chunk_off, const int32_t* chunk_len, int32_t nchunk, float* sumsq,
                     hipStream_t stream);
/* total-norm clip coefficient: coef[0] = min(1, max_norm / (sqrt(sum_active sumsq) + 1e-6)), coef[1] = total norm */
int univl_clip_coef(const float* sumsq, const UnivlSeg* segs, int32_t nseg, float max_norm, float* coef, hipStream_t stream);
/* g *= coef[0] over all active segments (in-place form of clip_grad_norm_) */
int univl_scale_grads(float* g, const UnivlSeg* segs, const int32_t* chunk_seg, const int64_t* chunk_off,
                      const int32_t* chunk_len, int32_t nchunk, const float* coef, hipStream_t stream);
typedef struct UnivlAdam {
    float* p; const float* g; float* m; float* v;   /* flat fp32 buffers                                  */
    void* p16;                   /* optional bf16 shadow of p (same offsets), rewritten by the step       */
    const UnivlSeg* segs; int32_t nseg;
    const int32_t* chunk_seg; const int64_t* chunk_off; const int32_t* chunk_len; int32_t nchunk;
    const float* sumsq;          /* per-segment sum of squares of the UNSCALED g                          */
    const float* coef;           /* optional global clip coefficient still to be applied (deferred clip)  */
    int32_t* step;               /* [nseg] per-parameter step counters (state['step'])                    */
    float b1, b2, eps;
    float warmup; int32_t t_total;   /* warmup_linear schedule (optimization.py:38-43), t_total -1: constant */
    float* seg_scalars;          /* scratch [nseg*2]                                                       */
    int32_t schedule;            /* 0 warmup_linear, 1 warmup_cosine, 2 warmup_constant (optimization.py:26-50) */
    /* Optional "rows nobody ever touched" shortcut for ONE row-structured tensor (the 30522 x 768 word table: at most B*W of its
     * rows receive a gradient per step, 15 % of all parameters).  row_flags[r] != 0 <=> row r of segment flag_seg has ever held a
     * non-zero gradient, moment or second moment (kept by univl_rows_append; the host sets every flag where it cannot know).  A
     * chunk of that segment whose rows are all unflagged has g = m = v = 0 exactly, so its BertAdam update is p -= lr * (wd * p)
     * -- the same bits the full formula gives (0 / (sqrt(0) + e) = 0) at 10 instead of 30 bytes per parameter.  The segment's
     * chunks must start on row boundaries.  NULL: off. */
    const uint8_t* row_flags; int32_t flag_seg; int32_t row_len;
    void* p16_lo;                /* optional: lo half of the shadow pair, bf16(p - bf16(p)), same offsets (UnivlGemm.B_lo)   */
} UnivlAdam;
int univl_bert_adam(const UnivlAdam* d, hipStream_t stream);
/* The same update over chunks [chunk_begin, chunk_begin + chunk_count) of the chunk table only; do_prep != 0 first runs the
 * per-tensor scalar kernel (once per step, before the first range).  max_blocks > 0 caps the grid (the kernel then walks
 * the range), which leaves compute units to kernels running concurrently on other streams: the pipelined training step
 * applies the update layer by layer next to the following forward pass (univl_amd/graphed.py). */
int univl_bert_adam_range(const UnivlAdam* d, int32_t chunk_begin, int32_t chunk_count, int32_t do_prep, int32_t max_blocks,
                          hipStream_t stream);
/* (Round 3: validated bit-exact against the sequential loop and the default of the captured step.)  A forward product (both operands K-major, bf16, 64 x 64 tiles) and chunks [chunk_begin, +chunk_count) of a
 * prepared BertAdam update (univl_bert_adam_range with do_prep ran before) in ONE launch: the update of step t rides with the
 * forward of step t + 1 (univl_amd/graphed.py).  Products the kernel does not carry are launched the ordinary way, followed by
 * the chunk range as its own launch -- same result.  max_blocks > 0 caps the workgroups given to the update. */
int univl_gemm_rider(const UnivlGemm* gemm, const UnivlAdam* adam, int32_t chunk_begin, int32_t chunk_count, int32_t max_blocks,
                     hipStream_t stream);
/* Round 5: the launch also carries products on the 64 x 128 tile (what univl_gemm picks from 1536 rows on).  Host-side question, no
 * device work: 1 if univl_gemm_rider carries chunks INSIDE this product's launch, 0 if it would enqueue the update behind the product
 * (the 128 x 128 and 256 x 256 tiles, transposed operands, fp32); negative: the descriptor's validation error.  A host spreads a
 * layer's chunks over the products that answer 1 (engine.EncoderStack.build_forward). */
int univl_gemm_rider_fits(const UnivlGemm* gemm);
/* K8 / K10 of the survey (module_bert.py:207-211, 246-250: dense -> dropout -> + input -> LayerNorm; the copies in module_visual.py /
 * module_cross.py): a forward product whose fp32 output is the input x of a LayerNorm, with that LayerNorm finished INSIDE the product's
 * launch -- each 64-row block of the output is normalised by the last workgroups to contribute to it (agent-scope release / acquire around
 * a per-block arrival counter), instead of by a second launch behind a kernel boundary.  Same arithmetic per row as univl_layernorm_fwd
 * (the same device function); the product's sums meet in hardware order exactly as in any split-K product, so the entry point refuses
 * deterministic mode.  Optionally also carries BertAdam chunks like univl_gemm_rider (adam may be NULL with chunk_count 0).
 *   gemm: bf16, both operands K-major, fp32 output C32 == ln->x with ldc = N = 768, no C16 / GELU / ACCUM / sumsq, at most 1024 rows
 *         (16 row blocks: the fold's last arrivals spin for the block's later ones, and the spinners must stay far below the resident slots);
 *   ln:   rows == gemm->M, N == 768, fp32 x, dtype bf16 (residual / dropout / y / stats / out32 / out16 as for univl_layernorm_fwd);
 *   counters: 2 * ceil(M / 64) int32 device words owned by this call site, ZERO before its first launch (the launch leaves them zero).
 * Returns UNIVL_EUNSUPPORTED for anything else (callers then enqueue univl_gemm / univl_gemm_rider + univl_layernorm_fwd);
 * dry_run != 0 validates without launching. */
int univl_gemm_ln(const UnivlGemm* gemm, const UnivlLayerNorm* ln, int32_t* counters, const UnivlAdam* adam, int32_t chunk_begin,
                  int32_t chunk_count, int32_t max_blocks, int32_t dry_run, hipStream_t stream);
/* The backward twin: univl_gemm_pair whose dgrad product's fp32 output is the upstream gradient of a LayerNorm backward (ln->dout ==
 * dgrad->C32, e.g. the FFN1 dgrad in front of BertSelfOutput's LayerNorm, module_bert.py:207-211 differentiated), with that LayerNorm
 * backward (dx32 / dxd16 rows, dgamma / dbeta / dbias column sums) finished inside the launch by the dgrad's last workgroups per 64-row
 * block.  At most 1024 rows (the rectangular dgrad body of 384+ rows only beside a 128 x 64 weight-gradient body), N = 768, C32 pre-zeroed (the dgrad's contributions become fp32 atomics), no position
 * rows (ln->dpos NULL); counters as for univl_gemm_ln; UNIVL_EUNSUPPORTED otherwise and in deterministic mode. */
int univl_gemm_pair_ln(const UnivlGemm* dgrad, const UnivlGemm* wgrad, const UnivlLayerNorm* ln, int32_t* counters, int32_t dry_run,
                       hipStream_t stream);
/* Large-LDS opt-in of the rider kernels on the stream's device; call once outside any stream capture before the first captured rider. */
int univl_gemm_rider_prime(hipStream_t stream);
/* *ctr += 1 (device word; used for per-replay dropout seeds) */
int univl_bump_counter(uint64_t* ctr, hipStream_t stream);
/* p16 <- bf16(p) over n elements */
int univl_cast_bf16(const float* p, void* p16, int64_t n, hipStream_t stream);
/* the shadow PAIR: p16 <- bf16(p), p16_lo <- bf16(p - p16) over n elements (p16 may be NULL: only the lo half is written) */
int univl_cast_bf16_pair(const float* p, void* p16, void* p16_lo, int64_t n, hipStream_t stream);
/* p <- fp32(p16) over n elements (the bf16 gradient exchange brings its buckets back into the fp32 gradient buffer) */
int univl_cast_f32(const void* p16, float* p, int64_t n, hipStream_t stream);

/* ------------------------------------------------------------------------------------- hardware probes */
int univl_probe_layouts(float* out, int32_t n_out, hipStream_t stream);
/* Measurement node: *out = the device's constant-rate wall clock (wall_clock64, 100 MHz on gfx950) when a one-thread kernel enqueued on
 * `stream` runs.  Placed between the nodes of a captured step it gives GPU-side start / end times of the step's branches WITHOUT a
 * profiler attached (scripts/probe_branches.py: rocprofv3 changes how the two branches of the step graph overlap). */
int univl_stamp(uint64_t* out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
