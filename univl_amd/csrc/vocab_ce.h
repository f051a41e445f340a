// K16: the tied vocabulary classifier fused with an ONLINE log-softmax cross entropy -- in training the [tokens, 30522]
// logits never exist in memory.  Included by gemm.hip (needs Tile / tile_mma / gemm_acc_only / pair_tile / smem_raw).
//
// Reference: BertLMPredictionHead.forward (module_bert.py:327-330: h . E_word^T + bias, E_word the tied 30522 x 768 table), its decoder
// copy (module_decoder.py:180-183), and CrossEntropyLoss(ignore_index=-1) on the result (modeling.py:253, 275).
//
// Forward  (vocab_ce_kernel<T, false>): the product is walked in 128 x 128 tiles (the K loop of gemm_tile, gemm_acc_only); a tile's epilogue
//   reduces its logits to one (max, sum of exp(logit - max)) pair per row -- 16-lane shuffles inside a wave, LDS across the waves of the
//   column direction -- and writes it to partial[row][column tile]; the one lane that owns (row, label[row]) stores that logit.
//   vocab_ce_rows_kernel then folds a row's 239 pairs into its log-sum-exp (a wave per row, fixed order) and vocab_ce_loss_kernel sums
//   lse - label logit over the rows that count (one workgroup, fixed order: the loss is bit-reproducible in every mode).
// Backward (vocab_ce_kernel<T, true>): the same product again; the epilogue turns a logit into
//   (exp(logit - lse[row]) - [col == label[row]]) * gout / n_valid   (0 on ignored rows)
//   and stores it in the compute type: the matrix autograd calls dlogits, which the two existing backward products consume
//   (dh = dlogits . E, dE += dlogits^T . h).  What K16 removes per step at 512 tokens: the 62 MB fp32 logits (written by the product, read
//   by the loss kernel), the loss kernel's 31 MB dlogits write + the scale pass over them; what it adds: one more product.
#pragma once

struct VocabCeArgs {
    const void* X; long ldx;            // [rows, K] head output (compute type), K-major
    const void* E; long lde;            // [V, K] tied word table (compute type), K-major
    const float* bias;                  // [V] or null
    int rows, V, K;
    const int64_t* labels; int ignore;
    float* partial; int slots;          // [rows, slots, 2]
    float* label_logit;                 // [rows]
    const float* lse;                   // [rows]            (backward)
    const float* scal;                  // scal[0] = n_valid (backward)
    const float* gout;                  // device scalar or null (= 1)
    void* dl; long lddl;                // [rows, lddl] compute type (backward)
    int nx, ny;
};

template <typename T, bool BWD, int WGN>
__global__ __launch_bounds__(128 * WGN, 2) void vocab_ce_kernel(VocabCeArgs a) {
    constexpr int BM = 128, BN = 128, WGM = 2, NC = 2;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    int bx, by, bz;
    pair_tile((int)blockIdx.x, a.nx * a.ny, a.nx, a.ny, 1, UNIVL_GEMM_XCD_MAP, 8, bx, by, bz);
    const int m0 = by * BM, n0 = bx * BN;
    f32x4_t acc[MI][NI];
    gemm_acc_only<T, false, false, BM, BN, NC, WGM, WGN>(reinterpret_cast<const T*>(a.X), a.ldx, reinterpret_cast<const T*>(a.E), a.lde, a.rows, a.V,
                                                         a.K, m0, n0, acc, smem_raw);
    // (the K loop ends with a barrier: the stages are free)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN, wn = wave % WGN;
    int col[NI];
    bool vcol[NI];
    float bv[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        col[b] = n0 + wn0 + 16 * b + i;
        vcol[b] = col[b] < a.V;
        bv[b] = (a.bias && vcol[b]) ? a.bias[col[b]] : 0.0f;
    }
    if constexpr (!BWD) {
        float* red = reinterpret_cast<float*>(smem_raw);          // [WGN][BM][2]
#pragma unroll
        for (int ma = 0; ma < MI; ++ma)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm0 + 16 * ma + 4 * g + r, row = m0 + rl;
                const long lab = row < a.rows ? (long)a.labels[row] : -2;
                float v[NI], m = -INFINITY;
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    v[b] = vcol[b] ? acc[ma][b][r] + bv[b] : -INFINITY;
                    m = fmaxf(m, v[b]);
                    if (vcol[b] && (long)col[b] == lab) a.label_logit[row] = v[b];
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                float s = 0.0f;
                if (m > -INFINITY) {
#pragma unroll
                    for (int b = 0; b < NI; ++b) s += __expf(v[b] - m);     // exp(-inf) = 0 for the columns beyond V
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
                if (i == 0) { red[(wn * BM + rl) * 2] = m; red[(wn * BM + rl) * 2 + 1] = s; }
            }
        __syncthreads();
        if (tid < BM && m0 + tid < a.rows) {
            float m = -INFINITY;
#pragma unroll
            for (int w = 0; w < WGN; ++w) m = fmaxf(m, red[(w * BM + tid) * 2]);
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < WGN; ++w) {
                const float mw = red[(w * BM + tid) * 2];
                if (mw > -INFINITY) s += red[(w * BM + tid) * 2 + 1] * __expf(mw - m);
            }
            float* out = a.partial + ((long)(m0 + tid) * a.slots + bx) * 2;
            out[0] = m;
            out[1] = s;
        }
    } else {
        const float nvalid = a.scal[0];
        const float up = a.gout ? a.gout[0] : 1.0f;
        T* dl = reinterpret_cast<T*>(a.dl);
#pragma unroll
        for (int ma = 0; ma < MI; ++ma)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm0 + 16 * ma + 4 * g + r;
                if (row >= a.rows) continue;
                const long lab = (long)a.labels[row];
                const bool counts = lab != (long)a.ignore && nvalid > 0.0f;
                const float sc = counts ? up / nvalid : 0.0f;
                const float l = counts ? a.lse[row] : 0.0f;
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    if (!vcol[b]) continue;
                    const float p = counts ? __expf(acc[ma][b][r] + bv[b] - l) : 0.0f;
                    dl[(long)row * a.lddl + col[b]] = from_f32<T>((p - ((long)col[b] == lab ? 1.0f : 0.0f)) * sc);
                }
            }
    }
}

// a wave per row: fold the row's `slots` (max, sum) pairs in slot order -> lse[row]; rowloss[row] = lse - label logit (0 on ignored rows)
__global__ __launch_bounds__(256) void vocab_ce_rows_kernel(const float* partial, int pitch, int slots, const float* label_logit, const int64_t* labels,
                                                            int ignore, int rows, float* lse, float* rowloss) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = partial + (long)row * pitch * 2;          // `slots` written column tiles of `pitch` reserved ones
    float m = -INFINITY;
    for (int j = lane; j < slots; j += 64) m = fmaxf(m, p[2 * j]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.0f;
    for (int j = lane; j < slots; j += 64) {
        const float mj = p[2 * j];
        if (mj > -INFINITY) s += p[2 * j + 1] * expf(mj - m);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        const float l = m + logf(s);
        lse[row] = l;
        rowloss[row] = ((long)labels[row] != (long)ignore) ? l - label_logit[row] : 0.0f;
    }
}

__device__ __forceinline__ float vocab_block_sum(float v, float* red) {      // 256 threads, fixed order
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: n_valid and the loss, both summed in row order (CrossEntropyLoss: mean over the rows that count, NaN if none does)
__global__ __launch_bounds__(256) void vocab_ce_loss_kernel(const float* rowloss, const int64_t* labels, int ignore, int rows, float* scal, float* loss) {
    __shared__ float red[4];
    float c = 0.0f, t = 0.0f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        c += ((long)labels[r] != (long)ignore) ? 1.0f : 0.0f;
        t += rowloss[r];
    }
    c = vocab_block_sum(c, red);
    t = vocab_block_sum(t, red);
    if (threadIdx.x == 0) {
        scal[0] = c;
        scal[1] = t;
        loss[0] = c > 0.0f ? t / c : NAN;
    }
}

template <typename T, int WGN>
static int vocab_ce_launch(const VocabCeArgs& a, bool bwd, hipStream_t stream) {
    constexpr int NC = 2, BK = NC * Mma<T>::CH, NT = 128 * WGN;
    constexpr size_t smem_k = 2 * (Tile<T, false, 128, BK, NT>::BYTES + Tile<T, false, 128, BK, NT>::BYTES);
    constexpr size_t smem = smem_k > (size_t)WGN * 128 * 2 * sizeof(float) ? smem_k : (size_t)WGN * 128 * 2 * sizeof(float);
    static bool done_f[UNIVL_MAX_DEVICES] = {}, done_b[UNIVL_MAX_DEVICES] = {};
    if (bwd) {
        if (smem > 48 * 1024) univl_allow_lds(vocab_ce_kernel<T, true, WGN>, smem, done_b);
        hipLaunchKernelGGL((vocab_ce_kernel<T, true, WGN>), dim3(a.nx * a.ny), dim3(NT), smem, stream, a);
    } else {
        if (smem > 48 * 1024) univl_allow_lds(vocab_ce_kernel<T, false, WGN>, smem, done_f);
        hipLaunchKernelGGL((vocab_ce_kernel<T, false, WGN>), dim3(a.nx * a.ny), dim3(NT), smem, stream, a);
    }
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
