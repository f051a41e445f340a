// Common device/host helpers for the UniVL gfx950 (MI355X, CDNA4) kernels.
//
// Conventions used by every kernel in this directory
//   * wave = 64 lanes; a workgroup is 256 threads = 4 waves unless stated otherwise.
//   * lane decomposition for 16x16 MFMA tiles:  i = lane & 15 ("row/col within the tile"), g = lane >> 4.
//   * compute type T is __bf16 (production, MFMA 16x16x32 bf16, fp32 accumulate) or float (parity mode, MFMA
//     16x16x4 f32 -- exact fp32 FMA chains).  Both are hidden behind Mma<T> below.
//   * CONTRACTION-SLOT MAP.  One "chunk" is CH contraction indices (32 for bf16, 16 for f32).  Lane (i, g)
//     owns the slots   bf16: kk(g,j) = j<4 ? 4g+j : 16+4g+(j-4)   (j = 0..7)
//                      f32 : kk(g,r) = 4g+r                        (r = 0..3; r selects one of the 4 MFMAs)
//     for BOTH operands.  Because A and B always use the same map, the hardware's own k-assignment inside one
//     MFMA is irrelevant; and the map is chosen so that a 16x16 accumulator tile (C layout: col = i,
//     row = 4g+reg) IS ALREADY an operand fragment for a following MFMA that contracts over the tile's rows
//     (two stacked tiles for bf16, one for f32) -- attention feeds softmax(P) to the PV product that way with no
//     shuffles, no LDS round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// 16-byte staging register.  A NATIVE vector, not HIP's uint4 struct: arrays of the struct type handed through
// inlined helpers were demoted to scratch memory by hipcc (ROCm 7.2), arrays of native vectors stay in VGPRs.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define UNIVL_WAVE 64

// ------------------------------------------------------------------------------------------- error handling
// Status codes of the C ABI: 0 ok, <0 argument error, >0 hipError_t.
enum { UNIVL_OK = 0, UNIVL_EINVAL = -1, UNIVL_EALIGN = -2, UNIVL_EUNSUPPORTED = -3 };
void univl_set_error(const char* fmt, ...);
#define UNIVL_CHECK_ARG(cond, code, ...)            \
    do {                                            \
        if (!(cond)) {                              \
            univl_set_error(__VA_ARGS__);           \
            return (code);                          \
        }                                           \
    } while (0)
#define UNIVL_LAUNCH_CHECK()                                              \
    do {                                                                  \
        hipError_t e__ = hipGetLastError();                               \
        if (e__ != hipSuccess) {                                          \
            univl_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                              \
        }                                                                 \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------------------------------- devices
// Entry points take the DEVICE FROM THEIR STREAM ARGUMENT (hipStreamGetDevice), never from the calling thread's
// "current device": the reference's evaluation fan-out (util.py:21-60) calls into the model from one thread per
// GPU, and autograd's backward threads have their own current device.  The null stream means the current device.
#define UNIVL_MAX_DEVICES 64
struct UnivlStreamDevice {
    int prev = -1, dev = -1;
    explicit UnivlStreamDevice(hipStream_t s) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        dev = prev;
        hipDevice_t d;
        if (s != nullptr && hipStreamGetDevice(s, &d) == hipSuccess && (int)d != prev) {
            dev = (int)d;
            (void)hipSetDevice(dev);
        } else {
            prev = -1;                       // nothing to restore
        }
    }
    ~UnivlStreamDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define UNIVL_ON_STREAM_DEVICE(stream) UnivlStreamDevice univl_dev_guard__(stream)

// Kernels that need more than 48 KB of dynamic LDS opt in ONCE PER DEVICE (the attribute is per device and per function).
template <typename K>
static inline void univl_allow_lds(K kernel, size_t bytes, bool (&done)[UNIVL_MAX_DEVICES]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const bool tracked = dev >= 0 && dev < UNIVL_MAX_DEVICES;
    if (tracked && done[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (tracked) done[dev] = true;
}

// ------------------------------------------------------------------------------------------- deterministic mode
// univl_set_deterministic(1) (C ABI; UNIVL_DETERMINISTIC=1 on the Python side): every floating-point sum whose ORDER the default
// kernels leave to the hardware (fp32 atomics: split-K, column sums of LayerNorm / bias gradients, embedding scatter-adds,
// gradient-norm partials, loss accumulators) is taken in a FIXED order instead, so that two runs on the same inputs are
// bit-identical.  It is the mode the parity tests run in; the production default keeps the atomics (fire-and-forget, no second pass).
// Column sums use "the last block to arrive reduces": every block publishes its partial vector in a scratch slice, bumps a counter,
// and the block that sees the final count adds the partials in block order.  Scatter-adds become gather-style follow-up kernels
// ("first occurrence of a destination sums all its sources in source order").  Scratch comes from a per-device ring owned by the
// library (allocated by univl_set_deterministic, never on the hot path of the default mode).
bool univl_deterministic();
void* univl_det_alloc(size_t bytes);      // 256-byte aligned slice of the current device's ring; nullptr (+ error string) if unavailable
int* univl_det_counter();                 // a zero-initialised device int; its user leaves it at zero again

#ifdef __HIPCC__
// Call from ALL threads of a block after the block's partials were stored: true in exactly one block, the last one to arrive, with
// every other block's partials visible to it.  The counter is reset by the caller's last block (det_reset).
__device__ __forceinline__ bool det_last_block(int* counter, int nblocks, int* flag_lds) {
    __threadfence();                       // agent-scope release of this thread's partial stores (write-back of this XCD's L2)
    __syncthreads();
    if (threadIdx.x == 0) *flag_lds = (atomicAdd(counter, 1) == nblocks - 1) ? 1 : 0;
    __syncthreads();
    const bool last = *flag_lds != 0;
    if (last) {
        __threadfence();                   // agent-scope acquire: the other XCDs' partials are read from memory, not from a stale L2
        if (threadIdx.x == 0) *counter = 0;   // ready for the next launch / graph replay that uses this counter
    }
    return last;
}
#endif

// ------------------------------------------------------------------------------------------- small helpers
__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }   // RNE on gfx950 (v_cvt_pk_bf16_f32)

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU of the reference (until_module.py:28-33) and its derivative.
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// erf-GELU (until_module.py:28-33) and its derivative for this body's epilogue.  libm's erff is two divergent branches of ~35 VALU
// instructions each -- at 128 results per lane and ONE workgroup per compute unit (nothing else to overlap with) that is ~15 us per
// 256 x 256 tile, longer than a 12-tile K loop.  Here: Abramowitz-Stegun 7.1.26, branch-free, |error| <= 1.5e-7 in erf (fp32
// round-off class; the results are rounded to bf16 = 4e-3 next):  with z = |x| / sqrt 2, t = 1 / (1 + p z), E = exp(-z^2) = exp(-x^2 / 2),
// q = poly(t) E:   Phi(x) = 0.5 (1 + erf(x / sqrt 2)) = 1 - q / 2  (x >= 0),  q / 2  (x < 0)   -- no cancellation in the negative tail;
// gelu = x Phi,  gelu' = Phi + x E / sqrt(2 pi)  (the same exponential).
__device__ __forceinline__ void g256_phi(float x, float& phi, float& E) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    E = __expf(-z * z);
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(t, q, 1.421413741f);
    q = fmaf(t, q, -0.284496736f);
    q = fmaf(t, q, 0.254829592f);
    q = q * t * E * 0.5f;
    phi = x >= 0.0f ? 1.0f - q : q;
}
__device__ __forceinline__ float g256_gelu(float x) { float phi, E; g256_phi(x, phi, E); return x * phi; }
__device__ __forceinline__ float g256_gelu_grad(float x) { float phi, E; g256_phi(x, phi, E); return fmaf(x * 0.39894228040143267794f, E, phi); }

// Counter-based RNG for dropout: one 32-bit draw per (seed, stream offset, element index); the same function
// regenerates the mask in the backward pass.  (No bit-parity with torch's Philox stream is possible or required:
// parity is checked at p = 0)
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t offset, uint64_t idx, float p, float inv_keep) {
    // returns 0 (dropped) or 1/(1-p) (kept)
    uint32_t r = mix32((seed * 0x9E3779B97F4A7C15ULL) ^ (offset * 0xD1B54A32D192ED03ULL + idx));
    return ((float)(r >> 8) * (1.0f / 16777216.0f)) < p ? 0.0f : inv_keep;
}

// ------------------------------------------------------------------------------------------- MMA policy
template <typename T> struct Mma;

template <> struct Mma<__bf16> {
    static constexpr int CH = 32;            // contraction indices per chunk
    static constexpr int EPC = 8;            // elements per 16-byte vector
    typedef bf16x8_t frag;
    // LDS pitches (elements).  K-major tile [rows][BK]: +16 B pad -> the two 8-byte reads of the 16 rows of a
    // fragment hit 16 distinct 4-dword bank groups.  T-major tile [BK][rows]: +32 B pad -> the eight 32-byte row
    // segments one half-wave transposes-reads hit distinct banks.
    static constexpr int kpad = 8, tpad = 16;
    __device__ static __forceinline__ f32x4_t mma(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // K-major: `p` points at element (tile row of this lane, chunk start k0) in LDS.
    __device__ static __forceinline__ frag lds_kmajor(const __bf16* p, int g) {
        bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(p + 4 * g);
        bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(p + 16 + 4 * g);
        frag f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
    // same, straight from global memory (loop-invariant operands held in registers).  Unconditional: callers clamp
    // the row so that the address is always valid (a branch around a load costs a full vmcnt(0) drain).
    __device__ static __forceinline__ frag gmem_kmajor(const __bf16* p, int g) {
        bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(p + 4 * g);
        bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(p + 16 + 4 * g);
        frag f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
    // T-major: `base` points at element (contraction row k0, tile column 0 of the 16 columns) in LDS; `pitch`
    // in elements.  ds_read_b64_tr_b16: within a 16-lane group the lanes supply the sixteen 8-byte pieces of a
    // 4(row) x 16(col) block (lane i -> row i>>2, cols 4(i&3)..+3) and lane i receives column i of the block.
    __device__ static __forceinline__ frag lds_tmajor(const __bf16* base, int pitch, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const __bf16* p = base + (4 * g + (i >> 2)) * pitch + 4 * (i & 3);
        typedef __attribute__((address_space(3))) short4_t lds_s4;
        short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
        short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 16 * pitch));
        union { short s[8]; frag f; } u;
        u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
        u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
        return u.f;
    }
    __device__ static __forceinline__ frag pack(bf16x4_t lo, bf16x4_t hi) {
        frag f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
    __device__ static __forceinline__ frag pack16(const unsigned char*) { return frag{}; }      // f32 form only
    __device__ static __forceinline__ frag gather4(const unsigned char*, const unsigned char*, const unsigned char*,
                                                   const unsigned char*) { return frag{}; }      // f32 form only
    // two transpose reads at explicit LDS byte addresses (swizzled layouts)
    __device__ static __forceinline__ frag tr_pair(const unsigned char* p0, const unsigned char* p1) {
        typedef __attribute__((address_space(3))) short4_t lds_s4;
        short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p0));
        short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p1));
        union { short s[8]; frag f; } u;
        u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
        u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
        return u.f;
    }
    // Two stacked accumulator tiles (contraction rows 0-15 in c0, 16-31 in c1) -> operand fragment.
    __device__ static __forceinline__ frag from_acc(f32x4_t c0, f32x4_t c1) {
        frag f;
        f[0] = (__bf16)c0[0]; f[1] = (__bf16)c0[1]; f[2] = (__bf16)c0[2]; f[3] = (__bf16)c0[3];
        f[4] = (__bf16)c1[0]; f[5] = (__bf16)c1[1]; f[6] = (__bf16)c1[2]; f[7] = (__bf16)c1[3];
        return f;
    }
};

template <> struct Mma<float> {
    static constexpr int CH = 16;
    static constexpr int EPC = 4;
    typedef f32x4_t frag;
    static constexpr int kpad = 4, tpad = 4;
    __device__ static __forceinline__ f32x4_t mma(frag a, frag b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
    __device__ static __forceinline__ frag lds_kmajor(const float* p, int g) {
        return *reinterpret_cast<const f32x4_t*>(p + 4 * g);
    }
    __device__ static __forceinline__ frag gmem_kmajor(const float* p, int g) {
        return *reinterpret_cast<const f32x4_t*>(p + 4 * g);
    }
    __device__ static __forceinline__ frag lds_tmajor(const float* base, int pitch, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const float* p = base + (4 * g) * pitch + i;
        frag f;
        f[0] = p[0]; f[1] = p[pitch]; f[2] = p[2 * pitch]; f[3] = p[3 * pitch];
        return f;
    }
    __device__ static __forceinline__ frag pack(bf16x4_t, bf16x4_t) { return frag{0.f, 0.f, 0.f, 0.f}; }   // bf16 form only
    __device__ static __forceinline__ frag tr_pair(const unsigned char*, const unsigned char*) { return frag{0.f, 0.f, 0.f, 0.f}; }
    __device__ static __forceinline__ frag pack16(const unsigned char* p) { return *reinterpret_cast<const f32x4_t*>(p); }
    __device__ static __forceinline__ frag gather4(const unsigned char* p0, const unsigned char* p1, const unsigned char* p2,
                                                   const unsigned char* p3) {
        frag f;
        f[0] = *reinterpret_cast<const float*>(p0); f[1] = *reinterpret_cast<const float*>(p1);
        f[2] = *reinterpret_cast<const float*>(p2); f[3] = *reinterpret_cast<const float*>(p3);
        return f;
    }
    // one accumulator tile covers the whole 16-wide chunk
    __device__ static __forceinline__ frag from_acc(f32x4_t c0, f32x4_t /*unused*/) { return c0; }
};

// dtype codes of the C ABI
enum { UNIVL_F32 = 0, UNIVL_BF16 = 1 };
