// Error string, version, device info and the hardware-layout probes that pin the assumptions of common.h
// (MFMA operand / accumulator lane maps and the ds_read_b64_tr_b16 transpose read) on the real gfx950.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <dlfcn.h>
#include <stdlib.h>
#include "common.h"
#include "univl_hip.h"

static thread_local char g_err[512] = "";

void univl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* univl_last_error(void) { return g_err; }
extern "C" int univl_version(void) { return 100; }

// sizeof() of the ABI structs so that the ctypes mirrors on the Python side can be verified at load time
extern "C" int univl_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(UnivlGemm);
        case 1: return (int)sizeof(UnivlLayerNorm);
        case 2: return (int)sizeof(UnivlAttention);
        case 3: return (int)sizeof(UnivlEmbedText);
        case 4: return (int)sizeof(UnivlPool);
        case 5: return (int)sizeof(UnivlSeg);
        case 6: return (int)sizeof(UnivlAdam);
        case 7: return (int)sizeof(UnivlVocabCE);
        default: return -1;
    }
}

extern "C" int univl_device_info(int* cu_count, char* name, int name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { univl_set_error("hipGetDevice: %s", hipGetErrorString(e)); return (int)e; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { univl_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return (int)e; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (name && name_len > 0) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return 0;
}

// ---- deterministic mode (common.h): flag, per-device scratch ring and counter pool
#include <mutex>
namespace {
struct DetArena { unsigned char* base = nullptr; size_t size = 0, head = 0; int* counters = nullptr; int next_counter = 0; };
constexpr int DET_COUNTERS = 1 << 16;
DetArena g_det[UNIVL_MAX_DEVICES];
std::mutex g_det_mutex;
int g_deterministic = 0;

DetArena* det_arena() {          // the current device's arena, created on first use (NOT inside a stream capture: hipMalloc)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UNIVL_MAX_DEVICES) return nullptr;
    DetArena& a = g_det[dev];
    if (a.base == nullptr) {
        const char* e = getenv("UNIVL_DET_ARENA_MB");
        const size_t mb = e ? (size_t)atol(e) : 1024;
        void* p = nullptr;
        void* c = nullptr;
        if (hipMalloc(&p, mb << 20) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMalloc(&c, DET_COUNTERS * sizeof(int)) != hipSuccess || hipMemset(c, 0, DET_COUNTERS * sizeof(int)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(p); if (c) (void)hipFree(c); return nullptr;
        }
        a.base = static_cast<unsigned char*>(p); a.size = mb << 20; a.head = 0;
        a.counters = static_cast<int*>(c); a.next_counter = 0;
    }
    return &a;
}
}  // namespace

bool univl_deterministic() { return g_deterministic != 0; }

void* univl_det_alloc(size_t bytes) {
    std::lock_guard<std::mutex> lock(g_det_mutex);
    DetArena* a = det_arena();
    bytes = (bytes + 255) & ~(size_t)255;
    if (a == nullptr || bytes > a->size / 4) {
        univl_set_error("deterministic mode: no scratch ring on this device (call univl_set_deterministic(1) on it outside a stream "
                        "capture; a single request must fit a quarter of UNIVL_DET_ARENA_MB), %zu bytes asked", bytes);
        return nullptr;
    }
    if (a->head + bytes > a->size) a->head = 0;      // ring: a slice is reused only after the whole ring went by
    void* p = a->base + a->head;
    a->head += bytes;
    return p;
}

int* univl_det_counter() {
    std::lock_guard<std::mutex> lock(g_det_mutex);
    DetArena* a = det_arena();
    if (a == nullptr) { univl_set_error("deterministic mode: no counter pool on this device"); return nullptr; }
    int* c = a->counters + a->next_counter;
    a->next_counter = (a->next_counter + 1) % DET_COUNTERS;
    return c;
}

extern "C" int univl_set_deterministic(int on) {
    std::lock_guard<std::mutex> lock(g_det_mutex);
    g_deterministic = on ? 1 : 0;
    if (on && det_arena() == nullptr) {
        univl_set_error("univl_set_deterministic: cannot allocate the scratch ring (UNIVL_DET_ARENA_MB) on the current device");
        return UNIVL_EINVAL;
    }
    return UNIVL_OK;
}

extern "C" int univl_get_deterministic(void) { return g_deterministic; }

// ---- several buffers cleared by ONE launch (a backward starts by zeroing ~8 accumulation buffers: one graph node, not 8)
namespace {
struct ZeroList { unsigned char* ptr[UNIVL_ZERO_MAX]; long bytes[UNIVL_ZERO_MAX]; };
__global__ __launch_bounds__(256) void zero_many_kernel(ZeroList z) {
    unsigned char* p = z.ptr[blockIdx.y];
    const long n16 = z.bytes[blockIdx.y] >> 4;
    u32x4_t* q = reinterpret_cast<u32x4_t*>(p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) q[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0) for (long i = (n16 << 4) + threadIdx.x; i < z.bytes[blockIdx.y]; i += 256) p[i] = 0;
}
}  // namespace

extern "C" int univl_zero_many(void* const* ptrs, const int64_t* bytes, int32_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(ptrs && bytes && n >= 1 && n <= UNIVL_ZERO_MAX, UNIVL_EINVAL, "univl_zero_many: n=%d (1..%d)", n, UNIVL_ZERO_MAX);
    ZeroList z;
    long most = 0;
    for (int i = 0; i < n; ++i) {
        UNIVL_CHECK_ARG(ptrs[i] && bytes[i] > 0 && aligned16(ptrs[i]), UNIVL_EALIGN, "univl_zero_many: buffer %d must be non-empty and 16-byte aligned", i);
        z.ptr[i] = static_cast<unsigned char*>(ptrs[i]);
        z.bytes[i] = bytes[i];
        most = bytes[i] > most ? bytes[i] : most;
    }
    long gx = (most / 16 + 2047) / 2048;                       // ~8 x 16 B per thread for the largest buffer
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(zero_many_kernel, dim3((unsigned)gx, n), dim3(256), 0, stream, z);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

// ---- several small device-to-device copies as ONE kernel (a training step stages five input tensors and hands the loss to the
// caller: as hipMemcpyAsync calls these are memcpy NODES of the captured step, which the runtime executes as blit kernels at a
// ~10 us cadence each -- measured in the round-2 kernel trace -- against ~5 us for one ordinary kernel node doing all of them)
namespace {
struct CopyList { const unsigned char* src[UNIVL_ZERO_MAX]; unsigned char* dst[UNIVL_ZERO_MAX]; long bytes[UNIVL_ZERO_MAX]; };
__global__ __launch_bounds__(256) void copy_many_kernel(CopyList c) {
    const unsigned char* s = c.src[blockIdx.y];
    unsigned char* d = c.dst[blockIdx.y];
    const long nb = c.bytes[blockIdx.y];
    const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;          // block-uniform
    const long n16 = vec ? (nb >> 4) : 0;
    const u32x4_t* sv = reinterpret_cast<const u32x4_t*>(s);
    u32x4_t* dv = reinterpret_cast<u32x4_t*>(d);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dv[i] = sv[i];
    if (blockIdx.x == 0) for (long i = (n16 << 4) + threadIdx.x; i < nb; i += 256) d[i] = s[i];
}
}  // namespace

extern "C" int univl_copy_many(const void* const* srcs, void* const* dsts, const int64_t* bytes, int32_t n, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(srcs && dsts && bytes && n >= 1 && n <= UNIVL_ZERO_MAX, UNIVL_EINVAL, "univl_copy_many: n=%d (1..%d)", n, UNIVL_ZERO_MAX);
    CopyList c;
    long most = 0;
    for (int i = 0; i < n; ++i) {
        UNIVL_CHECK_ARG(srcs[i] && dsts[i] && bytes[i] > 0, UNIVL_EINVAL, "univl_copy_many: buffer %d is null or empty", i);
        c.src[i] = static_cast<const unsigned char*>(srcs[i]);
        c.dst[i] = static_cast<unsigned char*>(dsts[i]);
        c.bytes[i] = bytes[i];
        most = bytes[i] > most ? bytes[i] : most;
    }
    for (int i = n; i < UNIVL_ZERO_MAX; ++i) { c.src[i] = c.src[0]; c.dst[i] = c.dst[0]; c.bytes[i] = 0; }
    long gx = (most / 16 + 1023) / 1024;                       // ~4 x 16 B per thread for the largest buffer
    gx = gx < 1 ? 1 : (gx > 2048 ? 2048 : gx);
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)gx, n), dim3(256), 0, stream, c);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}

extern "C" int univl_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { univl_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return (int)e; }
    UNIVL_CHECK_ARG(device >= 0 && device < n, UNIVL_EINVAL, "univl_init: device %d of %d", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { univl_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return (int)e; }
    UNIVL_CHECK_ARG(strncmp(prop.gcnArchName, "gfx950", 6) == 0, UNIVL_EUNSUPPORTED,
                    "univl_init: device %d is %s; libunivl_hip holds gfx950 (MI355X) code only", device, prop.gcnArchName);
    return UNIVL_OK;
}

// ---- RCCL, resolved lazily so that the library itself has no link-time dependency on it
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static void* g_rccl_handle = nullptr;
static nccl_allreduce_fn g_allreduce = nullptr;

extern "C" int univl_destroy(void) {
    {
        std::lock_guard<std::mutex> lock(g_det_mutex);
        for (DetArena& a : g_det) {
            if (a.base) (void)hipFree(a.base);
            if (a.counters) (void)hipFree(a.counters);
            a = DetArena();
        }
    }
    g_allreduce = nullptr;
    if (g_rccl_handle) { dlclose(g_rccl_handle); g_rccl_handle = nullptr; }
    return UNIVL_OK;
}

extern "C" int univl_allreduce_bucket(void* buf, size_t n, int dtype, int average, void* comm, hipStream_t side) {
    UNIVL_ON_STREAM_DEVICE(side);
    UNIVL_CHECK_ARG(buf && comm && n > 0 && (dtype == UNIVL_F32 || dtype == UNIVL_BF16), UNIVL_EINVAL,
                    "univl_allreduce_bucket: bad argument");
    if (!g_allreduce) {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");               // the copy the host already loaded (torch's)
        if (!sym) {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                g_rccl_handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (g_rccl_handle) break;
            }
            if (g_rccl_handle) sym = dlsym(g_rccl_handle, "ncclAllReduce");
        }
        UNIVL_CHECK_ARG(sym != nullptr, UNIVL_EUNSUPPORTED, "univl_allreduce_bucket: RCCL (ncclAllReduce) not found");
        g_allreduce = reinterpret_cast<nccl_allreduce_fn>(sym);
    }
    // rccl.h: ncclFloat32 = 7, ncclBfloat16 = 9; ncclSum = 0, ncclAvg = 4
    const int rc = g_allreduce(buf, buf, n, dtype == UNIVL_BF16 ? 9 : 7, average ? 4 : 0, comm, side);
    if (rc != 0) { univl_set_error("univl_allreduce_bucket: ncclAllReduce returned %d", rc); return 1000 + rc; }
    return UNIVL_OK;
}

namespace {

// deterministic small-integer test matrices (exact in bf16 and fp32)
__device__ __forceinline__ float tv(int a, int b, int salt) {
    int h = (a * 7 + b * 13 + salt * 5 + a * b) % 5;
    return (float)(h - 2);
}

// out layout (floats):
//   [0,256)      bf16  C = X.Y^T, X,Y K-major 16x32          (C[r][c] at r*16+c)
//   [256,512)    bf16  same product with X staged T-major ([32][16]) and read with the transpose read
//   [512,768)    f32   C = X.Y^T, 16x16 contraction, K-major
//   [768,1024)   f32   same with X T-major
//   [1024,1280)  bf16  chained: D = Z . S,  S[32x16] = [X1;X2].Y^T fed through Mma::from_acc
//   [1280,1536)  f32   chained: D = Z . S,  S[16x16] = X.Y^T
//   [1536,1792)  raw ds_read_b64_tr_b16 dump: lane l, element e at 1536 + 4*l + e, image value = 64*row + col
template <typename T>
__device__ void probe_mma(float* out_k, float* out_t, float* out_chain, T* lds) {
    constexpr int CH = Mma<T>::CH;
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    constexpr int PK = CH + Mma<T>::kpad, PT = 16 + Mma<T>::tpad;
    T* xk = lds;                 // [16][PK]   X K-major
    T* yk = xk + 16 * PK;        // [16][PK]   Y K-major
    T* xt = yk + 16 * PK;        // [CH][PT]   X T-major
    T* zk = xt + CH * PT;        // [16][PK]   Z K-major (chain)
    for (int e = lane; e < 16 * CH; e += 64) {
        const int r = e / CH, k = e % CH;
        xk[r * PK + k] = from_f32<T>(tv(r, k, 1));
        yk[r * PK + k] = from_f32<T>(tv(r, k, 2));
        xt[k * PT + r] = from_f32<T>(tv(r, k, 1));
        zk[r * PK + k] = from_f32<T>(tv(r, k, 3));
    }
    __syncthreads();
    typename Mma<T>::frag fy = Mma<T>::lds_kmajor(yk + i * PK, g);
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    c = Mma<T>::mma(Mma<T>::lds_kmajor(xk + i * PK, g), fy, c);
    for (int r = 0; r < 4; ++r) out_k[(4 * g + r) * 16 + i] = c[r];
    f32x4_t ct = {0.f, 0.f, 0.f, 0.f};
    ct = Mma<T>::mma(Mma<T>::lds_tmajor(xt, PT, lane), fy, ct);
    for (int r = 0; r < 4; ++r) out_t[(4 * g + r) * 16 + i] = ct[r];
    // chain: S tiles = X{1,2}.Y^T with rows = contraction index of the next product
    f32x4_t s0 = c, s1 = {0.f, 0.f, 0.f, 0.f};
    if (CH == 32) {
        // second stacked tile: X2[r][k] = tv(r+16, k, 1)
        __syncthreads();
        for (int e = lane; e < 16 * CH; e += 64) { const int r = e / CH, k = e % CH; xk[r * PK + k] = from_f32<T>(tv(r + 16, k, 1)); }
        __syncthreads();
        s1 = Mma<T>::mma(Mma<T>::lds_kmajor(xk + i * PK, g), fy, s1);
    }
    typename Mma<T>::frag fs = Mma<T>::from_acc(s0, s1);     // operand B: [contraction CH][16 cols]
    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
    d = Mma<T>::mma(Mma<T>::lds_kmajor(zk + i * PK, g), fs, d);
    for (int r = 0; r < 4; ++r) out_chain[(4 * g + r) * 16 + i] = d[r];
}

__global__ void probe_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    probe_mma<__bf16>(out, out + 256, out + 1024, reinterpret_cast<__bf16*>(lds));
    __syncthreads();
    probe_mma<float>(out + 512, out + 768, out + 1280, reinterpret_cast<float*>(lds));
    __syncthreads();
    short* img = reinterpret_cast<short*>(lds);   // 16 rows x 64 cols, value = 64*row + col
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 16 * 64; e += 64) img[e] = (short)e;
    __syncthreads();
    // each lane supplies the 8-byte piece (row = lane>>2 within 16 rows, cols 4*(lane&3)..) -> dump what it gets
    typedef __attribute__((address_space(3))) short4_t lds_s4;
    const int row = (lane >> 2) & 15, cc = 4 * (lane & 3);
    short4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + row * 64 + cc));
    for (int e = 0; e < 4; ++e) out[1536 + 4 * lane + e] = (float)r[e];
}

}  // namespace

extern "C" int univl_probe_layouts(float* out, int32_t n_out, hipStream_t stream) {
    UNIVL_ON_STREAM_DEVICE(stream);
    UNIVL_CHECK_ARG(out && n_out >= 1792, UNIVL_EINVAL, "univl_probe_layouts: need 1792 floats");
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, stream, out);
    UNIVL_LAUNCH_CHECK();
    return UNIVL_OK;
}
