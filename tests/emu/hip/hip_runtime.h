// TEST TOOLING ONLY -- a host-side stand-in for <hip/hip_runtime.h>.
//
// It lets the *unmodified* HIP sources under flowmol_amd/csrc/ be compiled with the host clang++
// (tests/emu/build_emu.sh puts this directory first on the include path) into
// tests/emu/libflowmol_emu.so, where every workgroup is executed on the CPU: lanes are
// coroutines, __syncthreads / MFMA / shuffles are rendezvous points, LDS is a heap buffer.
// Purpose: debug index math, LDS layouts, MFMA fragment maps and weight packing in the build
// container (which has no GPU) before spending GPU minutes.  The product never loads this
// library: flowmol_amd/_lib.py only opens libflowmol_hip.so and fails loudly if it is missing.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define FM_HOST_EMULATION 1

// ---------------------------------------------------------------- qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------- vector types
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

// ---------------------------------------------------------------- runtime (emu_rt.cpp)
namespace emu {
struct Lane;
struct BlockCtx {
    dim3 blockIdx, blockDim, gridDim;
    unsigned char* lds;
    size_t lds_bytes;
};
struct LaneView { dim3 tid; };
Lane* cur();
BlockCtx& blk();
const dim3& tid();
void syncthreads();
float shfl_f(float v, int src_lane, int width);
int shfl_i(int v, int src_lane, int width);
void mfma16(float a, float b, float* c4);          // 16x16x4 f32
void mfma4(float a, float b, float* c4);           // 4x4x1, 16 blocks
void mfma32(float a, float b, float* c16);         // 32x32x2 f32
void mfma16_bf16(const unsigned short* a8, const unsigned short* b8, float* c4);   // 16x16x32 bf16
void mfma16_f16(const unsigned short* a8, const unsigned short* b8, float* c4);    // 16x16x32 f16 (subnormals kept)
unsigned long long ballot(int pred);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
int lane_id();
}  // namespace emu

#define threadIdx (emu::tid())
#define blockIdx (emu::blk().blockIdx)
#define blockDim (emu::blk().blockDim)
#define gridDim (emu::blk().gridDim)
#define warpSize 64

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::blk().lds);

static inline void __syncthreads() { emu::syncthreads(); }

static inline float __shfl(float v, int src, int width = 64) { return emu::shfl_f(v, src, width); }
static inline int __shfl(int v, int src, int width = 64) { return emu::shfl_i(v, src, width); }
static inline float __shfl_xor(float v, int mask, int width = 64) { return emu::shfl_f(v, emu::lane_id() ^ mask, width); }
static inline int __shfl_xor(int v, int mask, int width = 64) { return emu::shfl_i(v, emu::lane_id() ^ mask, width); }
static inline float __shfl_down(float v, unsigned d, int width = 64) {
    int l = emu::lane_id();
    int s = ((l % width) + (int)d < width) ? l + (int)d : l;
    return emu::shfl_f(v, s, 64);
}
static inline int __shfl_down(int v, unsigned d, int width = 64) {
    int l = emu::lane_id();
    int s = ((l % width) + (int)d < width) ? l + (int)d : l;
    return emu::shfl_i(v, s, 64);
}
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred); }

typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
static inline emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c, int, int, int) {
    float t[4] = {c[0], c[1], c[2], c[3]};
    emu::mfma16(a, b, t);
    return emu_f32x4{t[0], t[1], t[2], t[3]};
}
static inline emu_f32x16 emu_mfma_32x32x2(float a, float b, emu_f32x16 c, int, int, int) {
    float t[16];
    for (int i = 0; i < 16; ++i) t[i] = c[i];
    emu::mfma32(a, b, t);
    emu_f32x16 r;
    for (int i = 0; i < 16; ++i) r[i] = t[i];
    return r;
}
static inline emu_f32x4 emu_mfma_4x4x1(float a, float b, emu_f32x4 c, int, int, int) {
    float t[4] = {c[0], c[1], c[2], c[3]};
    emu::mfma4(a, b, t);
    return emu_f32x4{t[0], t[1], t[2], t[3]};
}
#define __builtin_amdgcn_mfma_f32_4x4x1f32 emu_mfma_4x4x1
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short emu_u16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    const emu_u16x8 ai = __builtin_bit_cast(emu_u16x8, a), bi = __builtin_bit_cast(emu_u16x8, b);
    unsigned short ha[8], hb[8];
    for (int q = 0; q < 8; ++q) { ha[q] = ai[q]; hb[q] = bi[q]; }
    float t[4] = {c[0], c[1], c[2], c[3]};
    emu::mfma16_bf16(ha, hb, t);
    return emu_f32x4{t[0], t[1], t[2], t[3]};
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_16x16x32_bf16
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x4 emu_mfma_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c, int, int, int) {
    const emu_u16x8 ai = __builtin_bit_cast(emu_u16x8, a), bi = __builtin_bit_cast(emu_u16x8, b);
    unsigned short ha[8], hb[8];
    for (int q = 0; q < 8; ++q) { ha[q] = ai[q]; hb[q] = bi[q]; }
    float t[4] = {c[0], c[1], c[2], c[3]};
    emu::mfma16_f16(ha, hb, t);
    return emu_f32x4{t[0], t[1], t[2], t[3]};
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 emu_mfma_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_16x16x4
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2

// ---------------------------------------------------------------- device math
static inline float __fdividef(float a, float b) { return a / b; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
// buffer-descriptor accesses: base pointer + byte count; raw-buffer range checking like the hardware's
// (an access whose voffset+imm range leaves [0, num_records) reads 0 / is dropped; soffset is not range-checked)
struct emu_rsrc { char* base; unsigned num_records; };
static inline emu_rsrc emu_make_rsrc(void* p, short, int n, int) { return emu_rsrc{static_cast<char*>(p), (unsigned)n}; }
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
static inline bool emu_buf_ok(const emu_rsrc& r, int voff, unsigned bytes) { return (unsigned long long)(unsigned)voff + bytes <= r.num_records; }
static inline unsigned emu_raw_buffer_load_b32(emu_rsrc r, int voff, int soff, int) {
    unsigned v = 0;
    if (emu_buf_ok(r, voff, 4)) memcpy(&v, r.base + (unsigned)voff + (unsigned)soff, 4);
    return v;
}
static inline emu_u32x2 emu_raw_buffer_load_b64(emu_rsrc r, int voff, int soff, int) {
    emu_u32x2 v = {0u, 0u};
    if (emu_buf_ok(r, voff, 8)) memcpy(&v, r.base + (unsigned)voff + (unsigned)soff, 8);
    return v;
}
static inline emu_u32x4 emu_raw_buffer_load_b128(emu_rsrc r, int voff, int soff, int) {
    emu_u32x4 v = {0u, 0u, 0u, 0u};
    if (emu_buf_ok(r, voff, 16)) memcpy(&v, r.base + (unsigned)voff + (unsigned)soff, 16);
    return v;
}
static inline void emu_raw_buffer_store_b32(unsigned d, emu_rsrc r, int voff, int soff, int) {
    if (emu_buf_ok(r, voff, 4)) memcpy(r.base + (unsigned)voff + (unsigned)soff, &d, 4);
}
static inline void emu_raw_buffer_store_b64(emu_u32x2 d, emu_rsrc r, int voff, int soff, int) {
    if (emu_buf_ok(r, voff, 8)) memcpy(r.base + (unsigned)voff + (unsigned)soff, &d, 8);
}
static inline void emu_raw_buffer_store_b128(emu_u32x4 d, emu_rsrc r, int voff, int soff, int) {
    if (emu_buf_ok(r, voff, 16)) memcpy(r.base + (unsigned)voff + (unsigned)soff, &d, 16);
}
#define __builtin_amdgcn_raw_buffer_store_b128 emu_raw_buffer_store_b128
#define __builtin_amdgcn_raw_buffer_store_b64 emu_raw_buffer_store_b64
#define __builtin_amdgcn_make_buffer_rsrc emu_make_rsrc
#define __builtin_amdgcn_raw_buffer_load_b32 emu_raw_buffer_load_b32
#define __builtin_amdgcn_raw_buffer_load_b64 emu_raw_buffer_load_b64
#define __builtin_amdgcn_raw_buffer_load_b128 emu_raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_store_b32 emu_raw_buffer_store_b32
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_readlane(v, l) __shfl((int)(v), (int)(l))   /* must be called by the whole wave */
#define __builtin_amdgcn_readfirstlane(x) (x)   /* value is wave-uniform by contract */
// DPP / swizzle lane exchanges (must be called by the whole wave): the source lane each control selects
static inline int emu_update_dpp(int /*old*/, int v, int ctrl, int, int, bool) {
    const int l = emu::lane_id();
    int src;
    if (ctrl < 0x100) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);           // quad_perm
    else if (ctrl == 0x140) src = (l & ~15) | (15 - (l & 15));                   // row_mirror
    else if (ctrl == 0x141) src = (l & ~7) | (7 - (l & 7));                      // row_half_mirror
    else abort();
    return emu::shfl_i(v, src, 64);
}
static inline int emu_ds_swizzle(int v, int pattern) {                           // bit-mask mode within groups of 32 lanes
    if (pattern & 0x8000) abort();
    const int l = emu::lane_id(), l5 = l & 31;
    const int src5 = ((l5 & (pattern & 31)) | ((pattern >> 5) & 31)) ^ ((pattern >> 10) & 31);
    return emu::shfl_i(v, (l & ~31) | src5, 64);
}
#define __builtin_amdgcn_update_dpp emu_update_dpp
#define __builtin_amdgcn_ds_swizzle emu_ds_swizzle
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
// correctly-rounded, never-contracted f32 ops (the emulation is built with -ffp-contract=off)
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }

// ---------------------------------------------------------------- atomics (blocks may run on several OS threads)
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }   /* lanes run one at a time on the host */
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), neu;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&neu, &f, 4);
    } while (!__atomic_compare_exchange_n(ip, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}

// ---------------------------------------------------------------- host API subset
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
typedef void* hipStream_t;
struct emu_event { double t; };
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

static inline hipError_t hipMalloc(void** p, size_t n) { *p = n ? aligned_alloc(256, (n + 255) / 256 * 256) : nullptr; return (*p || !n) ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated hip error"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "host-emulation");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 8;
    return hipSuccess;
}
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
double emu_now_ms();
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0}; return hipSuccess; }
#define hipEventDisableTiming 2u
#define hipHostMallocDefault 0u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event{0}; return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = emu_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })
