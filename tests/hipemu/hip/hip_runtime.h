// hip/hip_runtime.h of the TEST-ONLY kernel emulator (tests/hipemu/README.md).
//
// This header stands in for ROCm's <hip/hip_runtime.h> when the sources under openhevc_amd/csrc/ are compiled for the
// host CPU (x86-64 clang, `-x c++`) into tests/hipemu/libohevc_hip_emu.so.  The emulator runs the UNCHANGED kernel
// source: every workgroup lane is a fiber, __syncthreads / wave-level exchanges are scheduling points (hipemu.cpp).
// It exists so that `pytest -m "not gpu"` can check the device code's arithmetic and index algebra without a GPU.
// It is NOT a product path: nothing under openhevc_amd/ loads it, and libohevc_hip.so never falls back to it.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <functional>

// ---------------------------------------------------------------- language keywords
#define __global__
#define __device__
#define __host__
#define __constant__ static const
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
struct Lane {                       // what the running fiber sees
    dim3 tid, bid, bdim, gdim;
    int  flat;                      // flat thread index inside the workgroup
};
extern thread_local Lane g_lane;
void     launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void     sync_block();
void     sync_wave();
int      block_or(int pred);                                  // __syncthreads_or
uint64_t wave_exchange(uint64_t mine, int src_lane);          // value `src_lane` (0..63 of my wave) published
uint64_t wave_ballot(bool pred);
uint64_t wave_first(uint64_t mine);                           // value of the lowest live lane
void     wave_gather(const void *mine, size_t bytes, void *all64);   // all64[lane*bytes..]: what every lane published (dead lanes: zeros)
[[noreturn]] void unsupported(const char *what);
void     spin();                                              // a poll that found nothing; aborts after 2^22 per launch
}  // namespace hipemu

#define threadIdx (::hipemu::g_lane.tid)
#define blockIdx  (::hipemu::g_lane.bid)
#define blockDim  (::hipemu::g_lane.bdim)
#define gridDim   (::hipemu::g_lane.gdim)
static constexpr int warpSize = 64;

// ---------------------------------------------------------------- runtime API (the subset the library uses)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef struct hipemuStream *hipStream_t;
typedef struct hipemuEvent  *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int  multiProcessorCount;
    size_t totalGlobalMem;
    int  warpSize;
};

extern "C" {
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize(void);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s);
hipError_t hipMemset(void *dst, int v, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t cuMaskSize, const uint32_t *cuMask);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void **>(p), n, f); }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e) { return hipStreamWaitEvent(s, e, 0); }

// Launches run synchronously on the calling host thread (streams are ordering-only objects, so this is one legal schedule).
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hipemu::launch(dim3(grid), dim3(block), (shmem), [=]() { kernel(__VA_ARGS__); })

// ---------------------------------------------------------------- device functions
static inline void __syncthreads() { ::hipemu::sync_block(); }
static inline int __syncthreads_or(int pred) { return ::hipemu::block_or(pred); }
#define __builtin_amdgcn_wave_barrier() ::hipemu::sync_wave()
#define __builtin_amdgcn_s_barrier() ::hipemu::sync_block()
#define __builtin_amdgcn_fence(order, scope, ...) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ::hipemu::spin()     // workgroups run one after another: a wait on a LATER workgroup never ends
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(ptr, order, scope) __atomic_load_n(ptr, order)
#define __hip_atomic_store(ptr, v, order, scope) __atomic_store_n(ptr, v, order)
#define __hip_atomic_fetch_add(ptr, v, order, scope) __atomic_fetch_add(ptr, v, order)
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() {}

template <typename T> static inline T __shfl(T v, int src, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shfl of wide types");
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    int lane = ::hipemu::g_lane.flat & 63;
    int s = (lane & ~(width - 1)) | (src & (width - 1));
    raw = ::hipemu::wave_exchange(raw, s);
    T r;
    memcpy(&r, &raw, sizeof(T));
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (::hipemu::g_lane.flat & 63) ^ mask, 64); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{
    int lane = ::hipemu::g_lane.flat & 63;
    int s = ((lane & (width - 1)) + (int)d < width) ? lane + (int)d : lane;
    return __shfl(v, s, 64);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{
    int lane = ::hipemu::g_lane.flat & 63;
    int s = ((lane & (width - 1)) >= (int)d) ? lane - (int)d : lane;
    return __shfl(v, s, 64);
}
static inline unsigned long long __ballot(int pred) { return ::hipemu::wave_ballot(pred != 0); }
static inline int __any(int pred) { return ::hipemu::wave_ballot(pred != 0) != 0; }
static inline int __all(int pred) { return ::hipemu::wave_ballot(pred == 0) == 0; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

template <typename T> static inline T hipemu_readfirstlane(T v)
{
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    raw = ::hipemu::wave_first(raw);
    T r;
    memcpy(&r, &raw, sizeof(T));
    return r;
}
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
#define __builtin_amdgcn_readlane(v, l) __shfl((v), (l), 64)
#define __builtin_amdgcn_ds_bpermute(addr, v) __shfl((int)(v), (int)(((unsigned)(addr)) >> 2) & 63, 64)

// v_perm_b32: bytes of {a (7..4), b (3..0)} picked by the four selector bytes (0x0c = 0x00, 0x0d.. = 0xff for >=0x0d per ISA: 12 -> 0, >=13 -> 0xff)
static inline unsigned hipemu_perm(unsigned a, unsigned b, unsigned sel)
{
    uint64_t src = (uint64_t(a) << 32) | b;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        unsigned s = (sel >> (8 * i)) & 0xff, byte;
        if (s <= 7) byte = (src >> (8 * s)) & 0xff;
        else if (s <= 11) byte = ((src >> (16 * (s - 8) + 15)) & 1) ? 0xff : 0x00;   // sign of the 16-bit halves
        else if (s == 12) byte = 0x00;
        else byte = 0xff;
        r |= byte << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_perm(a, b, sel) hipemu_perm((a), (b), (sel))
static inline unsigned hipemu_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((uint64_t(hi) << 32) | lo) >> (sh & 31)); }
#define __builtin_amdgcn_alignbit(hi, lo, sh) hipemu_alignbit((hi), (lo), (sh))
static inline unsigned hipemu_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((uint64_t(hi) << 32) | lo) >> (8 * (sh & 3))); }
#define __builtin_amdgcn_alignbyte(hi, lo, sh) hipemu_alignbyte((hi), (lo), (sh))

typedef short hipemu_s16x2 __attribute__((ext_vector_type(2)));
static inline int hipemu_sdot2(hipemu_s16x2 a, hipemu_s16x2 b, int c, bool clamp)
{
    (void)clamp;
    return (int)((unsigned)c + (unsigned)((int)a.x * (int)b.x) + (unsigned)((int)a.y * (int)b.y));
}
#define __builtin_amdgcn_sdot2(a, b, c, clamp) hipemu_sdot2((a), (b), (c), (clamp))
static inline hipemu_s16x2 hipemu_cvt_pk_i16(int lo, int hi)
{
    auto sat = [](int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); };
    hipemu_s16x2 r = { sat(lo), sat(hi) };
    return r;
}
#define __builtin_amdgcn_cvt_pk_i16(lo, hi) hipemu_cvt_pk_i16((lo), (hi))

// 24-bit multiplies (v_mul_i32_i24 / v_mul_u32_u24): the operands' low 24 bits, sign- / zero-extended; low 32 bits of the product
static inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }

// matrix-core and transposing-LDS-read instructions: wave collectives, emulated in hipemu_mfma.hpp
#include "hipemu_mfma.hpp"

// ---------------------------------------------------------------- atomics / small math
static inline unsigned long long clock64() { return 0; }      // (no device clock on the host: the phase counters of diagnosis builds read 0)
template <typename T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicMax(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <typename T> static inline T atomicMin(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <typename T> static inline T atomicCAS(T *p, T expected, T desired)
{
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}
template <typename T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
