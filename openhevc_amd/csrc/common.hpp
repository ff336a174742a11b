// common.hpp -- shared host/device helpers for libohevc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "ohevc_hip.h"

// a constant table defined in a header (one copy per translation unit); the host emulation of tests/hipemu spells __constant__ itself
#ifdef OHEVC_HIPEMU
#define OHEVC_CONST_TABLE static const
#else
#define OHEVC_CONST_TABLE static __constant__
#endif

namespace ohevc {

// ---- error plumbing: every HIP failure is recorded (thread-local) and surfaces as OHEVC_ERR_HIP
void set_error(const char *fmt, ...);

#define OHEVC_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::ohevc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),   \
                               __FILE__, __LINE__);                                      \
            return OHEVC_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

// The bit depths the reference instantiates (hevcdsp.c:1044-1062: 8, 9, 10, 12, 14), plus 11, which the same formulas cover.
#define OHEVC_BIT_DEPTH_OK(bd) (((bd) >= 8 && (bd) <= 12) || (bd) == 14)
#define OHEVC_REQUIRE(cond, msg)                                                         \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ::ohevc::set_error("bad argument: %s (%s)", msg, #cond);                     \
            return OHEVC_ERR_ARG;                                                        \
        }                                                                                \
    } while (0)

// Kernel-side view of the up-to-3 planes of one picture (passed by value as a kernel argument).
struct PlaneSet {
    unsigned char *data[3];
    int            stride[3];   // bytes
    int            width[3];
    int            height[3];
};

// align: required alignment (bytes) of plane base and stride; the TU kernels use 16-byte row accesses (16),
// the per-sample kernels only need natural pixel alignment (pixel size).
inline int make_plane_set(const ohevc_plane planes[3], PlaneSet &ps, int align = 16)
{
    for (int i = 0; i < 3; i++) {
        ps.data[i]   = static_cast<unsigned char *>(planes[i].data);
        ps.stride[i] = planes[i].stride;
        ps.width[i]  = planes[i].width;
        ps.height[i] = planes[i].height;
        if (planes[i].data) {
            OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(planes[i].data) & (align - 1)) == 0, "plane data is not sufficiently aligned");
            OHEVC_REQUIRE((planes[i].stride & (align - 1)) == 0 && planes[i].stride > 0, "plane stride must be a positive multiple of the required alignment");
        }
    }
    return OHEVC_OK;
}

// lane-varying plane select without taking the address of the by-value kernel argument (keeps it out of scratch)
#define PLANE_PTR3(ps, idx)    ((idx) == 0 ? (ps).data[0] : (idx) == 1 ? (ps).data[1] : (ps).data[2])
#define PLANE_STRIDE3(ps, idx) ((idx) == 0 ? (ps).stride[0] : (idx) == 1 ? (ps).stride[1] : (ps).stride[2])
#define PLANE_WIDTH3(ps, idx)  ((idx) == 0 ? (ps).width[0] : (idx) == 1 ? (ps).width[1] : (ps).width[2])
#define PLANE_HEIGHT3(ps, idx) ((idx) == 0 ? (ps).height[0] : (idx) == 1 ? (ps).height[1] : (ps).height[2])

// ---- device helpers
// address spaces for loads whose kind the compiler cannot prove: global memory (global_load, not flat_load) and memory that is constant
// for the kernel's lifetime and read at a wave-uniform address (s_load)
#ifdef OHEVC_HIPEMU
#define OHEVC_GLOBAL_AS
#define OHEVC_CONST_AS
#else
#define OHEVC_GLOBAL_AS __attribute__((address_space(1)))
#define OHEVC_CONST_AS __attribute__((address_space(4)))
#endif

typedef short          s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int   u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int   u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename To, typename From>
__device__ __forceinline__ To bitcast(From v) { return __builtin_bit_cast(To, v); }

// acc + a.lo*b.lo + a.hi*b.hi on int16 halves: v_dot2c_i32_i16 (2 MACs per VALU op; exact, no saturation)
__device__ __forceinline__ int dot2_i16(unsigned a, unsigned b, int acc)
{
    return __builtin_amdgcn_sdot2(bitcast<s16x2>(a), bitcast<s16x2>(b), acc, false);
}

// dot2_i16_first, sat_pack_u8_i16: the two helpers written as gfx950 instructions (found on the include path, so that the
// host emulation used by the CPU tests can supply plain-C++ equivalents without touching this file)
}  // namespace ohevc
#include <ohevc_gfx950_ops.hpp>
namespace ohevc {

// The library's ONE place that looks at the process environment (runtime.hip: config()), read once.  Everything a decoder can choose per
// instance is an option of the context (ohevc_ctx_set_option) or of the sample back end (ohhip_options); kernel forms and executors that lost
// their A/B comparisons are selectable through include/ohevc_debug.h only (tests, the lab build).  What is left here is what belongs to the
// process: timeouts for slow (sanitizer) builds, start-up sizes, diagnosis output.
struct Config {
    int ref_wait_seconds = 20;        // OHEVC_REF_WAIT_SECONDS: how long a frame thread waits for another thread's frame end before it gives up
    int prewarm_kib = 3072;           // OHEVC_PREWARM_KIB: first size of a context's upload buffers, touched when the context is made (0: off)
    int issuer_threads = 4;           // OHEVC_ISSUER_THREADS: threads that issue asynchronous frame ends (ohevc_frame_end_async), per picture store
    int park_threads = 0;             // OHEVC_PARK_THREADS: issuer threads for PARKED frame ends (ohevc_frame_end_deferred); 0: the threads that unblock them issue them
    int picture_batch = 1;            // OHEVC_PICTURE_BATCH=0: every device picture its own allocation (AddressSanitizer / guard-page runs)
    const char *frames_token = nullptr;    // OHEVC_FRAMES_TOKEN: what the ranks of the sockets wire present to each other (set by the launcher)
    // OHEVC_TRACE=word[,word...]: diagnosis output on stderr
    bool trace_order = false, trace_timing = false, trace_ctb = false, trace_levels = false, trace_launches = false, trace_sao = false, trace_reg = false, trace_upload = false, trace_pin = false;
    bool profile_slots = false;       // "slots": cycle counters per table-slot family, printed when a context forgets its tables
    bool ctb_debug = false;           // "ctbdebug": the CTB executor's sync words after every launch
    int trace_at[3] = {-1, -1, -1};   // "at=plane:x:y": every recorded job whose block covers that sample
};
const Config &config();

// {clip_int16(lo), clip_int16(hi)} packed: v_cvt_pk_i16_i32
__device__ __forceinline__ unsigned sat_pack_i16(int lo, int hi)
{
    return bitcast<unsigned>(__builtin_amdgcn_cvt_pk_i16(lo, hi));
}

// per-half clamp(a + b, 0, maxv) with a, b packed int16 pairs (saturating add: v_pk_add_i16 clamp)
__device__ __forceinline__ unsigned add_clamp_px2(unsigned a, unsigned b, unsigned maxv2)
{
    s16x2 s = __builtin_elementwise_add_sat(bitcast<s16x2>(a), bitcast<s16x2>(b));
    s16x2 z = { 0, 0 };
    s = __builtin_elementwise_max(s, z);
    s = __builtin_elementwise_min(s, bitcast<s16x2>(maxv2));
    return bitcast<unsigned>(s);
}

// The same for 16-bit PIXELS that may hold anything up to 65535: clamp(px + res, 0, maxv) with px unsigned, res int16.
// Legal pictures never exceed maxv, but the reference's constrained-intra substitution can leave 0x8080 samples in a
// prediction (hevcpred_template.c:159-161 memsets BYTES to 128; intra PUs of another slice are "intra" yet never copied),
// and its transform_add then computes av_clip_pixel(dst + res) with dst read as uint16 (hevcdsp_template.c:45-111).
// Bias both halves by -32768, add with int16 saturation, undo the bias: below 0 / above 65535 saturate to the right ends.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned add_clamp_upx2(unsigned res, unsigned px, unsigned maxv2)
{
    s16x2 s = __builtin_elementwise_add_sat(bitcast<s16x2>(res), bitcast<s16x2>(px ^ 0x80008000u));
    u16x2 u = bitcast<u16x2>(bitcast<unsigned>(s) ^ 0x80008000u);
    u = __builtin_elementwise_min(u, bitcast<u16x2>(maxv2));
    return bitcast<unsigned>(u);
}

__host__ __device__ constexpr unsigned pack16(int lo, int hi)
{
    return (static_cast<unsigned>(lo) & 0xffffu) | ((static_cast<unsigned>(hi) & 0xffffu) << 16);
}

}  // namespace ohevc
