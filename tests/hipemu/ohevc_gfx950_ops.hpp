// ohevc_gfx950_ops.hpp of the TEST-ONLY host emulation: what the two single-instruction helpers of
// openhevc_amd/csrc/ohevc_gfx950_ops.hpp compute, in plain C++ (this directory precedes csrc/ on the emulator's include path).
#pragma once

namespace ohevc {

// v_dot2_i32_i16 with src2 = 0: a.lo * k.lo + a.hi * k.hi on signed 16-bit halves
static inline int dot2_i16_first(unsigned a, unsigned kconst)
{
    return (int)(short)(a & 0xffff) * (int)(short)(kconst & 0xffff) + (int)(short)(a >> 16) * (int)(short)(kconst >> 16);
}

// v_sat_pk_u8_i16: both signed 16-bit halves clamped to [0, 255], packed into the low 16 bits
static inline unsigned sat_pack_u8_i16(unsigned x)
{
    const int lo = (short)(x & 0xffff), hi = (short)(x >> 16);
    const unsigned a = lo < 0 ? 0 : lo > 255 ? 255 : lo, b = hi < 0 ? 0 : hi > 255 ? 255 : hi;
    return a | (b << 8);
}

static inline unsigned align_bytes(unsigned hi, unsigned lo, unsigned sh)
{
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    return (unsigned)(v >> (8 * (sh & 3)));
}
static inline int dot4_i8(unsigned a, unsigned b, int acc)
{
    for (int k = 0; k < 4; k++) acc += (int)(signed char)(a >> (8 * k)) * (int)(signed char)(b >> (8 * k));
    return acc;
}
static inline unsigned perm_b32(unsigned hi, unsigned lo, unsigned sel)
{
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 0xff;
        const unsigned b = s < 8 ? (unsigned)(v >> (8 * s)) & 0xff : s == 0x0c ? 0u : 0xffu;      // (the sign-replicating selectors 8..11 are not used)
        r |= b << (8 * i);
    }
    return r;
}
static inline unsigned pack_i8x4(int a, int b, int c, int d)
{
    return ((unsigned)a & 0xffu) | (((unsigned)b & 0xffu) << 8) | (((unsigned)c & 0xffu) << 16) | ((unsigned)d << 24);
}

// cache maintenance between workgroups: nothing to do on one coherent host memory
static inline void xcd_acquire() {}
static inline void xcd_release() {}
static inline void issue_order_fence() {}
static inline void wait_all_but_6_loads() {}

}  // namespace ohevc
