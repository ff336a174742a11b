// ohevc_gfx950_ops.hpp -- helpers spelt as single gfx950 instructions (inline assembly).  Included by common.hpp.
#pragma once

namespace ohevc {

// first term of a dot-product chain: a.lo*k.lo + a.hi*k.hi with NO accumulator input (VOP3P form, src2 = inline 0),
// so the chain needs no zero-initialising v_mov; the packed constant travels in an SGPR (s_mov on the scalar unit)
__device__ __forceinline__ int dot2_i16_first(unsigned a, unsigned kconst)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(kconst));
    return r;
}

// {sat_u8(x.i16[0]), sat_u8(x.i16[1])} in the low 16 bits: v_sat_pk_u8_i16 (clamp to [0,255] and pack in one op)
__device__ __forceinline__ unsigned sat_pack_u8_i16(unsigned x)
{
    unsigned r;
    asm("v_sat_pk_u8_i16_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// bytes sh .. sh + 3 of the eight bytes hi:lo (v_alignbyte_b32); four signed bytes against four signed bytes plus an accumulator
// (v_dot4_i32_i8); four small constants as the bytes of a dword
__device__ __forceinline__ unsigned align_bytes(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ int dot4_i8(unsigned a, unsigned b, int acc) { return __builtin_amdgcn_sdot4((int)a, (int)b, acc, false); }
// v_perm_b32: byte i of the result is picked by byte i of sel - 0..3: that byte of lo, 4..7: of hi, 0x0c: zero
__device__ __forceinline__ unsigned perm_b32(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ unsigned pack_i8x4(int a, int b, int c, int d)
{
    return ((unsigned)a & 0xffu) | (((unsigned)b & 0xffu) << 8) | (((unsigned)c & 0xffu) << 16) | ((unsigned)d << 24);
}

// acquire / release between workgroups of one XCD (they share an L2): drop what this CU's L1 holds; wait until this
// wavefront's stores have left for the L2
__device__ __forceinline__ void xcd_acquire() { asm volatile("buffer_inv sc1" ::: "memory"); }
__device__ __forceinline__ void xcd_release() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// memory operations are neither moved across this point by the compiler nor re-ordered around it by the scheduler: the ISSUE order of two
// groups of loads is what the code says (vector memory returns in order, so the order decides who waits for whom)
__device__ __forceinline__ void wait_all_but_6_loads() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }      // (diagnosis: ohevc_debug_intra_chain_clocks)
__device__ __forceinline__ void issue_order_fence() { asm volatile("" ::: "memory"); }

}  // namespace ohevc
