// tu_generic.hpp -- the residual of one transform block of ANY kind by one wavefront, in direct matrix form, into wave-private LDS.
// Shared by the cross-component body of tu_kernels.hip and the CTB executor of ctb_kernels.hip (rarely-hot paths: generic on purpose).
#pragma once
#include "common.hpp"

namespace ohevc {

OHEVC_CONST_TABLE signed char kCos128[128] = {       // 64*sqrt(2)*cos(m*pi/64) as the standard rounds it (hevcdsp.c:879-944 spelt out)
    64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4,
    0, -4, -9, -13, -18, -22, -25, -31, -36, -38, -43, -46, -50, -54, -57, -61, -64, -67, -70, -73, -75, -78, -80, -82, -83, -85, -87, -88, -89, -90, -90, -90,
    -64, -90, -90, -90, -89, -88, -87, -85, -83, -82, -80, -78, -75, -73, -70, -67, -64, -61, -57, -54, -50, -46, -43, -38, -36, -31, -25, -22, -18, -13, -9, -4,
    0, 4, 9, 13, 18, 22, 25, 31, 36, 38, 43, 46, 50, 54, 57, 61, 64, 67, 70, 73, 75, 78, 80, 82, 83, 85, 87, 88, 89, 90, 90, 90 };
OHEVC_CONST_TABLE signed char kDst4[4][4] = { {29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29} };   // :170-203

#define CROSS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// residual of one dense N x N block of any kind into out[N*N] (wave-private LDS, row-major int16); tmp = same-size scratch
__device__ __forceinline__ void residual_generic(int kind, int log2, const int16_t *__restrict__ blk, int bit_depth, short *tmp, short *out, int lane)
{
    const int N = 1 << log2, NN = N * N, mask = N - 1;
    auto clip16 = [](int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; };
    if (kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4) {
        const bool dst = kind == OHEVC_TU_DST4;
        const int step = 32 >> log2;
        // inverse transform: out[k] = sum_j M[j][k] * in[j]; DCT M[j][k] = cos((j * step) * (2k + 1)), :210-301
        auto coef = [&](int j, int k) { return dst ? (int)kDst4[j][k] : (int)kCos128[(j * step * (2 * k + 1)) & 127]; };
        for (int o = lane; o < NN; o += 64) {              // pass 1: columns, shift 7
            const int k = o >> log2, c = o & mask;
            int acc = 0;
            for (int j = 0; j < N; j++) acc += coef(j, k) * (int)blk[j * N + c];
            tmp[o] = (short)clip16((acc + 64) >> 7);
        }
        CROSS_SYNC();
        const int shift = 20 - bit_depth, add = 1 << (shift - 1);
        for (int o = lane; o < NN; o += 64) {              // pass 2: rows
            const int r = o >> log2, k = o & mask;
            int acc = 0;
            for (int j = 0; j < N; j++) acc += coef(j, k) * (int)tmp[r * N + j];
            out[o] = (short)clip16((acc + add) >> shift);
        }
    } else if (kind == OHEVC_TU_DC) {                      // :303-316
        // at BIT_DEPTH 14 the reference's `1 << (shift - 1)` has a negative count (hevcdsp_template.c:307-308); its gcc build folds
        // that to 0 (pinned by tests/test_oracle_vs_reference.py against oracle/_ref), which is also the natural reading of shift 0
        const int shift = 14 - bit_depth, add = shift > 0 ? 1 << (shift - 1) : 0;
        const int v = ((((int)blk[0] + 1) >> 1) + add) >> shift;
        for (int o = lane; o < NN; o += 64) out[o] = (short)v;
    } else {                                               // transform_skip :139-163, transquant bypass, + rdpcm :114-136
        const bool skip = kind == OHEVC_TU_SKIP || kind == OHEVC_TU_SKIP_RDPCM_H || kind == OHEVC_TU_SKIP_RDPCM_V;
        const bool vert = kind == OHEVC_TU_SKIP_RDPCM_V || kind == OHEVC_TU_BYPASS_RDPCM_V;
        const bool horz = kind == OHEVC_TU_SKIP_RDPCM_H || kind == OHEVC_TU_BYPASS_RDPCM_H;
        const int shift = 15 - bit_depth - log2;
        for (int o = lane; o < NN; o += 64) {
            int v = blk[o];
            if (skip) v = shift > 0 ? (v + (1 << (shift - 1))) >> shift : (int)(short)(v << -shift);
            out[o] = (short)v;
        }
        CROSS_SYNC();
        if ((vert || horz) && lane < N) {                  // running sums with the reference's int16 wrap-around
            int acc = 0;
            for (int i = 0; i < N; i++) {
                const int o = horz ? lane * N + i : i * N + lane;
                acc = (int)(short)(acc + out[o]);
                out[o] = (short)acc;
            }
        }
    }
    CROSS_SYNC();
}


}  // namespace ohevc
