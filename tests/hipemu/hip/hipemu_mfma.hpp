// Wave-collective instructions of gfx950 the kernels use, emulated on gathered lane data (TEST-ONLY, see hip_runtime.h).
// The fragment layouts are the ones tools/probe_mfma_layout.py measured on an MI355X (profiles/r01z_mfma_layout_probe.json):
//   v_mfma_i32_32x32x32_i8   A[m = lane & 31][k = 16 (lane >> 5) + byte], B[k = 16 (lane >> 5) + byte][n = lane & 31],
//                            D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31] for accumulator register r
//   ds_read_b64_tr_b16       in each group of 16 lanes, lane l receives element l & 3 of the four int16 that lanes
//                            (l & 15) >> 2, + 4, + 8, + 12 of its group addressed
#pragma once

typedef int hipemu_v4i  __attribute__((ext_vector_type(4)));
typedef int hipemu_v16i __attribute__((ext_vector_type(16)));
typedef short hipemu_v4s __attribute__((ext_vector_type(4)));

static inline hipemu_v16i hipemu_mfma_i32_32x32x32_i8(hipemu_v4i a, hipemu_v4i b, hipemu_v16i c)
{
    signed char all_a[64][16], all_b[64][16];
    ::hipemu::wave_gather(&a, 16, all_a);
    ::hipemu::wave_gather(&b, 16, all_b);
    const int lane = ::hipemu::g_lane.flat & 63, n = lane & 31;
    hipemu_v16i d = c;
    for (int r = 0; r < 16; r++) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int acc = 0;
        for (int k = 0; k < 32; k++)
            acc += (int)all_a[(k >> 4) * 32 + m][k & 15] * (int)all_b[(k >> 4) * 32 + n][k & 15];
        d[r] = (int)((unsigned)d[r] + (unsigned)acc);
    }
    return d;
}
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, cbsz, abid, blgp) hipemu_mfma_i32_32x32x32_i8((a), (b), (c))

// v_mfma_i32_16x16x64_i8: A[m = lane & 15][slot group lane >> 4, 16 slots], B[same slots][n = lane & 15],
// D[m = 4 (lane >> 4) + r][n = lane & 15] (the dtype-independent 16x16 C/D map of gfx950, cdna_hip_programming.md "Fragment layout")
static inline hipemu_v4i hipemu_mfma_i32_16x16x64_i8(hipemu_v4i a, hipemu_v4i b, hipemu_v4i c)
{
    signed char all_a[64][16], all_b[64][16];
    ::hipemu::wave_gather(&a, 16, all_a);
    ::hipemu::wave_gather(&b, 16, all_b);
    const int lane = ::hipemu::g_lane.flat & 63, n = lane & 15;
    hipemu_v4i d = c;
    for (int r = 0; r < 4; r++) {
        const int m = 4 * (lane >> 4) + r;
        int acc = 0;
        for (int g = 0; g < 4; g++)
            for (int j = 0; j < 16; j++)
                acc += (int)all_a[g * 16 + m][j] * (int)all_b[g * 16 + n][j];
        d[r] = (int)((unsigned)d[r] + (unsigned)acc);
    }
    return d;
}
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, cbsz, abid, blgp) hipemu_mfma_i32_16x16x64_i8((a), (b), (c))

// v_mfma_i32_16x16x32_i8: as above with 8 slots per lane (A, B: 64 bits)
static inline hipemu_v4i hipemu_mfma_i32_16x16x32_i8(long a, long b, hipemu_v4i c)
{
    signed char all_a[64][8], all_b[64][8];
    ::hipemu::wave_gather(&a, 8, all_a);
    ::hipemu::wave_gather(&b, 8, all_b);
    const int lane = ::hipemu::g_lane.flat & 63, n = lane & 15;
    hipemu_v4i d = c;
    for (int r = 0; r < 4; r++) {
        const int m = 4 * (lane >> 4) + r;
        int acc = 0;
        for (int g = 0; g < 4; g++)
            for (int j = 0; j < 8; j++)
                acc += (int)all_a[g * 16 + m][j] * (int)all_b[g * 16 + n][j];
        d[r] = (int)((unsigned)d[r] + (unsigned)acc);
    }
    return d;
}
#define __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, cbsz, abid, blgp) hipemu_mfma_i32_16x16x32_i8((a), (b), (c))

static inline hipemu_v4s hipemu_ds_read_tr16_b64(const void *p)
{
    uint64_t mine = (uint64_t)(uintptr_t)p, all[64];
    ::hipemu::wave_gather(&mine, 8, all);
    const int lane = ::hipemu::g_lane.flat & 63, base = lane & ~15, sub = (lane & 15) >> 2;
    hipemu_v4s r;
    for (int j = 0; j < 4; j++) {
        short v;
        memcpy(&v, (const unsigned char *)(uintptr_t)all[base + 4 * j + sub] + 2 * (lane & 3), 2);
        r[j] = v;
    }
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16_b64((const void *)(p))
