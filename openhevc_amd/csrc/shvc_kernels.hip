// shvc_kernels.hip -- SHVC inter-layer up-sampling: the 13 upsample_* slots of HEVCDSPContext (hevcdsp.h:106-123).
//
// The reference resamples the base-layer picture into the enhancement layer's inter-layer reference picture either in one
// go (upsample_base_layer_frame, hevcdsp_template.c:2165-2438) or, in the shipped build (ACTIVE_PU_UPSAMPLING, hevc.h:117),
// CTB by CTB on demand through upsample_filter_block_{luma,cr}_{h,v}[idx] and the emulated_edge_up_{h,v} helpers
// (hevc_filter.c:1175-1310).  Both are the same separable filter: a horizontal pass into int16, a vertical pass with a fixed
// 12-bit rounding (N_SHIFT, hevcdsp.h:40-41), 16 phases, 8 taps luma / 4 taps chroma (H.265 tables H.1 / H.2); they only
// differ in how a column / row finds its base-layer position and phase (general formula, or the fixed x2 / x1.5 patterns of
// the idx 1 / 2 slots).  Here that part is a per-column and per-row MAP built on the host (ohevc_upsample_make_maps, below:
// the reference's formulas, cited there); the kernel is a gather-filter over a sliding window of horizontally filtered rows and never sees a scale
// factor.  Coordinates are clamped instead of reading emulated edges.  Bytes per unit: P per written sample + the
// base-layer picture once (it is re-read through L2: 64 taps per luma sample, 16 per chroma sample).
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "common.hpp"

namespace ohevc {

__constant__ signed char kUpLuma[16][8] = {
    {  0, 0,   0, 64,  0,   0, 0,  0 }, {  0, 1,  -3, 63,  4,  -2, 1,  0 }, { -1, 2,  -5, 62,  8,  -3, 1,  0 }, { -1, 3,  -8, 60, 13,  -4, 1,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 }, { -1, 4, -11, 52, 26,  -8, 3, -1 }, { -1, 3,  -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { -1, 4, -10, 34, 45, -11, 4, -1 }, { -1, 4, -10, 31, 47,  -9, 3, -1 }, { -1, 3,  -8, 26, 52, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 }, {  0, 1,  -4, 13, 60,  -8, 3, -1 }, {  0, 1,  -3,  8, 62,  -5, 2, -1 }, {  0, 1,  -2,  4, 63,  -3, 1,  0 } };
__constant__ signed char kUpChroma[16][4] = {
    {  0, 64,  0,  0 }, { -2, 62,  4,  0 }, { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
    { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, {  0,  4, 62, -2 } };

__constant__ unsigned kUpInv[32] = { 0x0u, 0xffffffffu, 0x80000000u, 0x55555556u, 0x40000000u, 0x33333334u, 0x2aaaaaabu, 0x24924925u, 0x20000000u, 0x1c71c71du, 0x1999999au, 0x1745d175u, 0x15555556u, 0x13b13b14u, 0x12492493u, 0x11111112u, 0x10000000u, 0xf0f0f10u, 0xe38e38fu, 0xd79435fu, 0xccccccdu, 0xc30c30du, 0xba2e8bbu, 0xb21642du, 0xaaaaaabu, 0xa3d70a4u, 0x9d89d8au, 0x97b425fu, 0x924924au, 0x8d3dcb1u, 0x8888889u, 0x8421085u };      // ceil(2^32 / d), d = 2 .. 31 (staging loop of the tile kernel)

// One thread produces ROWS consecutive output rows of one column.  Consecutive output rows read base-layer rows that advance by
// at most one per row (the enhancement layer is never smaller than the base layer), so the horizontally filtered values live
// in a sliding window of TAPS registers: TAPS + ROWS - 1 horizontal filters per thread instead of TAPS * ROWS.  The window is
// indexed by base-layer row, not by output row, so any monotonic row map works (a jump forces a refill).
// Round 2: the base-layer samples a workgroup's 64 x (4 ROWS) output tile reads - at most (4 ROWS + TAPS) x (64 + TAPS) of them - are
// staged in LDS once (clamped coordinates resolved there) instead of being gathered from global memory TAPS times per filtered value
// (120 one-sample loads per thread before, ~12 now); a tile whose maps jump (window larger than the LDS tile) takes the gather form.
constexpr int UP_WC = 80, UP_WR = 48;        // LDS window: columns x rows (x1 .. x2 scaling needs 72 x 40)
template <typename Pixel, int TAPS, int ROWS>
__global__ __launch_bounds__(256) void upsample_kernel(ohevc_plane dst, ohevc_plane src, const ohevc_upsample_tap *__restrict__ cols,
                                                       const int16_t *__restrict__ col_of, const ohevc_upsample_tap *__restrict__ rows,
                                                       int src_cols, int src_rows, int bit_depth)
{
    constexpr int HALF = TAPS / 2 - 1;
    __shared__ Pixel win_lds[UP_WR][UP_WC];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS;
    const unsigned char *sbase = static_cast<const unsigned char *>(src.data);
    // ---- the tile's base-layer window (wave-uniform arithmetic on the maps)
    const int tx0 = blockIdx.x * 64, tx1 = min(tx0 + 63, dst.width - 1), ty0 = blockIdx.y * 4 * ROWS, ty1 = min(ty0 + 4 * ROWS - 1, dst.height - 1);
    const int cmin = cols[col_of[tx0]].pos - HALF, cmax = cols[col_of[tx1]].pos - HALF + TAPS - 1;
    const int rmin = rows[ty0].pos - HALF, rmax = rows[ty1].pos - HALF + TAPS - 1;
    const bool staged = cmax - cmin < UP_WC && rmax - rmin < UP_WR && cmax >= cmin && rmax >= rmin;
    if (staged) {
        const int wc = cmax - cmin + 1, wr = rmax - rmin + 1;
        for (int i = threadIdx.x; i < wc * wr; i += 256) {
            const int r = i / wc, cc = i - r * wc;
            int ry = rmin + r, rx = cmin + cc;
            ry = ry < 0 ? 0 : ry > src_rows - 1 ? src_rows - 1 : ry;
            rx = rx < 0 ? 0 : rx > src_cols - 1 ? src_cols - 1 : rx;
            win_lds[r][cc] = reinterpret_cast<const Pixel *>(sbase + (size_t)ry * src.stride)[rx];
        }
    }
    __syncthreads();
    if (x >= dst.width || y0 >= dst.height) return;
    const ohevc_upsample_tap tc = cols[col_of[x]];
    int cx[TAPS], ch[TAPS];                                    // clamped source columns and this column's horizontal taps
#pragma unroll
    for (int k = 0; k < TAPS; k++) {
        const int rx = tc.pos - HALF + k;
        cx[k] = rx < 0 ? 0 : rx > src_cols - 1 ? src_cols - 1 : rx;
        ch[k] = TAPS == 8 ? (int)kUpLuma[tc.phase][k] : (int)kUpChroma[tc.phase][k];
    }
    const int lc = tc.pos - HALF - cmin;                       // this column's first tap inside the staged window
    auto hfilt = [&](int row) {                                 // horizontal pass of one base-layer row (clamped), as int16
        int h = 0;
        if (staged && row >= rmin && row <= rmax && lc >= 0 && lc + TAPS <= UP_WC) {
            const Pixel *p = &win_lds[row - rmin][lc];
#pragma unroll
            for (int k = 0; k < TAPS; k++) h += ch[k] * (int)p[k];
        } else {
            const int ry = row < 0 ? 0 : row > src_rows - 1 ? src_rows - 1 : row;
            const Pixel *p = reinterpret_cast<const Pixel *>(sbase + (size_t)ry * src.stride);
#pragma unroll
            for (int k = 0; k < TAPS; k++) h += ch[k] * (int)p[cx[k]];
        }
        return (int)(short)h;                                   // the reference keeps this pass in int16: it wraps above 8 bit
    };
    int win[TAPS], base = 0x40000000;                           // win[k] = hfilt(base + k); no window yet
    const int maxv = (1 << bit_depth) - 1;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int y = y0 + r;
        if (y >= dst.height) break;
        const ohevc_upsample_tap tr = rows[y];                  // wave-uniform (one row of 64 columns per wavefront)
        const int first = tr.pos - HALF;
        if (first == base + 1) {                                // the common step: slide by one row
#pragma unroll
            for (int k = 0; k + 1 < TAPS; k++) win[k] = win[k + 1];
            win[TAPS - 1] = hfilt(first + TAPS - 1);
            base = first;
        } else if (first != base) {                             // first row of the strip, or a jump in the row map
#pragma unroll
            for (int k = 0; k < TAPS; k++) win[k] = hfilt(first + k);
            base = first;
        }
        int acc = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++) acc += (TAPS == 8 ? (int)kUpLuma[tr.phase][k] : (int)kUpChroma[tr.phase][k]) * win[k];
        int v = (acc + (1 << 11)) >> 12;                        // I_OFFSET / N_SHIFT, hevcdsp.h:40-41
        v = v < 0 ? 0 : v > maxv ? maxv : v;
        *(reinterpret_cast<Pixel *>(static_cast<unsigned char *>(dst.data) + (size_t)y * dst.stride) + x) = (Pixel)v;
    }
}

// ---- round 4: the tile form.  The kernel above gives every thread a column strip and lets it run BOTH passes on its own: 11 horizontal
// filters of 8 one-byte LDS reads for 8 output rows (the strips of a workgroup overlap by 7 base-layer rows each), a sliding register
// window, ~30 scalar multiply-adds per output sample - it was bound by instruction issue at 4.6 % of the HBM rate (profiles/r03end_*).
// Here a workgroup of 256 threads owns a tile of 64 columns x 64 output rows and the passes are separate:
//   1. the tile's base-layer window (at most 72 rows x 76 columns) goes to LDS once, four samples per load;
//   2. every (base row, output column) pair is filtered horizontally ONCE: the eight samples come out of three aligned LDS dwords
//      (v_alignbyte) and meet the taps in two v_dot4_i32_i8 (8 bit: samples biased by -128, the taps of a phase sum to 64) / four
//      v_dot2_i32_i16 (above 8 bit); the int16 results lie column-major in LDS, a column's rows next to each other;
//   3. an output sample is four LDS dwords (its column's eight consecutive rows: the row map is wave-uniform, so is their alignment) and
//      four v_dot2_i32_i16 against the row's packed taps.
// A tile whose maps jump (window larger than the LDS arrays) is left to the gather form above.
constexpr int UPT_ROWS = 64, UPT_WR = 72, UPT_WC = 72 + 8, UPT_HS = 2 * UPT_WR + 4;     // H column stride in bytes: 37 dwords, odd (x1 needs 64 + 7 rows)
// The workgroup's LDS: the window, the horizontal pass's columns, the tile's row-map entries and the 16 phases' taps (as int16 pairs, and for
// 8-bit samples as int8 quads): a row of the vertical pass otherwise costs two dependent scalar loads from memory (rows[y], then the constant
// table) - the kernel was bound by that latency.  Declared by the kernel, not by the body: the three-plane kernel runs the body with 8 and
// with 4 taps and must not pay for two copies.
template <typename Pixel> struct UpTileLds {
    __attribute__((aligned(16))) unsigned char win[UPT_WR * UPT_WC * (int)sizeof(Pixel)];
    __attribute__((aligned(16))) unsigned char hcol[64 * UPT_HS + 64];      // (+ 64: the matrix-core vertical pass reads whole 32-row pieces, the last column's beyond its 72 rows - against zero taps)
    ohevc_upsample_tap srow[UPT_ROWS];
    __attribute__((aligned(16))) unsigned stap16[16][4];
    unsigned stap8[16][2];
    unsigned vtap[16][11];              // matrix-core vertical pass: a phase's taps as bytes 16 .. 23 of 44, zeros around them
};
typedef int up_v4i __attribute__((ext_vector_type(4)));
typedef int up_v16i __attribute__((ext_vector_type(16)));
template <typename Pixel, int TAPS, bool MFMA>
__device__ __forceinline__ void upsample_tile_body(UpTileLds<Pixel> &lds, const ohevc_plane &dst, const ohevc_plane &src, const ohevc_upsample_tap *__restrict__ cols,
                                                   const int16_t *__restrict__ col_of, const ohevc_upsample_tap *__restrict__ rows,
                                                   int src_cols, int src_rows, int bit_depth, int tile_x, int tile_y)
{
    constexpr int HALF = TAPS / 2 - 1, P = (int)sizeof(Pixel);
    unsigned char *const win = lds.win, *const hcol = lds.hcol;
    ohevc_upsample_tap *const srow = lds.srow;
    unsigned (*const stap16)[4] = lds.stap16;
    unsigned (*const stap8)[2] = lds.stap8;
    const int tid = threadIdx.x, lx = tid & 63, part = tid >> 6;
    const int tx0 = tile_x * 64, tx1 = min(tx0 + 63, dst.width - 1), ty0 = tile_y * UPT_ROWS, ty1 = min(ty0 + UPT_ROWS - 1, dst.height - 1);
    // the tile's base-layer window; its first column rounded down to a multiple of four samples
    const int cmin = (cols[col_of[tx0]].pos - HALF) & ~3, cmax = cols[col_of[tx1]].pos - HALF + TAPS - 1;
    const int rmin = rows[ty0].pos - HALF, rmax = rows[ty1].pos - HALF + TAPS - 1;
    const int wc = cmax - cmin + 1, wr = rmax - rmin + 1;
    const unsigned char *sbase = static_cast<const unsigned char *>(src.data);
    const int x = min(tx0 + lx, dst.width - 1);
    const ohevc_upsample_tap tc = cols[col_of[x]];                  // (asked for here: two dependent loads that the window's staging hides)
    if (wc > UPT_WC - 4 || wr > UPT_WR || wc <= 0 || wr <= 0) {
        // maps that jump (never the reference's x1 .. x2 patterns): every output sample on its own, straight from memory
        const int xg = tx0 + lx;
        if (xg >= dst.width) return;
        const int maxv = (1 << bit_depth) - 1;
        for (int k = 0; k < UPT_ROWS / 4; k++) {
            const int y = ty0 + part * (UPT_ROWS / 4) + k;
            if (y >= dst.height) break;
            const ohevc_upsample_tap tr = rows[y];
            int acc = 1 << 11;
            for (int q = 0; q < TAPS; q++) {
                int ry = tr.pos - HALF + q;
                ry = ry < 0 ? 0 : ry > src_rows - 1 ? src_rows - 1 : ry;
                const Pixel *prow = reinterpret_cast<const Pixel *>(sbase + (size_t)ry * src.stride);
                int h = 0;
                for (int t = 0; t < TAPS; t++) {
                    int rx = tc.pos - HALF + t;
                    rx = rx < 0 ? 0 : rx > src_cols - 1 ? src_cols - 1 : rx;
                    h += (TAPS == 8 ? (int)kUpLuma[tc.phase][t] : (int)kUpChroma[tc.phase][t]) * (int)prow[rx];
                }
                acc += (TAPS == 8 ? (int)kUpLuma[tr.phase][q] : (int)kUpChroma[tr.phase][q]) * (int)(short)h;
            }
            int v = acc >> 12;
            v = v < 0 ? 0 : v > maxv ? maxv : v;
            *(reinterpret_cast<Pixel *>(static_cast<unsigned char *>(dst.data) + (size_t)y * dst.stride) + xg) = (Pixel)v;
        }
        return;
    }
    if (tid < UPT_ROWS) srow[tid] = rows[min(ty0 + tid, dst.height - 1)];
    else if (tid >= 64 && tid < 128) {
        const int ph = (tid - 64) >> 2, q = (tid - 64) & 3;
        const int a = 2 * q < TAPS ? (TAPS == 8 ? (int)kUpLuma[ph][2 * q] : (int)kUpChroma[ph][2 * q]) : 0;
        const int b = 2 * q + 1 < TAPS ? (TAPS == 8 ? (int)kUpLuma[ph][2 * q + 1] : (int)kUpChroma[ph][2 * q + 1]) : 0;
        stap16[ph][q] = pack16(a, b);
    } else if (tid >= 128 && tid < 160) {
        const int ph = (tid - 128) >> 1, q = (tid - 128) & 1;
        const signed char *t8 = TAPS == 8 ? kUpLuma[ph] : kUpChroma[ph];
        stap8[ph][q] = 4 * q < TAPS ? pack_i8x4(t8[4 * q], t8[4 * q + 1], t8[4 * q + 2], t8[4 * q + 3]) : 0u;
    } else if (MFMA && tid >= 160 && tid < 176) {
        const int ph = tid - 160;
        const signed char *t8 = TAPS == 8 ? kUpLuma[ph] : kUpChroma[ph];
#pragma unroll
        for (int q = 0; q < 11; q++) lds.vtap[ph][q] = 0u;
        lds.vtap[ph][4] = pack_i8x4(t8[0], t8[1], t8[2], t8[3]);
        if (TAPS == 8) lds.vtap[ph][5] = pack_i8x4(t8[4], t8[5], t8[6], t8[7]);
    }
    // ---- 1. the window, four samples per thread and step.  8-bit samples are stored as sample - 128 (the horizontal pass multiplies signed
    // bytes: one xor per four samples here instead of two per filtered sample there).  (A 16 x 16 thread layout without the division
    // ran slower, r4u2: a third of its lanes idle at the x2 window's 11 dwords per row.)
    {
        const int wc4 = (wc + 3) >> 2;
        const bool inside = cmin >= 0 && cmin + 4 * wc4 <= src_cols;
        // (i / wc4 as a multiplication: exact for i < 8192, wc4 < 64 - the window is at most 20 x 72 - where the division is ~25 instructions)
        const unsigned inv = kUpInv[wc4 & 31];                     // (wc4 <= 20: the window is at most 80 columns)
        for (int i = tid; i < wc4 * wr; i += 256) {
            const int r = wc4 == 1 ? i : (int)(((unsigned long long)(unsigned)i * inv) >> 32), c4 = i - r * wc4;
            int ry = rmin + r;
            ry = ry < 0 ? 0 : ry > src_rows - 1 ? src_rows - 1 : ry;
            const unsigned char *srow_p = sbase + __umul24((unsigned)ry, (unsigned)src.stride);
            Pixel v[4];
            if (inside) {
                if constexpr (P == 1) *reinterpret_cast<unsigned *>(v) = *reinterpret_cast<const unsigned *>(srow_p + cmin + 4 * c4);
                else                  *reinterpret_cast<u32x2 *>(v) = *reinterpret_cast<const u32x2 *>(srow_p + (size_t)(cmin + 4 * c4) * 2);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int rx = cmin + 4 * c4 + k;
                    rx = rx < 0 ? 0 : rx > src_cols - 1 ? src_cols - 1 : rx;
                    v[k] = reinterpret_cast<const Pixel *>(srow_p)[rx];
                }
            }
            if constexpr (P == 1) *reinterpret_cast<unsigned *>(win + r * UPT_WC + 4 * c4) = *reinterpret_cast<const unsigned *>(v) ^ 0x80808080u;
            else                  *reinterpret_cast<u32x2 *>(win + (r * UPT_WC + 4 * c4) * 2) = *reinterpret_cast<const u32x2 *>(v);
        }
    }
    __syncthreads();
    // ---- 2. horizontal pass: thread = (output column lx, base rows part, part + 4, ...)
    {
        const int lc = tc.pos - HALF - cmin;                       // first tap's column inside the window (>= 0)
        unsigned tp[4];                                             // the phase's taps, packed
        if constexpr (P == 1) { tp[0] = stap8[tc.phase][0]; tp[1] = stap8[tc.phase][1]; tp[2] = tp[3] = 0u; }
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) tp[k] = stap16[tc.phase][k];
        }
        for (int r = part; r < wr; r += 4) {
            int h;
            if constexpr (P == 1) {
                const unsigned *w32 = reinterpret_cast<const unsigned *>(win + r * UPT_WC) + (lc >> 2);
                const unsigned a0 = w32[0], a1 = w32[1], a2 = TAPS == 8 ? w32[2] : 0u;
                const unsigned sh = (unsigned)(lc & 3);
                h = dot4_i8(align_bytes(a1, a0, sh), tp[0], 128 * 64);       // (the window holds samples - 128 as int8; the taps sum to 64)
                if constexpr (TAPS == 8) h = dot4_i8(align_bytes(a2, a1, sh), tp[1], h);
            } else {
                const unsigned *w32 = reinterpret_cast<const unsigned *>(win + (r * UPT_WC) * 2) + (lc >> 1);
                const bool odd = lc & 1;
                unsigned a[5];
#pragma unroll
                for (int k = 0; k < TAPS / 2 + 1; k++) a[k] = w32[k];
                h = 0;
#pragma unroll
                for (int k = 0; k < TAPS / 2; k++) h = dot2_i16(odd ? align_bytes(a[k + 1], a[k], 2u) : a[k], tp[k], h);
            }
            *reinterpret_cast<short *>(hcol + lx * UPT_HS + 2 * r) = (short)h;      // the reference keeps this pass in int16: it wraps above 8 bit
        }
    }
    __syncthreads();
    if constexpr (MFMA) {
        // ---- 3 (round 6). The vertical pass on the matrix cores.  For 32 output rows the pass is a product with a BAND matrix: out[col][row] =
        // sum_k H[col][k] * V[k][row], V[k][row] = tap (k - first(row)) of the row's phase (hevcdsp_template.c:1835-1953: the same eight / four
        // multiply-adds per sample, the other 24 / 28 products of a row are zeros the matrix cores do not charge for).  A wavefront takes a
        // 32-column x 32-row quarter of the tile: A = its columns of the horizontal pass (int16, column-major in LDS: 16 consecutive rows of
        // a column are 8 dwords) split into high and low bytes the way the 32x32 inverse transform splits its coefficients (tu_kernels.hip:
        // the low byte biased by -128, made good by 128 x the taps' sum = 128 x 64), B = the rows' taps at their places, built per lane from
        // a 40-byte padded copy of the phase's taps.  D[col][row] puts four NEIGHBOURING columns of a row into one lane: one store per four
        // samples.  32 output rows reach over at most 39 base rows (x1; 23 at x2, 29 at x1.5): one or two K steps of 32.  22 vector
        // instructions per output sample become ~9.
        // (everything that steers the matrix instructions is wave-uniform AND in scalar registers - readfirstlane: the matrix cores do not look
        // at the execution mask, a K step "skipped" behind a vector condition runs all the same, on registers nobody loaded)
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cb = wave & 1, rb = wave >> 1, n = lane & 31, h = lane >> 5;
        auto first_row = [&](int half) { return __builtin_amdgcn_readfirstlane(((int)srow[32 * half].pos - HALF - rmin) & ~1); };      // (even: 16 rows of a column start on a dword)
        auto rows_spanned = [&](int half) { return __builtin_amdgcn_readfirstlane((int)srow[32 * half + 31].pos - HALF - rmin + TAPS) - first_row(half); };
        const int kbase = first_row(rb), span = rows_spanned(rb);
        // (the whole workgroup takes this form or the dot-product form below: both row halves of the tile must fit two K steps)
        const bool mfma_done = rows_spanned(0) <= 64 && rows_spanned(1) <= 64;
        constexpr int TS = 64 * P + 4;                          // row stride of the output tile in LDS: 17 / 33 dwords, odd
        unsigned char *const tile = win;
        static_assert(64 * TS <= UPT_WR * UPT_WC * P, "the output tile fits the window's LDS");
        if (mfma_done) {
            const int nk = span > 32 ? 2 : 1;
            const ohevc_upsample_tap tr = srow[32 * rb + n];
            const int f = (int)tr.pos - HALF - rmin;
            up_v4i ahi[2], alo[2], bv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                if (ks < nk) {
                    const unsigned *hc = reinterpret_cast<const unsigned *>(hcol + (32 * cb + n) * UPT_HS + 2 * (kbase + 32 * ks + 16 * h));
                    unsigned w[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) w[q] = hc[q];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ahi[ks][q] = (int)perm_b32(w[2 * q + 1], w[2 * q], 0x07050301u);
                        alo[ks][q] = (int)(perm_b32(w[2 * q + 1], w[2 * q], 0x06040200u) ^ 0x80808080u);
                    }
                    // bytes s .. s + 15 of the padded taps (the taps are bytes 16 .. 23): s = 16 + (first row of this lane's piece - first tap row)
                    int sft = kbase + 32 * ks + 16 * h - f;
                    sft = 16 + (sft < -16 ? -16 : sft > 8 ? 8 : sft);
                    const unsigned *vt = &lds.vtap[tr.phase][sft >> 2];
                    const unsigned t0 = vt[0], t1 = vt[1], t2 = vt[2], t3 = vt[3], t4 = vt[4], sh = (unsigned)sft & 3u;
                    bv[ks][0] = (int)align_bytes(t1, t0, sh); bv[ks][1] = (int)align_bytes(t2, t1, sh);
                    bv[ks][2] = (int)align_bytes(t3, t2, sh); bv[ks][3] = (int)align_bytes(t4, t3, sh);
                }
            }
            const up_v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            up_v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[0], bv[0], zero, 0, 0, 0);
            if (nk > 1) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[1], bv[1], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = (d[r] << 8) + (128 * 64 + (1 << 11));      // + the low bytes' bias, + I_OFFSET (hevcdsp.h:40-41)
            d = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[0], bv[0], d, 0, 0, 0);
            if (nk > 1) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[1], bv[1], d, 0, 0, 0);
            // D[col][row] leaves a lane four neighbouring samples of ONE row and a wavefront 32 rows: stored from here a store instruction would
            // touch 8 bytes of each of 32 lines (measured: the pass ran slower than the dot products it replaced, profiles/r6zf_*).  The quarter
            // tiles meet in LDS instead (the window of stage 1 is free since the second barrier) and leave as whole rows, 16 / 32 bytes a thread.
            {
                const int maxv = (1 << bit_depth) - 1;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    int v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int t = d[4 * g + j] >> 12; v[j] = t < 0 ? 0 : t > maxv ? maxv : t; }      // N_SHIFT
                    unsigned char *tp = tile + (32 * rb + n) * TS + (32 * cb + 8 * g + 4 * h) * P;
                    if constexpr (P == 1) *reinterpret_cast<unsigned *>(tp) = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
                    else { reinterpret_cast<unsigned *>(tp)[0] = (unsigned)v[0] | ((unsigned)v[1] << 16); reinterpret_cast<unsigned *>(tp)[1] = (unsigned)v[2] | ((unsigned)v[3] << 16); }
                }
            }
        }
        if (mfma_done) {
            __syncthreads();
            // thread -> (row, 16-byte piece of it); 16-bit samples: two pieces
            const int row = tid >> 2, piece = tid & 3, y = ty0 + row;
            const bool aligned = ((reinterpret_cast<uintptr_t>(dst.data) | (unsigned)dst.stride) & 15) == 0;
            if (y < dst.height) {
                unsigned char *orow = static_cast<unsigned char *>(dst.data) + __umul24((unsigned)y, (unsigned)dst.stride) + (unsigned)tx0 * (unsigned)P;
#pragma unroll
                for (int part2 = 0; part2 < P; part2++) {
                    const int b0 = 16 * (piece + 4 * part2);    // byte offset inside the tile row
                    const unsigned *tp = reinterpret_cast<const unsigned *>(tile + row * TS + b0);
                    const u32x4 o = u32x4{ tp[0], tp[1], tp[2], tp[3] };
                    const int x0 = tx0 + b0 / P;                // first sample of the piece
                    if (aligned && x0 + 16 / P <= dst.width) {
                        *reinterpret_cast<u32x4 *>(orow + b0) = o;
                    } else {
                        const unsigned ow[4] = { o.x, o.y, o.z, o.w };
#pragma unroll
                        for (int e = 0; e < 16 / P; e++)
                            if (x0 + e < dst.width) {
                                if constexpr (P == 1) orow[b0 + e] = (unsigned char)(ow[e >> 2] >> (8 * (e & 3)));
                                else reinterpret_cast<unsigned short *>(orow + b0)[e] = (unsigned short)(ow[e >> 1] >> (16 * (e & 1)));
                            }
                    }
                }
            }
            return;
        }
        // (a row map that reaches over more than 64 base rows in 32 output rows: never the reference's x1 .. x2 patterns - the dot-product form below)
    }
    // ---- 3. vertical pass: thread = (four neighbouring output columns, four consecutive output rows).  One store per four samples (a lane
    // per column stored single bytes: 64 bytes per store instruction of a wavefront), and what depends on the row only - the row's map
    // entry, its taps, the address - once per four samples.
    const int lq = tid & 15, rg = tid >> 4, xq = tx0 + 4 * lq;
    if (xq >= dst.width) return;
    const int maxv = (1 << bit_depth) - 1;
    const bool whole = xq + 4 <= dst.width && ((reinterpret_cast<uintptr_t>(dst.data) | (unsigned)dst.stride) & 3) == 0;
    unsigned a[4][5] = {};
    int have = -1;                                                  // the first row a[][] holds: consecutive output rows mostly share it
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int yy = rg * 4 + k, y = ty0 + yy;
        if (y >= dst.height) break;
        const ohevc_upsample_tap tr = srow[yy];
        const int f = tr.pos - HALF - rmin;                         // first of the TAPS rows
        if (f != have) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned *hc = reinterpret_cast<const unsigned *>(hcol + (4 * lq + j) * UPT_HS) + (f >> 1);
#pragma unroll
                for (int q = 0; q < TAPS / 2 + 1; q++) a[j][q] = hc[q];
#pragma unroll
                for (int q = 0; q < TAPS / 2; q++) a[j][q] = align_bytes(a[j][q + 1], a[j][q], (unsigned)(f & 1) * 2u);      // (no branch: a shift of 0 or 2 bytes)
            }
            have = f;
        }
        const u32x4 tq = *reinterpret_cast<const u32x4 *>(&stap16[tr.phase][0]);       // the row's four tap pairs in one LDS read
        const unsigned tqa[4] = { tq.x, tq.y, tq.z, tq.w };
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int acc = 1 << 11;                                      // I_OFFSET, hevcdsp.h:40-41
#pragma unroll
            for (int q = 0; q < TAPS / 2; q++) acc = dot2_i16(a[j][q], tqa[q], acc);
            acc >>= 12;                                             // N_SHIFT
            v[j] = acc < 0 ? 0 : acc > maxv ? maxv : acc;
        }
        unsigned char *out = static_cast<unsigned char *>(dst.data) + (__umul24((unsigned)y, (unsigned)dst.stride) + (unsigned)xq * (unsigned)P);
        if (whole) {
            if constexpr (P == 1) *reinterpret_cast<unsigned *>(out) = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
            else                  *reinterpret_cast<u32x2 *>(out) = u32x2{ (unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16) };
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (xq + j < dst.width) reinterpret_cast<Pixel *>(out)[j] = (Pixel)v[j];
        }
    }
}

template <typename Pixel, int TAPS, bool MFMA>
__global__ __launch_bounds__(256) void upsample_tile_kernel(ohevc_plane dst, ohevc_plane src, const ohevc_upsample_tap *__restrict__ cols,
                                                            const int16_t *__restrict__ col_of, const ohevc_upsample_tap *__restrict__ rows,
                                                            int src_cols, int src_rows, int bit_depth)
{
    __shared__ UpTileLds<Pixel> lds;
    upsample_tile_body<Pixel, TAPS, MFMA>(lds, dst, src, cols, col_of, rows, src_cols, src_rows, bit_depth, (int)blockIdx.x, (int)blockIdx.y);
}

// The three planes of an inter-layer picture in ONE launch (a two-layer decode made three per picture, 5.5-8.5 us each): the luma tiles,
// then the tiles of the two chroma planes; a workgroup finds its plane from its number.
struct UpPlaneArgs {
    ohevc_plane dst, src;
    const ohevc_upsample_tap *cols, *rows;
    const int16_t *col_of;
    int src_cols, src_rows, tiles_x, tiles;
};
struct UpPictureArgs { UpPlaneArgs pl[3]; };
template <typename Pixel, bool MFMA>
__global__ __launch_bounds__(256) void upsample_tile3_kernel(UpPictureArgs a, int bit_depth)
{
    __shared__ UpTileLds<Pixel> lds;
    int t = (int)blockIdx.x;
    if (t < a.pl[0].tiles) {
        const UpPlaneArgs &q = a.pl[0];
        upsample_tile_body<Pixel, 8, MFMA>(lds, q.dst, q.src, q.cols, q.col_of, q.rows, q.src_cols, q.src_rows, bit_depth, t % q.tiles_x, t / q.tiles_x);
        return;
    }
    t -= a.pl[0].tiles;
    const int c = t < a.pl[1].tiles ? 1 : 2;
    if (c == 2) t -= a.pl[1].tiles;
    // (chosen with selects, not through an index: an indexed by-value argument goes to scratch memory)
    const ohevc_plane dst = c == 1 ? a.pl[1].dst : a.pl[2].dst, src = c == 1 ? a.pl[1].src : a.pl[2].src;
    const ohevc_upsample_tap *cols = c == 1 ? a.pl[1].cols : a.pl[2].cols, *rows = c == 1 ? a.pl[1].rows : a.pl[2].rows;
    const int16_t *col_of = c == 1 ? a.pl[1].col_of : a.pl[2].col_of;
    const int src_cols = c == 1 ? a.pl[1].src_cols : a.pl[2].src_cols, src_rows = c == 1 ? a.pl[1].src_rows : a.pl[2].src_rows;
    const int tiles_x = c == 1 ? a.pl[1].tiles_x : a.pl[2].tiles_x;
    upsample_tile_body<Pixel, 4, MFMA>(lds, dst, src, cols, col_of, rows, src_cols, src_rows, bit_depth, t % tiles_x, t / tiles_x);
}

// Where an enhancement-layer column / row reads the base layer: centre tap position and phase.
//   variant 0: the general formula of upsample_base_layer_frame (hevcdsp_template.c:2217-2226, 2255-2262, 2317-2325, 2364-2372)
//              and of the *_all block slots (:1835-1953);
//   variant 1 / 2: what the x2 / x1.5 block slots compute instead (:1956-2163) -- fixed phase patterns from the sample's
//              parity / residue that ignore the phase offsets carried by add* (kept: this is what the reference decodes).
static void axis_map(int variant, bool chroma, bool vertical, int v, int start, int scale, int add, ohevc_upsample_tap &t)
{
    const int d = v - start;
    int pos, phase;
    if (variant == 0 || (chroma && vertical)) {                 // chroma rows keep the scaled position in every variant
        const int r16 = ((d * scale + add) >> 12) + (chroma && vertical ? -4 : 0);       // the -4: :1945, :2044, :2147, :2367
        pos = r16 >> 4; phase = r16 & 15;
        static const int x2v[2] = { 14, 6 }, x15v[3] = { 15, 9, 4 };
        if (variant == 1) phase = x2v[v & 1];                   // up_sample_filter_chroma_x2_v[y & 1], :2046 (hevcdsp.c:1020-1024)
        if (variant == 2) phase = x15v[v % 3];                  // up_sample_filter_x1_5chroma[y % 3], :2149 (hevcdsp.c:1007-1012)
    } else if (variant == 1) {                                  // x2: phases 0 / 8 (hevcdsp.c:988-992, 1014-1018)
        if (!chroma) { phase = ((vertical ? d : v) & 1) * 8; pos = d >> 1; }             // :1968-1970 (x & 1), :2018-2019 ((y - top) & 1)
        else         { phase = (v & 1) * 8;                  pos = v >> 1; }             // :1993-1995: x >> 1, not (x - left) >> 1
    } else {                                                    // x1.5: phases 0 / 11 / 5 (hevcdsp.c:994-1005)
        static const int ph[3] = { 0, 11, 5 };
        phase = ph[d % 3]; pos = (d << 1) / 3;                  // :2072-2074, :2097-2099, :2124-2125
    }
    t.pos = (int16_t)pos; t.phase = (uint8_t)phase; t.reserved = 0;
}

}  // namespace ohevc

static int g_upsample_variant = 0;      // 0: tile form, vertical pass on the matrix cores (shipped since round 6), 2: tile form with the dot-product vertical pass (round 4), 1: round-2 strip form
extern "C" int ohevc_debug_set_upsample_variant(int v) { const int prev = g_upsample_variant; g_upsample_variant = v; return prev; }

extern "C" int ohevc_upsample_make_maps(const ohevc_upsample_params *p, int plane, ohevc_upsample_tap *cols, int16_t *col_of,
                                        ohevc_upsample_tap *rows, int *src_cols, int *src_rows)
{
    using namespace ohevc;
    OHEVC_REQUIRE(p != nullptr && cols != nullptr && col_of != nullptr && rows != nullptr && src_cols != nullptr && src_rows != nullptr, "null argument");
    OHEVC_REQUIRE(plane >= 0 && plane < 3, "plane");
    OHEVC_REQUIRE(p->el_width > 0 && p->el_height > 0 && p->bl_width > 0 && p->bl_height > 0 && p->el_width < 32768 && p->el_height < 32768, "picture sizes");
    OHEVC_REQUIRE(p->idx >= 0 && p->idx <= 2, "idx must be 0 (general), 1 (x2) or 2 (x1.5); x1 (SNR) scalability is a plain copy (hevc_filter.c:1187-1190)");
    const bool chroma = plane != 0;
    const int variant = p->block_slots && (p->idx == 1 || p->idx == 2) ? p->idx : 0;
    const int w = chroma ? p->el_width >> 1 : p->el_width, h = chroma ? p->el_height >> 1 : p->el_height;
    const int left = chroma ? p->win_left >> 1 : p->win_left, top = chroma ? p->win_top >> 1 : p->win_top;
    const int right_end = w - (chroma ? p->win_right >> 1 : p->win_right), bottom_end = h - (chroma ? p->win_bottom >> 1 : p->win_bottom);
    OHEVC_REQUIRE(left >= 0 && top >= 0 && right_end > left && bottom_end > top, "scaled reference layer window");
    // clamp of the horizontal position: [left, right_end]; the frame function's chroma pass stops one earlier (:2318)
    const int right_clip = (chroma && !p->block_slots) ? right_end - 1 : right_end;
    const int sx = chroma ? p->scale_x_chroma : p->scale_x_luma, ax = chroma ? p->add_x_chroma : p->add_x_luma;
    const int sy = chroma ? p->scale_y_chroma : p->scale_y_luma, ay = chroma ? p->add_y_chroma : p->add_y_luma;
    for (int i = 0; i < w; i++) {
        axis_map(variant, chroma, false, std::min(std::max(i, left), right_clip), left, sx, ax, cols[i]);
        // the vertical pass walks the intermediate columns with a pointer that only advances inside [left, right_end - 2]
        // (:2270, 2284, 2292 and the block slots alike): output column i reads intermediate column min(i, right_end - 1) - left
        col_of[i] = (int16_t)std::max(0, std::min(i, right_end - 1) - left);
    }
    for (int j = 0; j < h; j++) axis_map(variant, chroma, true, std::min(std::max(j, top), bottom_end - 1), top, sy, ay, rows[j]);
    // base-layer extent the passes clamp to: luma min(BL height, EL height) rows (:2214); chroma max(BL height, EL height / 2) / 2 (:2306-2312)
    *src_cols = chroma ? p->bl_width >> 1 : p->bl_width;
    *src_rows = chroma ? (std::max(p->bl_height, p->el_height >> 1) >> 1) : std::min(p->bl_height, p->el_height);
    return OHEVC_OK;
}

extern "C" int ohevc_dev_upsample_picture(const ohevc_plane dst[3], const ohevc_plane src[3], int bit_depth, const ohevc_upsample_tap *const cols[3],
                                          const int16_t *const col_of[3], const ohevc_upsample_tap *const rows[3], const int src_cols[3], const int src_rows[3],
                                          void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(dst != nullptr && src != nullptr && cols != nullptr && col_of != nullptr && rows != nullptr && src_cols != nullptr && src_rows != nullptr, "null argument");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    if (g_upsample_variant == 1) {                       // the strip form has no three-plane kernel
        for (int pl = 0; pl < 3; pl++) {
            int rc = ohevc_dev_upsample_plane(&dst[pl], &src[pl], bit_depth, pl != 0, cols[pl], col_of[pl], rows[pl], src_cols[pl], src_rows[pl], stream);
            if (rc != OHEVC_OK) return rc;
        }
        return OHEVC_OK;
    }
    UpPictureArgs a;
    int total = 0;
    for (int pl = 0; pl < 3; pl++) {
        OHEVC_REQUIRE(dst[pl].data != nullptr && src[pl].data != nullptr && cols[pl] != nullptr && col_of[pl] != nullptr && rows[pl] != nullptr, "planes / maps");
        OHEVC_REQUIRE(dst[pl].width > 0 && dst[pl].height > 0 && src_cols[pl] > 0 && src_rows[pl] > 0, "sizes");
        UpPlaneArgs &q = a.pl[pl];
        q.dst = dst[pl]; q.src = src[pl]; q.cols = cols[pl]; q.col_of = col_of[pl]; q.rows = rows[pl];
        q.src_cols = std::min(src_cols[pl], src[pl].width); q.src_rows = std::min(src_rows[pl], src[pl].height);      // (ohevc_dev_upsample_plane)
        q.tiles_x = (dst[pl].width + 63) / 64;
        q.tiles = q.tiles_x * ((dst[pl].height + UPT_ROWS - 1) / UPT_ROWS);
        total += q.tiles;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_upsample_variant == 0) {
        if (bit_depth == 8) hipLaunchKernelGGL((upsample_tile3_kernel<uint8_t, true>), dim3(total), dim3(256), 0, st, a, bit_depth);
        else                hipLaunchKernelGGL((upsample_tile3_kernel<uint16_t, true>), dim3(total), dim3(256), 0, st, a, bit_depth);
    } else {
        if (bit_depth == 8) hipLaunchKernelGGL((upsample_tile3_kernel<uint8_t, false>), dim3(total), dim3(256), 0, st, a, bit_depth);
        else                hipLaunchKernelGGL((upsample_tile3_kernel<uint16_t, false>), dim3(total), dim3(256), 0, st, a, bit_depth);
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_dev_upsample_plane(const ohevc_plane *dst, const ohevc_plane *src, int bit_depth, int chroma, const ohevc_upsample_tap *cols,
                                        const int16_t *col_of, const ohevc_upsample_tap *rows, int src_cols, int src_rows, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(dst != nullptr && src != nullptr && dst->data != nullptr && src->data != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(cols != nullptr && col_of != nullptr && rows != nullptr, "maps");
    OHEVC_REQUIRE(dst->width > 0 && dst->height > 0 && src_cols > 0 && src_rows > 0, "sizes");
    // never read below the plane that was handed over (the reference would read its frame padding there, see make_maps)
    src_cols = std::min(src_cols, src->width); src_rows = std::min(src_rows, src->height);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_upsample_variant != 1) {
        const dim3 tgrid((dst->width + 63) / 64, (dst->height + UPT_ROWS - 1) / UPT_ROWS);
#define UP_LAUNCH(PIX, TP, MF) hipLaunchKernelGGL((upsample_tile_kernel<PIX, TP, MF>), tgrid, dim3(256), 0, st, *dst, *src, cols, col_of, rows, src_cols, src_rows, bit_depth)
        if (g_upsample_variant == 0) {
            if (bit_depth == 8) { if (chroma) UP_LAUNCH(uint8_t, 4, true); else UP_LAUNCH(uint8_t, 8, true); }
            else                { if (chroma) UP_LAUNCH(uint16_t, 4, true); else UP_LAUNCH(uint16_t, 8, true); }
        } else {
            if (bit_depth == 8) { if (chroma) UP_LAUNCH(uint8_t, 4, false); else UP_LAUNCH(uint8_t, 8, false); }
            else                { if (chroma) UP_LAUNCH(uint16_t, 4, false); else UP_LAUNCH(uint16_t, 8, false); }
        }
#undef UP_LAUNCH
        OHEVC_HIP_TRY(hipGetLastError());
        return OHEVC_OK;
    }
    constexpr int ROWS = 8;                                     // output rows per thread (sliding window of filtered base-layer rows)
    const dim3 grid((dst->width + 63) / 64, (dst->height + 4 * ROWS - 1) / (4 * ROWS));
    if (bit_depth == 8) {
        if (chroma) hipLaunchKernelGGL((upsample_kernel<uint8_t, 4, ROWS>), grid, dim3(256), 0, st, *dst, *src, cols, col_of, rows, src_cols, src_rows, bit_depth);
        else        hipLaunchKernelGGL((upsample_kernel<uint8_t, 8, ROWS>), grid, dim3(256), 0, st, *dst, *src, cols, col_of, rows, src_cols, src_rows, bit_depth);
    } else {
        if (chroma) hipLaunchKernelGGL((upsample_kernel<uint16_t, 4, ROWS>), grid, dim3(256), 0, st, *dst, *src, cols, col_of, rows, src_cols, src_rows, bit_depth);
        else        hipLaunchKernelGGL((upsample_kernel<uint16_t, 8, ROWS>), grid, dim3(256), 0, st, *dst, *src, cols, col_of, rows, src_cols, src_rows, bit_depth);
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}
