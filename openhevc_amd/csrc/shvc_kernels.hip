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
