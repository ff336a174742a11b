// mc_kernels.hip -- luma (8-tap) / chroma (4-tap) motion compensation with uni / bi / weighted prediction.
//
// Replaces luma_mc_uni/bi + chroma_mc_uni/bi (hevc.c:1641-1949), the ten put_hevc_{qpel,epel}* table families
// (hevcdsp_template.c:610-1609) and vdsp.emulated_edge_mc (videodsp_template.c:26-100):
//   * one wavefront per prediction block, walking it in 16x16 tiles;
//   * the (w+T-1) x (h+T-1) reference window is gathered into LDS with CLAMPED picture coordinates -- the same
//     samples the reference's edge emulation would have replicated -- so no padded copy is ever made;
//   * horizontal pass -> int16 LDS tile (the reference's tmp_array, :763-776), vertical pass from LDS;
//   * bi-prediction evaluates both references inside the job, so the reference's int16 tmp[64*64] hand-off
//     between put_hevc_qpel and put_hevc_qpel_bi (hevc.c:1761-1764) stays in registers.
// The four (mx,my) cases of the reference collapse into one exact formulation:
//   tmp  = mx ? (sum fh*src) >> (BD-8) : src << (14-BD)            (14-bit intermediate, fits int16)
//   v14  = my ? (sum fv*tmp) >> 6      : tmp                       ((S << (14-BD)) >> 6 == S >> (BD-8) exactly)
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace ohevc {

#ifdef OHEVC_LAB
// ff_hevc_qpel_filters / ff_hevc_epel_filters (libavcodec/hevcdsp.c:1028-1042): the standard's interpolation taps
__device__ const signed char kLumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};
__device__ const signed char kChromaTaps[8][4] = {
    { 0, 64, 0, 0 },
    { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 },
    { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 },
};

constexpr int MC_TILE = 16;
constexpr int MC_WIN  = MC_TILE + 7;        // window rows/cols for the 8-tap case
constexpr int MC_WINP = MC_WIN + 1;         // padded LDS row (int16)
struct McShared {
    short win[MC_WIN][MC_WINP];
    short tmp[MC_WIN][MC_TILE];
};

#endif

// The rounding term of the final shift.  The reference drops it at BIT_DEPTH 14 ("#if BIT_DEPTH < 14 ... #else int offset = 0",
// hevcdsp_template.c:653-657, 677-681, 808-812, 835-839, ...), where the uni shift is 0 and the bi shift 1.
__device__ __forceinline__ int mc_round(int shift, int bit_depth) { return bit_depth < 14 ? 1 << (shift - 1) : 0; }

#ifdef OHEVC_LAB          // round 1's first kernel (ohevc_debug_set_mc_variant(1)): the lab build only
template <typename Pixel>
__device__ __forceinline__ void mc_tile_ref(McShared &sh, const ohevc_plane &ref, int sx, int sy, int mx, int my,
                                            bool luma, int tw, int th, int bit_depth, int lane, int *v)
{
    const int taps = luma ? 8 : 4, before = luma ? 3 : 1;
    const int ww = tw + taps - 1, wh = th + taps - 1;
    const unsigned char *base = static_cast<const unsigned char *>(ref.data);
    const int xmax = ref.width - 1, ymax = ref.height - 1;
    // ---- gather window, coordinates clamped to the picture (== emulated_edge_mc's replication)
    for (int idx = lane; idx < ww * wh; idx += 64) {
        const int wy = idx / ww, wx = idx - wy * ww;
        int x = sx + wx - before, y = sy + wy - before;
        x = x < 0 ? 0 : x > xmax ? xmax : x;
        y = y < 0 ? 0 : y > ymax ? ymax : y;
        const Pixel p = *reinterpret_cast<const Pixel *>(base + (size_t)y * ref.stride + (size_t)x * sizeof(Pixel));
        sh.win[wy][wx] = (short)p;
    }
    __syncthreads();
    // ---- horizontal pass over all window rows
    const signed char *fh = luma ? kLumaTaps[mx] : kChromaTaps[mx];
    for (int idx = lane; idx < tw * wh; idx += 64) {
        const int r = idx / tw, x = idx - r * tw;
        int s;
        if (mx) {
            s = 0;
            for (int k = 0; k < taps; k++) s += fh[k] * (int)(unsigned short)sh.win[r][x + k];
            s >>= bit_depth - 8;
        } else {
            s = (int)(unsigned short)sh.win[r][x + before] << (14 - bit_depth);
        }
        sh.tmp[r][x] = (short)s;
    }
    __syncthreads();
    // ---- vertical pass: up to 4 samples per lane
    const signed char *fv = luma ? kLumaTaps[my] : kChromaTaps[my];
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
        const int idx = lane + 64 * k4;
        int s = 0;
        if (idx < tw * th) {
            const int y = idx / tw, x = idx - y * tw;
            if (my) {
                for (int k = 0; k < taps; k++) s += fv[k] * (int)sh.tmp[y + k][x];
                s >>= 6;
            } else {
                s = sh.tmp[y + before][x];
            }
        }
        v[k4] = s;
    }
    __syncthreads();
}

template <typename Pixel>
__global__ __launch_bounds__(64) void mc_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs,
                                                const ohevc_mc_job *__restrict__ jobs, int njobs, int bit_depth)
{
    __shared__ McShared sh;
    const int lane = threadIdx.x;
    const ohevc_mc_job jb = jobs[blockIdx.x];
    const bool luma = jb.plane == 0, bi = jb.flags & OHEVC_MC_BI, weighted = jb.flags & OHEVC_MC_WEIGHTED;
    const ohevc_plane ref0 = refs[3 * jb.ref0 + jb.plane];
    const ohevc_plane ref1 = refs[3 * (bi ? jb.ref1 : jb.ref0) + jb.plane];
    unsigned char *dbase = PLANE_PTR3(dst, jb.plane);
    const int dstride = PLANE_STRIDE3(dst, jb.plane);
    const int maxv = (1 << bit_depth) - 1;

    for (int ty = 0; ty < jb.h; ty += MC_TILE)
        for (int tx = 0; tx < jb.w; tx += MC_TILE) {
            const int tw = jb.w - tx < MC_TILE ? jb.w - tx : MC_TILE;
            const int th = jb.h - ty < MC_TILE ? jb.h - ty : MC_TILE;
            int v0[4], v1[4];
            mc_tile_ref<Pixel>(sh, ref0, jb.sx0 + tx, jb.sy0 + ty, jb.mx0, jb.my0, luma, tw, th, bit_depth, lane, v0);
            if (bi)
                mc_tile_ref<Pixel>(sh, ref1, jb.sx1 + tx, jb.sy1 + ty, jb.mx1, jb.my1, luma, tw, th, bit_depth, lane, v1);
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                const int idx = lane + 64 * k4;
                if (idx >= tw * th) continue;
                const int y = idx / tw, x = idx - y * tw;
                int out;
                if (!bi && !weighted) {             // put_hevc_*_uni_*: hevcdsp_template.c:626-640,796-943
                    const int shift = 14 - bit_depth;
                    out = (v0[k4] + mc_round(shift, bit_depth)) >> shift;
                } else if (bi && !weighted) {       // put_hevc_*_bi_*: :642-666,822-983
                    const int shift = 15 - bit_depth;
                    out = (v1[k4] + v0[k4] + mc_round(shift, bit_depth)) >> shift;
                } else if (!bi) {                   // put_hevc_*_uni_w_*: :668-690,985-1134
                    const int shift = jb.denom + 14 - bit_depth;
                    out = ((v0[k4] * jb.wx0 + mc_round(shift, bit_depth)) >> shift) + jb.ox0 * (1 << (bit_depth - 8));
                } else {                            // put_hevc_*_bi_w_*: :692-716,1012-1174
                    const int log2wd = jb.denom + 14 - bit_depth;
                    const int o0 = jb.ox0 * (1 << (bit_depth - 8)), o1 = jb.ox1 * (1 << (bit_depth - 8));
                    out = (v1[k4] * jb.wx1 + v0[k4] * jb.wx0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
                }
                out = out < 0 ? 0 : out > maxv ? maxv : out;
                *reinterpret_cast<Pixel *>(dbase + (size_t)(jb.y + ty + y) * dstride + (size_t)(jb.x + tx + x) * sizeof(Pixel)) = (Pixel)out;
            }
        }
}

// ------------------------------------------------------------------ v2: packed-pair dot-product form
#endif  // OHEVC_LAB

// Same semantics as mc_kernel; per 16x16 tile and reference:
//   1. stage the (tw+7) x (th+7) window into LDS as int16, row-major, window column 0 at LDS column 0.  Interior windows
//      use aligned dword loads (4 / 2 samples per lane-load); windows touching the picture edge take the clamped
//      per-sample path.
//   2. horizontal pass: lane = (window row, group of 4 outputs): 12 consecutive int16 = 6 dwords -> 4 outputs, each
//      4 x v_dot2_i32_i16 against packed tap pairs (odd outputs use v_alignbit'ed pairs).  Results go to LDS
//      column-major [x][row].
//   3. vertical pass: the SAME primitive along the other axis: lane = (column, group of 4 rows).
//   Zero phases use the identity filter {0,0,0,64,0,0,0,0}: 64*p == p << 6, (64*t) >> 6 == t  -> all four (mx,my)
//   cases of the reference in one exact path; chroma's 4-tap filters are the 8-tap form with four zero taps.
__device__ const signed char kLumaTaps8[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__device__ const signed char kChromaTaps8[8][8] = {
    { 0, 64, 0, 0, 0, 0, 0, 0 }, { -2, 58, 10, -2, 0, 0, 0, 0 }, { -4, 54, 16, -2, 0, 0, 0, 0 }, { -6, 46, 28, -4, 0, 0, 0, 0 },
    { -4, 36, 36, -4, 0, 0, 0, 0 }, { -4, 28, 46, -6, 0, 0, 0, 0 }, { -2, 16, 54, -4, 0, 0, 0, 0 }, { -2, 10, 58, -2, 0, 0, 0, 0 } };

#ifdef OHEVC_LAB
constexpr int MC2_PITCH = 24;                 // int16 elements per LDS line (23 used + 1 pad; 48 bytes keeps dwordx2 alignment)
struct Mc2Shared {
    short win[MC_WIN][MC2_PITCH];             // [window row][window col]
    short tmp[MC_TILE][MC2_PITCH];            // [output col][window row]  (column-major for the vertical pass)
};
#endif

__device__ __forceinline__ unsigned tap_pair(const signed char *f, int k)
{
    return pack16((int)f[k], (int)f[k + 1]);
}

// out[i] = init + sum_{k<8} f[k] * v[i + k], i = 0..3, v = the 12 int16 held in d[0..5]
__device__ __forceinline__ void filt4(const unsigned *d, unsigned f01, unsigned f23, unsigned f45, unsigned f67, int init, int *out)
{
    const unsigned m0 = __builtin_amdgcn_alignbit(d[1], d[0], 16), m1 = __builtin_amdgcn_alignbit(d[2], d[1], 16);
    const unsigned m2 = __builtin_amdgcn_alignbit(d[3], d[2], 16), m3 = __builtin_amdgcn_alignbit(d[4], d[3], 16);
    const unsigned m4 = __builtin_amdgcn_alignbit(d[5], d[4], 16);
    out[0] = dot2_i16(d[3], f67, dot2_i16(d[2], f45, dot2_i16(d[1], f23, dot2_i16(d[0], f01, init))));
    out[1] = dot2_i16(m3, f67, dot2_i16(m2, f45, dot2_i16(m1, f23, dot2_i16(m0, f01, init))));
    out[2] = dot2_i16(d[4], f67, dot2_i16(d[3], f45, dot2_i16(d[2], f23, dot2_i16(d[1], f01, init))));
    out[3] = dot2_i16(m4, f67, dot2_i16(m3, f45, dot2_i16(m2, f23, dot2_i16(m1, f01, init))));
}

#ifdef OHEVC_LAB          // round 1's second kernel (variant 2); its filt4 / tap_pair primitives above live on in mc3
template <typename Pixel>
__device__ __forceinline__ void mc2_tile_ref(Mc2Shared &sh, const ohevc_plane &ref, int sx, int sy, const signed char *fh,
                                             const signed char *fv, int before, int taps, int tw, int th, int bit_depth,
                                             int lane, int *v)
{
    const int ww = tw + taps - 1, wh = th + taps - 1;
    const unsigned char *base = static_cast<const unsigned char *>(ref.data);
    const int wx0 = sx - before, wy0 = sy - before;               // picture position of window (0,0)
    constexpr int PPD = 4 / (int)sizeof(Pixel);                    // samples per dword
    const int xa = wx0 & ~(PPD - 1);                               // aligned start of the row span
    const int ndw = (wx0 + ww - xa + PPD - 1) / PPD;               // dwords per window row (<= 7 / 12)
    const bool interior = wx0 >= 0 && wy0 >= 0 && wy0 + wh <= ref.height && xa + ndw * PPD <= ref.width &&
                          ((ref.stride | (int)(reinterpret_cast<uintptr_t>(base))) & 3) == 0;
    if (interior) {
        constexpr int DPR = sizeof(Pixel) == 1 ? 8 : 16;           // dword slots per row (power of two >= ndw)
        for (int idx = lane; idx < wh * DPR; idx += 64) {
            const int r = idx / DPR, dw = idx % DPR;
            if (dw >= ndw) continue;
            const unsigned raw = *reinterpret_cast<const unsigned *>(base + (size_t)(wy0 + r) * ref.stride + (size_t)(xa + dw * PPD) * sizeof(Pixel));
            const int c0 = xa + dw * PPD - wx0;                    // window column of the first sample in this dword
#pragma unroll
            for (int j = 0; j < PPD; j++) {
                const int c = c0 + j;
                const int pv = sizeof(Pixel) == 1 ? (int)((raw >> (8 * j)) & 0xff) : (int)((raw >> (16 * j)) & 0xffff);
                if (c >= 0 && c < MC2_PITCH) sh.win[r][c] = (short)pv;
            }
        }
    } else {
        const int xmax = ref.width - 1, ymax = ref.height - 1;
        for (int idx = lane; idx < ww * wh; idx += 64) {
            const int wy = idx / ww, wx = idx - wy * ww;
            int x = wx0 + wx, y = wy0 + wy;
            x = x < 0 ? 0 : x > xmax ? xmax : x;
            y = y < 0 ? 0 : y > ymax ? ymax : y;
            sh.win[wy][wx] = (short)*reinterpret_cast<const Pixel *>(base + (size_t)y * ref.stride + (size_t)x * sizeof(Pixel));
        }
    }
    __syncthreads();
    // ---- horizontal pass: lane = (window row r, output group q of 4 columns)
    {
        const unsigned f01 = tap_pair(fh, 0), f23 = tap_pair(fh, 2), f45 = tap_pair(fh, 4), f67 = tap_pair(fh, 6);
        const int hshift = bit_depth - 8;
        for (int idx = lane; idx < wh * 4; idx += 64) {
            const int r = idx >> 2, q = idx & 3;
            if (q * 4 >= tw) continue;
            const u32x2 *p = reinterpret_cast<const u32x2 *>(&sh.win[r][q * 4]);
            const u32x2 a = p[0], b = p[1], c = p[2];
            const unsigned d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
            int o[4];
            filt4(d, f01, f23, f45, f67, 0, o);
#pragma unroll
            for (int j = 0; j < 4; j++) sh.tmp[q * 4 + j][r] = (short)(o[j] >> hshift);
        }
    }
    __syncthreads();
    // ---- vertical pass: lane = (output column x, group g of 4 rows)
    {
        const unsigned f01 = tap_pair(fv, 0), f23 = tap_pair(fv, 2), f45 = tap_pair(fv, 4), f67 = tap_pair(fv, 6);
        const int x = lane & 15, g = lane >> 4;
        int o[4] = { 0, 0, 0, 0 };
        if (x < tw && g * 4 < th) {
            const u32x2 *p = reinterpret_cast<const u32x2 *>(&sh.tmp[x][g * 4]);
            const u32x2 a = p[0], b = p[1], c = p[2];
            const unsigned d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
            filt4(d, f01, f23, f45, f67, 0, o);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = o[j] >> 6;
    }
    __syncthreads();
}

template <typename Pixel>
__global__ __launch_bounds__(64) void mc2_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs,
                                                 const ohevc_mc_job *__restrict__ jobs, int njobs, int bit_depth)
{
    __shared__ __attribute__((aligned(16))) Mc2Shared sh;
    const int lane = threadIdx.x;
    const ohevc_mc_job jb = jobs[blockIdx.x];
    const bool luma = jb.plane == 0, bi = jb.flags & OHEVC_MC_BI, weighted = jb.flags & OHEVC_MC_WEIGHTED;
    const ohevc_plane ref0 = refs[3 * jb.ref0 + jb.plane];
    const ohevc_plane ref1 = refs[3 * (bi ? jb.ref1 : jb.ref0) + jb.plane];
    unsigned char *dbase = PLANE_PTR3(dst, jb.plane);
    const int dstride = PLANE_STRIDE3(dst, jb.plane);
    const int maxv = (1 << bit_depth) - 1;
    const int before = luma ? 3 : 1, taps = luma ? 8 : 4;
    const signed char *fh0 = luma ? kLumaTaps8[jb.mx0] : kChromaTaps8[jb.mx0], *fv0 = luma ? kLumaTaps8[jb.my0] : kChromaTaps8[jb.my0];
    const signed char *fh1 = luma ? kLumaTaps8[jb.mx1] : kChromaTaps8[jb.mx1], *fv1 = luma ? kLumaTaps8[jb.my1] : kChromaTaps8[jb.my1];
    const int x = lane & 15, g = lane >> 4;

    for (int ty = 0; ty < jb.h; ty += MC_TILE)
        for (int tx = 0; tx < jb.w; tx += MC_TILE) {
            const int tw = jb.w - tx < MC_TILE ? jb.w - tx : MC_TILE;
            const int th = jb.h - ty < MC_TILE ? jb.h - ty : MC_TILE;
            int v0[4], v1[4] = { 0, 0, 0, 0 };
            mc2_tile_ref<Pixel>(sh, ref0, jb.sx0 + tx, jb.sy0 + ty, fh0, fv0, before, taps, tw, th, bit_depth, lane, v0);
            if (bi)
                mc2_tile_ref<Pixel>(sh, ref1, jb.sx1 + tx, jb.sy1 + ty, fh1, fv1, before, taps, tw, th, bit_depth, lane, v1);
            if (x >= tw) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int y = g * 4 + j;
                if (y >= th) continue;
                int out;
                if (!bi && !weighted) {
                    const int shift = 14 - bit_depth;
                    out = (v0[j] + mc_round(shift, bit_depth)) >> shift;
                } else if (bi && !weighted) {
                    const int shift = 15 - bit_depth;
                    out = (v1[j] + v0[j] + mc_round(shift, bit_depth)) >> shift;
                } else if (!bi) {
                    const int shift = jb.denom + 14 - bit_depth;
                    out = ((v0[j] * jb.wx0 + mc_round(shift, bit_depth)) >> shift) + jb.ox0 * (1 << (bit_depth - 8));
                } else {
                    const int log2wd = jb.denom + 14 - bit_depth;
                    const int o0 = jb.ox0 * (1 << (bit_depth - 8)), o1 = jb.ox1 * (1 << (bit_depth - 8));
                    out = (v1[j] * jb.wx1 + v0[j] * jb.wx0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
                }
                out = out < 0 ? 0 : out > maxv ? maxv : out;
                *reinterpret_cast<Pixel *>(dbase + (size_t)(jb.y + ty + y) * dstride + (size_t)(jb.x + tx + x) * sizeof(Pixel)) = (Pixel)out;
            }
        }
}

#endif  // OHEVC_LAB

// ------------------------------------------------------------------ v3: v2's arithmetic, more parallelism
// SLOTS = 1: one job per wavefront in 16x16 tiles.  SLOTS = 4: four jobs of at most 8x8 samples per wavefront (16 lanes
// each) -- chroma blocks of 4:2:0 content and small PUs are mostly that size, and a 64-lane wave is 3/4 idle on them.
// In both forms the windows of BOTH references are requested before the first barrier, so a bi-predicted block exposes
// one global-memory latency instead of two.
template <int SLOTS> struct Mc3Cfg {
    static constexpr int SL = 64 / SLOTS;            // lanes per job slot
    static constexpr int T = SLOTS == 1 ? 16 : 8;    // tile edge
    static constexpr int WIN = T + 7;                // window rows
    static constexpr int PITCH = T + 8;              // int16 per LDS line (window cols + pad; dwordx2 aligned)
    static constexpr int GROUPS = T / 4;             // 4-output groups per line
};
template <int SLOTS> struct Mc3Shared {
    short win[SLOTS][2][Mc3Cfg<SLOTS>::WIN][Mc3Cfg<SLOTS>::PITCH];
    short tmp[SLOTS][Mc3Cfg<SLOTS>::T][Mc3Cfg<SLOTS>::PITCH];
};

// returns the OR of everything staged (16-bit samples: two per dword) -- the caller tests it against the bit depth's range
template <typename Pixel, int SLOTS>
__device__ __forceinline__ unsigned mc3_stage(short (*win)[Mc3Cfg<SLOTS>::PITCH], const ohevc_plane &ref, int wx0, int wy0, int ww, int wh,
                                              int sub, bool active)
{
    using C = Mc3Cfg<SLOTS>;
    unsigned seen = 0;
    const unsigned char *base = static_cast<const unsigned char *>(ref.data);
    constexpr int PPD = 4 / (int)sizeof(Pixel);
    constexpr int DPR = (sizeof(Pixel) == 1 || SLOTS == 4) ? 8 : 16;      // dword slots per window row (>= max ndw)
    const int xa = wx0 & ~(PPD - 1);
    const int ndw = (wx0 + ww - xa + PPD - 1) / PPD;
    const bool interior = wx0 >= 0 && wy0 >= 0 && wy0 + wh <= ref.height && xa + ndw * PPD <= ref.width &&
                          ((ref.stride | (int)(reinterpret_cast<uintptr_t>(base))) & 3) == 0;
    if (!active) return 0;
    if (interior) {
#pragma unroll 1
        for (int idx = sub; idx < wh * DPR; idx += C::SL) {
            const int r = idx / DPR, dw = idx % DPR;
            if (dw >= ndw) continue;
            const unsigned raw = *reinterpret_cast<const unsigned *>(base + (size_t)(wy0 + r) * ref.stride + (size_t)(xa + dw * PPD) * sizeof(Pixel));
            const int c0 = xa + dw * PPD - wx0;
            if (sizeof(Pixel) == 2) seen |= raw;       // an alignment sample just outside the window may be in it: harmless (mc3_redo is exact)
#pragma unroll
            for (int j = 0; j < PPD; j++) {
                const int c = c0 + j;
                const int pv = sizeof(Pixel) == 1 ? (int)((raw >> (8 * j)) & 0xff) : (int)((raw >> (16 * j)) & 0xffff);
                if (c >= 0 && c < C::PITCH) win[r][c] = (short)pv;
            }
        }
    } else {
        const int xmax = ref.width - 1, ymax = ref.height - 1;
#pragma unroll 1
        for (int idx = sub; idx < ww * wh; idx += C::SL) {
            const int wy = idx / ww, wx = idx - wy * ww;
            int x = wx0 + wx, y = wy0 + wy;
            x = x < 0 ? 0 : x > xmax ? xmax : x;
            y = y < 0 ? 0 : y > ymax ? ymax : y;
            const Pixel pv = *reinterpret_cast<const Pixel *>(base + (size_t)y * ref.stride + (size_t)x * sizeof(Pixel));
            win[wy][wx] = (short)pv;
            if (sizeof(Pixel) == 2) seen |= (unsigned)pv;
        }
    }
    return seen;
}

// horizontal + vertical pass of one staged window -> v[0..3] (4 rows of this lane's column), 14-bit intermediate
template <int SLOTS>
__device__ __forceinline__ void mc3_filter(short (*win)[Mc3Cfg<SLOTS>::PITCH], short (*tmp)[Mc3Cfg<SLOTS>::PITCH], const signed char *fh,
                                           const signed char *fv, int tw, int th, int wh, int bit_depth, int sub, bool active, int *v)
{
    using C = Mc3Cfg<SLOTS>;
    {
        const unsigned f01 = tap_pair(fh, 0), f23 = tap_pair(fh, 2), f45 = tap_pair(fh, 4), f67 = tap_pair(fh, 6);
        const int hshift = bit_depth - 8;
#pragma unroll 1
        for (int it = 0; it < (C::WIN * C::GROUPS + C::SL - 1) / C::SL; it++) {
            const int idx = sub + it * C::SL, r = idx / C::GROUPS, q = idx % C::GROUPS;
            if (active && r < wh && q * 4 < tw) {
                const u32x2 *p = reinterpret_cast<const u32x2 *>(&win[r][q * 4]);
                const u32x2 a = p[0], b = p[1], c = p[2];
                const unsigned d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
                int o[4];
                filt4(d, f01, f23, f45, f67, 0, o);
#pragma unroll
                for (int j = 0; j < 4; j++) tmp[q * 4 + j][r] = (short)(o[j] >> hshift);
            }
        }
    }
    __syncthreads();
    {
        const unsigned f01 = tap_pair(fv, 0), f23 = tap_pair(fv, 2), f45 = tap_pair(fv, 4), f67 = tap_pair(fv, 6);
        const int x = sub % C::T, g = sub / C::T;
        int o[4] = { 0, 0, 0, 0 };
        if (active && x < tw && g * 4 < th) {
            const u32x2 *p = reinterpret_cast<const u32x2 *>(&tmp[x][g * 4]);
            const u32x2 a = p[0], b = p[1], c = p[2];
            const unsigned d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
            filt4(d, f01, f23, f45, f67, 0, o);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = o[j] >> 6;
    }
    __syncthreads();
}

// Samples above the bit depth's range (16-bit planes only).  Above 8 bit the reference's constrained intra prediction leaves
// samples of up to 0x8080 in its pictures (its byte-wise memset, hevcpred_template.c:117-141) and later pictures predict from them.
// The int16 LDS tiles of this kernel are exact for every sample that fits the bit depth; for larger ones they would wrap where the
// reference computes in int (only the h-pass of its hv case lands in an int16 array, hevcdsp_template.c:763-776) and its full-sample
// uni case is a memcpy that carries them through unclipped (:626-640).  A tile whose windows hold such a sample is not written here:
// its bit is set in wild_mask[job] (bit = tile index; every job's word is written, so the buffer needs no clearing) and
// mc3_redo_kernel, launched right behind, computes it in the reference's arithmetic.
template <typename Pixel, int SLOTS>
__global__ __launch_bounds__(64) void mc3_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs,
                                                 const ohevc_mc_job *__restrict__ jobs, int njobs, int bit_depth,
                                                 unsigned short *__restrict__ wild_mask)
{
    using C = Mc3Cfg<SLOTS>;
    __shared__ __attribute__((aligned(16))) Mc3Shared<SLOTS> sh;
    const int lane = threadIdx.x;
    const int slot = SLOTS == 1 ? 0 : lane / C::SL, sub = SLOTS == 1 ? lane : lane % C::SL;   // SLOTS == 1: provably wave-uniform job
    const int jidx = blockIdx.x * SLOTS + slot;
    const bool have = jidx < njobs;
    const ohevc_mc_job jb = jobs[have ? jidx : njobs - 1];
    const bool luma = jb.plane == 0, bi = jb.flags & OHEVC_MC_BI, weighted = jb.flags & OHEVC_MC_WEIGHTED;
    const ohevc_plane ref0 = refs[3 * jb.ref0 + jb.plane];
    const ohevc_plane ref1 = refs[3 * (bi ? jb.ref1 : jb.ref0) + jb.plane];
    unsigned char *dbase = PLANE_PTR3(dst, jb.plane);
    const int dstride = PLANE_STRIDE3(dst, jb.plane);
    const int maxv = (1 << bit_depth) - 1;
    const int before = luma ? 3 : 1, taps = luma ? 8 : 4;
    const signed char *fh0 = luma ? kLumaTaps8[jb.mx0] : kChromaTaps8[jb.mx0], *fv0 = luma ? kLumaTaps8[jb.my0] : kChromaTaps8[jb.my0];
    const signed char *fh1 = luma ? kLumaTaps8[jb.mx1] : kChromaTaps8[jb.mx1], *fv1 = luma ? kLumaTaps8[jb.my1] : kChromaTaps8[jb.my1];
    const int x = sub % C::T, g = sub / C::T;
    // all slots iterate the same (maximal) number of tiles so that the barriers stay uniform; SLOTS == 4 jobs fit one tile
    const int ntx = SLOTS == 1 ? (jb.w + C::T - 1) / C::T : 1, nty = SLOTS == 1 ? (jb.h + C::T - 1) / C::T : 1;
    const unsigned wild_bits = 0x10001u * (unsigned)(0xffff & ~maxv);
    unsigned tile_mask = 0;
    for (int tyi = 0; tyi < nty; tyi++)
        for (int txi = 0; txi < ntx; txi++) {
            const int tx = txi * C::T, ty = tyi * C::T;
            const int tw = jb.w - tx < C::T ? jb.w - tx : C::T, th = jb.h - ty < C::T ? jb.h - ty : C::T;
            const int ww = tw + taps - 1, wh = th + taps - 1;
            const bool active = have && tw > 0 && th > 0;
            unsigned seen = mc3_stage<Pixel, SLOTS>(sh.win[slot][0], ref0, jb.sx0 + tx - before, jb.sy0 + ty - before, ww, wh, sub, active);
            seen |= mc3_stage<Pixel, SLOTS>(sh.win[slot][1], ref1, jb.sx1 + tx - before, jb.sy1 + ty - before, ww, wh, sub, active && bi);
            bool wild = false;                    // block-uniform (all job slots of the block share the verdict)
            if (sizeof(Pixel) == 2) {
                wild = __syncthreads_or((seen & wild_bits) != 0) != 0;
                if (wild) { tile_mask |= 1u << (tyi * ntx + txi); continue; }
            } else {
                __syncthreads();
            }
            int v0[4], v1[4] = { 0, 0, 0, 0 };
            mc3_filter<SLOTS>(sh.win[slot][0], sh.tmp[slot], fh0, fv0, tw, th, wh, bit_depth, sub, active, v0);
            mc3_filter<SLOTS>(sh.win[slot][1], sh.tmp[slot], fh1, fv1, tw, th, wh, bit_depth, sub, active && bi, v1);
            if (!active || x >= tw) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int y = g * 4 + j;
                if (y >= th) continue;
                int out;
                if (!bi && !weighted) {
                    const int shift = 14 - bit_depth;
                    out = (v0[j] + mc_round(shift, bit_depth)) >> shift;
                } else if (bi && !weighted) {
                    const int shift = 15 - bit_depth;
                    out = (v1[j] + v0[j] + mc_round(shift, bit_depth)) >> shift;
                } else if (!bi) {
                    const int shift = jb.denom + 14 - bit_depth;
                    out = ((v0[j] * jb.wx0 + mc_round(shift, bit_depth)) >> shift) + jb.ox0 * (1 << (bit_depth - 8));
                } else {
                    const int log2wd = jb.denom + 14 - bit_depth;
                    const int o0 = jb.ox0 * (1 << (bit_depth - 8)), o1 = jb.ox1 * (1 << (bit_depth - 8));
                    out = (v1[j] * jb.wx1 + v0[j] * jb.wx0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
                }
                out = out < 0 ? 0 : out > maxv ? maxv : out;
                *reinterpret_cast<Pixel *>(dbase + (size_t)(jb.y + ty + y) * dstride + (size_t)(jb.x + tx + x) * sizeof(Pixel)) = (Pixel)out;
            }
        }
    if (sizeof(Pixel) == 2 && have && sub == 0) wild_mask[jidx] = (unsigned short)tile_mask;
}

// ------------------------------------------------------------------ the tiles mc3_kernel left out (see there)
// The 14-bit intermediate of ONE sample, in the reference's own arithmetic case by case (put_hevc_{qpel,epel}_{pixels,h,v,hv},
// hevcdsp_template.c:610-624,731-794,1185-1247), straight from global memory with the coordinates clamped like the staging does.
__device__ int mc3_exact(const ohevc_plane &ref, int sx, int sy, const signed char *fh, const signed char *fv, bool frac_x, bool frac_y, int taps,
                         int before, int bit_depth)
{
    const unsigned char *base = static_cast<const unsigned char *>(ref.data);
    const int xmax = ref.width - 1, ymax = ref.height - 1;
    auto px = [&](int x, int y) {
        x = x < 0 ? 0 : x > xmax ? xmax : x;
        y = y < 0 ? 0 : y > ymax ? ymax : y;
        return (int)*reinterpret_cast<const unsigned short *>(base + (size_t)y * ref.stride + (size_t)x * 2);
    };
    if (!frac_x && !frac_y) return px(sx, sy) << (14 - bit_depth);
    if (!frac_y) {
        int s = 0;
        for (int k = 0; k < taps; k++) s += fh[k] * px(sx + k - before, sy);
        return s >> (bit_depth - 8);
    }
    if (!frac_x) {
        int s = 0;
        for (int k = 0; k < taps; k++) s += fv[k] * px(sx, sy + k - before);
        return s >> (bit_depth - 8);
    }
    int acc = 0;
    for (int r = 0; r < taps; r++) {
        int s = 0;
        for (int k = 0; k < taps; k++) s += fh[k] * px(sx + k - before, sy + r - before);
        acc += fv[r] * (int)(short)(s >> (bit_depth - 8));       // the h-pass lands in an int16 tmp[] (:763-776)
    }
    return acc >> 6;
}

// One wavefront per 64 jobs: the lanes read 64 mask words, the wave then works through the (rare) jobs with a bit set, one sample
// per lane and step.  `tile` = the tile edge of the mc3_kernel instantiation that wrote the masks (16, or 8 for the small-job form).
template <typename Mask>
__global__ __launch_bounds__(64) void mc3_redo_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs, const ohevc_mc_job *__restrict__ jobs,
                                                      int njobs, int bit_depth, const Mask *__restrict__ wild_mask, int tile)
{
    const int lane = threadIdx.x;
    const int maxv = (1 << bit_depth) - 1;
    for (int base = blockIdx.x * 64; base < njobs; base += gridDim.x * 64) {
        const unsigned mine = base + lane < njobs ? wild_mask[base + lane] : 0;
        unsigned long long todo = __ballot(mine != 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned mask = (unsigned)__shfl((int)mine, src);
            const ohevc_mc_job jb = jobs[base + src];
            const bool luma = jb.plane == 0, bi = jb.flags & OHEVC_MC_BI, weighted = jb.flags & OHEVC_MC_WEIGHTED;
            const ohevc_plane ref0 = refs[3 * jb.ref0 + jb.plane];
            const ohevc_plane ref1 = refs[3 * (bi ? jb.ref1 : jb.ref0) + jb.plane];
            unsigned char *dbase = PLANE_PTR3(dst, jb.plane);
            const int dstride = PLANE_STRIDE3(dst, jb.plane);
            const int before = luma ? 3 : 1, taps = luma ? 8 : 4;
            const signed char *fh0 = luma ? kLumaTaps8[jb.mx0] : kChromaTaps8[jb.mx0], *fv0 = luma ? kLumaTaps8[jb.my0] : kChromaTaps8[jb.my0];
            const signed char *fh1 = luma ? kLumaTaps8[jb.mx1] : kChromaTaps8[jb.mx1], *fv1 = luma ? kLumaTaps8[jb.my1] : kChromaTaps8[jb.my1];
            const int ntx = (jb.w + tile - 1) / tile;
            for (unsigned bits = mask; bits; bits &= bits - 1) {
                const int t = __ffs((int)bits) - 1, tx = (t % ntx) * tile, ty = (t / ntx) * tile;
                const int tw = jb.w - tx < tile ? jb.w - tx : tile, th = jb.h - ty < tile ? jb.h - ty : tile;
                for (int i = lane; i < tw * th; i += 64) {
                    const int x = tx + i % tw, y = ty + i / tw;
                    unsigned short *out_px = reinterpret_cast<unsigned short *>(dbase + (size_t)(jb.y + y) * dstride + (size_t)(jb.x + x) * 2);
                    int v0 = mc3_exact(ref0, jb.sx0 + x, jb.sy0 + y, fh0, fv0, jb.mx0 != 0, jb.my0 != 0, taps, before, bit_depth), v1 = 0;
                    if (bi) {                     // hevc.c:1761-1773: list 0 goes through the int16 hand-off array, list 1 does not
                        v0 = (int)(short)v0;
                        v1 = mc3_exact(ref1, jb.sx1 + x, jb.sy1 + y, fh1, fv1, jb.mx1 != 0, jb.my1 != 0, taps, before, bit_depth);
                    }
                    int out;
                    if (!bi && !weighted) {
                        if (!jb.mx0 && !jb.my0) { *out_px = (unsigned short)(v0 >> (14 - bit_depth)); continue; }      // the memcpy case: no clip
                        const int shift = 14 - bit_depth;
                        out = (v0 + mc_round(shift, bit_depth)) >> shift;
                    } else if (bi && !weighted) {
                        const int shift = 15 - bit_depth;
                        out = (v1 + v0 + mc_round(shift, bit_depth)) >> shift;
                    } else if (!bi) {
                        const int shift = jb.denom + 14 - bit_depth;
                        out = ((v0 * jb.wx0 + mc_round(shift, bit_depth)) >> shift) + jb.ox0 * (1 << (bit_depth - 8));
                    } else {
                        const int log2wd = jb.denom + 14 - bit_depth;
                        const int o0 = jb.ox0 * (1 << (bit_depth - 8)), o1 = jb.ox1 * (1 << (bit_depth - 8));
                        out = (v1 * jb.wx1 + v0 * jb.wx0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
                    }
                    *out_px = (unsigned short)(out < 0 ? 0 : out > maxv ? maxv : out);
                }
            }
        }
    }
}

// 1 = first (scalar) kernel, 2 = packed-pair kernel, 3 = LDS tiles (mc3), 4 (shipped) = matrix cores (mc4) for tiles and mc3's four-jobs-per-
// wavefront form for the small-block batch (it wins there: profiles/r02zi), 5 = mc4 for both.  (env: A/B of whole-decoder runs)
int g_mc_variant = 6;
#ifdef OHEVC_LAB
int g_mc_twin = 0;          // ohevc_debug_set_mc_variant(102 / 103 / 104): mc4q_kernel's traffic-only / arithmetic-only twin / the kernel itself
#endif

#include "mc4_kernel.hpp"
#include "mc4q_kernel.hpp"

}  // namespace ohevc

// One mask word per job, written by mc3_kernel<uint16_t> and read by mc3_redo_kernel behind it on the same stream: a buffer per
// stream, grown on demand (growing waits for the stream: a launch in flight may still use the old one).
namespace {
struct WildScratch { int device; hipStream_t stream; unsigned short *buf; size_t cap; };
std::mutex g_wild_m;
std::vector<WildScratch> g_wild;
int wild_scratch(hipStream_t st, int njobs, unsigned short **out)
{
    int dev = 0;
    OHEVC_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_wild_m);
    WildScratch *w = nullptr;
    for (auto &e : g_wild) if (e.device == dev && e.stream == st) { w = &e; break; }
    if (!w) { g_wild.push_back({dev, st, nullptr, 0}); w = &g_wild.back(); }
    if ((size_t)njobs > w->cap) {
        if (w->buf) { OHEVC_HIP_TRY(hipStreamSynchronize(st)); OHEVC_HIP_TRY(hipFree(w->buf)); w->buf = nullptr; w->cap = 0; }
        const size_t cap = std::max<size_t>((size_t)njobs * 2, 65536);
        OHEVC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w->buf), cap * sizeof(unsigned)));       // mc4 uses 32-bit words (atomicOr)
        w->cap = cap;
    }
    *out = w->buf;
    return OHEVC_OK;
}
}  // namespace

// a stream is going away (ohevc_ctx_destroy): its mask buffer with it.  The caller has synchronised the stream.
void ohevc_mc_forget_stream(void *stream)
{
    std::lock_guard<std::mutex> g(g_wild_m);
    for (size_t i = 0; i < g_wild.size(); i++)
        if (g_wild[i].stream == static_cast<hipStream_t>(stream)) {
            if (g_wild[i].buf) (void)hipFree(g_wild[i].buf);
            g_wild.erase(g_wild.begin() + (long)i);
            break;
        }
}

static int mc_launch(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                     const ohevc_mc_job *jobs, int njobs, void *stream, bool small, int max_w = 64, int max_h = 64)
{
    using namespace ohevc;
    OHEVC_REQUIRE(dst != nullptr, "dst");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(refs != nullptr && n_ref_slots > 0, "refs");
    OHEVC_REQUIRE(jobs != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0, "jobs must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(dst, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned short *wild = nullptr;
    const bool v4q = small && g_mc_variant == 6;                      // four small blocks per matrix-core tile (mc4q_kernel)
    const bool v4 = !v4q && (g_mc_variant == 5 || ((g_mc_variant == 4 || g_mc_variant == 6) && !small));
#ifdef OHEVC_LAB
    const bool v3 = v4 || v4q || small || (g_mc_variant != 1 && g_mc_variant != 2);
#else
    const bool v3 = true;              // variants 1 and 2 exist in the lab build only: they fall through to mc3 here
#endif
    if (bit_depth > 8 && v3) {
        rc = wild_scratch(st, njobs, &wild);
        if (rc != OHEVC_OK) return rc;
    }
    const int redo_grid = std::min(256, (njobs + 63) / 64);
    if (v4q) {
        const int grid = (((njobs + 3) / 4 + 3) / 4 + 7) & ~7;          // quads of jobs, 4 wavefronts per workgroup, XCD-contiguous ranges
        unsigned *wild32 = reinterpret_cast<unsigned *>(wild);
        if (bit_depth > 8) OHEVC_HIP_TRY(hipMemsetAsync(wild32, 0, (size_t)njobs * sizeof(unsigned), st));
#ifdef OHEVC_LAB
        if (g_mc_twin == 1 && bit_depth == 8)      hipLaunchKernelGGL((mc4q_kernel<uint8_t, 1, 1>), dim3(grid), dim3(256), 0, st, ps, refs, n_ref_slots, jobs, njobs, bit_depth, wild32);
        else if (g_mc_twin == 1)                   hipLaunchKernelGGL((mc4q_kernel<uint16_t, 1, 1>), dim3(grid), dim3(256), 0, st, ps, refs, n_ref_slots, jobs, njobs, bit_depth, wild32);
        else if (g_mc_twin == 2 && bit_depth == 8) hipLaunchKernelGGL((mc4q_kernel<uint8_t, 1, 2>), dim3(grid), dim3(256), 0, st, ps, refs, n_ref_slots, jobs, njobs, bit_depth, wild32);
        else if (g_mc_twin == 2)                   hipLaunchKernelGGL((mc4q_kernel<uint16_t, 1, 2>), dim3(grid), dim3(256), 0, st, ps, refs, n_ref_slots, jobs, njobs, bit_depth, wild32);
        else
#endif
        if (bit_depth == 8) hipLaunchKernelGGL((mc4q_kernel<uint8_t, 1, 0>), dim3(grid), dim3(256), 0, st, ps, refs, n_ref_slots, jobs, njobs, bit_depth, wild32);
        else                hipLaunchKernelGGL((mc4q_kernel<uint16_t, 1, 0>), dim3(grid), dim3(256), 0, st, ps, refs, n_ref_slots, jobs, njobs, bit_depth, wild32);
        if (bit_depth > 8) hipLaunchKernelGGL((mc3_redo_kernel<unsigned>), dim3(redo_grid), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild32, 16);
    } else if (v4) {                          // the matrix-core form: work unit = one 16x16 tile, 4 units per wavefront, 4 wavefronts per workgroup
        const bool multi = max_w > 16 || max_h > 16;
        const int tiles = ((max_w + 15) / 16) * ((max_h + 15) / 16);
        const auto up8 = [](int v) { return (v + 7) & ~7; };            // XCD-contiguous job ranges (mc4_kernel)
        const dim3 grid = multi ? dim3(up8((njobs + 3) / 4), (tiles + MC4_UNITS - 1) / MC4_UNITS) : dim3(up8((njobs + 4 * MC4_UNITS - 1) / (4 * MC4_UNITS)));
        unsigned *wild32 = reinterpret_cast<unsigned *>(wild);
        if (bit_depth == 8) {
            if (multi) hipLaunchKernelGGL((mc4_kernel<uint8_t, true>), grid, dim3(256), 0, st, ps, refs, jobs, njobs, bit_depth, wild32);
            else       hipLaunchKernelGGL((mc4_kernel<uint8_t, false>), grid, dim3(256), 0, st, ps, refs, jobs, njobs, bit_depth, wild32);
        } else {
            OHEVC_HIP_TRY(hipMemsetAsync(wild32, 0, (size_t)njobs * sizeof(unsigned), st));
            if (multi) hipLaunchKernelGGL((mc4_kernel<uint16_t, true>), grid, dim3(256), 0, st, ps, refs, jobs, njobs, bit_depth, wild32);
            else       hipLaunchKernelGGL((mc4_kernel<uint16_t, false>), grid, dim3(256), 0, st, ps, refs, jobs, njobs, bit_depth, wild32);
            hipLaunchKernelGGL((mc3_redo_kernel<unsigned>), dim3(redo_grid), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild32, 16);
        }
    } else if (small) {
        const int grid = (njobs + 3) / 4;
        if (bit_depth == 8) hipLaunchKernelGGL((mc3_kernel<uint8_t, 4>), dim3(grid), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild);
        else {
            hipLaunchKernelGGL((mc3_kernel<uint16_t, 4>), dim3(grid), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild);
            hipLaunchKernelGGL((mc3_redo_kernel<unsigned short>), dim3(redo_grid), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild, 8);
        }
#ifdef OHEVC_LAB
    } else if (g_mc_variant == 1) {
        if (bit_depth == 8) hipLaunchKernelGGL((mc_kernel<uint8_t>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth);
        else                hipLaunchKernelGGL((mc_kernel<uint16_t>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth);
    } else if (g_mc_variant == 2) {
        if (bit_depth == 8) hipLaunchKernelGGL((mc2_kernel<uint8_t>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth);
        else                hipLaunchKernelGGL((mc2_kernel<uint16_t>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth);
#endif
    } else {
        if (bit_depth == 8) hipLaunchKernelGGL((mc3_kernel<uint8_t, 1>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild);
        else {
            hipLaunchKernelGGL((mc3_kernel<uint16_t, 1>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild);
            hipLaunchKernelGGL((mc3_redo_kernel<unsigned short>), dim3(redo_grid), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth, wild, 16);
        }
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_dev_mc_batch(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                                  const ohevc_mc_job *jobs, int njobs, void *stream)
{
    return mc_launch(dst, refs, n_ref_slots, bit_depth, jobs, njobs, stream, false);
}

extern "C" int ohevc_dev_mc_batch_bounded(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                                          const ohevc_mc_job *jobs, int njobs, int max_w, int max_h, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(max_w >= 1 && max_w <= 64 && max_h >= 1 && max_h <= 64, "job size bound");
    return mc_launch(dst, refs, n_ref_slots, bit_depth, jobs, njobs, stream, false, max_w, max_h);
}

extern "C" int ohevc_dev_mc_batch_small(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                                        const ohevc_mc_job *jobs, int njobs, void *stream)
{
    return mc_launch(dst, refs, n_ref_slots, bit_depth, jobs, njobs, stream, true, 8, 8);
}

extern "C" int ohevc_debug_set_mc_variant(int variant)
{
    int old = ohevc::g_mc_variant;
    if (variant >= 1 && variant <= 6) ohevc::g_mc_variant = variant;
#ifdef OHEVC_LAB
    if (variant >= 102 && variant <= 104) ohevc::g_mc_twin = variant == 104 ? 0 : variant - 101;
#endif
    return old;
}
