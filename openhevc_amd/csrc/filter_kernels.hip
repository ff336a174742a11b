// filter_kernels.hip -- in-loop filters: deblocking (luma/chroma, both edge directions) and SAO (band/edge).
//
// Deblocking replaces hevc_{h,v}_loop_filter_{luma,chroma}[_c] (hevcdsp_template.c:1629-1787): one job = one
// table call = an 8-sample edge = two 4-line segments; one lane per line, the per-segment decisions (which need
// lines 0 and 3) are exchanged with quad shuffles.  All edges of one direction are independent (an edge reads
// 4 and writes at most 3 samples on each side; edges are 8 apart), so a whole picture's vertical edges are one
// launch and its horizontal edges the next -- the order deblocking_filter_CTB enforces (hevc_filter.c:385-580).
//
// SAO replaces sao_band_filter / sao_edge_filter[0|1] (hevcdsp_template.c:340-567): reads the deblocked copy
// (the reference's sao_frame, hevc_filter.c:269-315), writes the picture; one workgroup per CTB plane block.
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

namespace ohevc {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int iclip(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

// One line of one 8-sample edge: lane `line` (0..7) of an 8-lane group; every exit is taken by whole 4-line segments.
// beta_in / tc_in before bit-depth scaling, tc_in of this lane's segment.
// The samples of one line across an edge, loaded before the filter's parameters are known (deblock_maps_kernel derives those through
// several dependent byte loads; the sample loads have to be in flight meanwhile, not behind them).
template <typename Pixel>
struct DbkLine {
    unsigned char *pix;     // sample q0 of the line
    int xs;                 // step across the edge, bytes
    bool vec;               // the 8 samples are one vector load / store
    int v[8];               // p3 p2 p1 p0 q0 q1 q2 q3, luma loaded sample by sample
    int c[4];               // p1 p0 q0 q1, chroma.  (Own registers per form: forms that share result registers wait for each other's loads.)
    u32x4 raw;              // luma, vec: the packed samples (unpacked by deblock_filter: touching the data here would put the wait for it here)
};

// FORM: which of the three load / store forms a line takes - 0 luma sample by sample, 1 luma with one vector access, 2 chroma; -1 decided per
// lane (deblock_kernel: any mix of jobs in a wavefront).  deblock_maps_kernel knows the form per WAVEFRONT and instantiates its tail once per
// form: inside one instance the forms would share result registers, and every form's loads would wait for the others'.
enum { DBK_FORM_ANY = -1, DBK_FORM_SAMPLES = 0, DBK_FORM_VECTOR = 1, DBK_FORM_CHROMA = 2 };
template <typename Pixel, int FORM = DBK_FORM_ANY>
__device__ __forceinline__ DbkLine<Pixel> deblock_load(unsigned char *plane_base, int stride, int jx, int jy, int jplane, bool vertical, int line, bool wanted = true)
{      // wanted = false: this line's segment is not filtered (chroma: its tc is 0).  The second segment of a chroma edge can lie outside the
       // plane (chroma heights / widths are multiples of 4, edges are 8 long); such a lane reads the matching line of the FIRST segment
       // instead - inside the plane, not used, never stored - rather than taking a branch around its loads.      // (The plane's base and pitch arrive resolved: picked from the PlaneSet behind a reference, the selection becomes a load from a scratch
       // copy of the kernel arguments.)
    DbkLine<Pixel> s;
    s.xs = vertical ? (int)sizeof(Pixel) : stride;
    const int ys = vertical ? stride : (int)sizeof(Pixel);      // step along the edge
    s.pix = plane_base + (size_t)jy * stride + (size_t)jx * sizeof(Pixel) + (size_t)(wanted ? line : (line & 3)) * ys;
    // Vertical luma edges: the 8 samples across the edge are 8 / 16 contiguous bytes of one row - one vector load, one vector store
    // (samples -4 and +3 are written back unchanged: no other edge of the vertical pass writes within 4 samples of this one).
    // Horizontal edges and chroma go sample by sample (along a horizontal edge the lanes of a job are neighbours in memory anyway).
    s.vec = FORM >= 0 ? FORM == DBK_FORM_VECTOR : vertical && jplane == 0 && ((reinterpret_cast<uintptr_t>(s.pix) - 4 * sizeof(Pixel)) & 3) == 0;
#define LD(i) ((int)*reinterpret_cast<const Pixel *>(s.pix + (ptrdiff_t)(i) * s.xs))
#pragma unroll
    for (int k = 0; k < 8; k++) s.v[k] = 0;
    s.raw = u32x4{ 0, 0, 0, 0 };
    s.c[0] = s.c[1] = s.c[2] = s.c[3] = 0;
    if (FORM >= 0 ? FORM == DBK_FORM_CHROMA : jplane != 0) {
        s.c[0] = LD(-2); s.c[1] = LD(-1); s.c[2] = LD(0); s.c[3] = LD(1);
    } else if (s.vec) {
        if (sizeof(Pixel) == 1) {
            const u32x2 raw = *reinterpret_cast<const u32x2 *>(s.pix - 4);
            s.raw.x = raw.x; s.raw.y = raw.y;
        } else {
            s.raw = *reinterpret_cast<const u32x4 *>(s.pix - 8);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) s.v[k] = LD(k - 4);
    }
#undef LD
    return s;
}

template <typename Pixel, int FORM = DBK_FORM_ANY>
__device__ __forceinline__ void deblock_filter(const DbkLine<Pixel> &s, int jplane, int flags, int beta_in, int tc_in, int line, int bit_depth, int l0, int l3)
{      // l0, l3: the lanes of the wavefront that hold lines 0 and 3 of this line's 4-line segment
    const int seg = line >> 2;
    const bool no_p = flags & (seg ? OHEVC_DBK_NO_P1 : OHEVC_DBK_NO_P0);
    const bool no_q = flags & (seg ? OHEVC_DBK_NO_Q1 : OHEVC_DBK_NO_Q0);
    unsigned char *const pix = s.pix;
    const int xs = s.xs, maxv = (1 << bit_depth) - 1;
    const bool vec = FORM >= 0 ? FORM == DBK_FORM_VECTOR : s.vec;
    int vv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) vv[k] = s.v[k];
    if (vec) {
        if (sizeof(Pixel) == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { vv[k] = (s.raw.x >> (8 * k)) & 0xff; vv[4 + k] = (s.raw.y >> (8 * k)) & 0xff; }
        } else {
            vv[0] = s.raw.x & 0xffff; vv[1] = s.raw.x >> 16; vv[2] = s.raw.y & 0xffff; vv[3] = s.raw.y >> 16;
            vv[4] = s.raw.z & 0xffff; vv[5] = s.raw.z >> 16; vv[6] = s.raw.w & 0xffff; vv[7] = s.raw.w >> 16;
        }
    }
    bool dirty = false;
#define ST(i, v) do { if (vec) { vv[(i) + 4] = (v); dirty = true; } else *reinterpret_cast<Pixel *>(pix + (ptrdiff_t)(i) * xs) = (Pixel)(v); } while (0)
    const int tc = tc_in << (bit_depth - 8);
    if (FORM >= 0 ? FORM == DBK_FORM_CHROMA : jplane != 0) {                                // hevc_loop_filter_chroma, :1725-1757
        if (tc <= 0) return;
        const int p1 = s.c[0], p0 = s.c[1], q0 = s.c[2], q1 = s.c[3];
        const int delta = iclip((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
        if (!no_p) ST(-1, iclip(p0 + delta, 0, maxv));
        if (!no_q) ST(0, iclip(q0 - delta, 0, maxv));
        return;
    }
    // hevc_loop_filter_luma, :1629-1723
    const int beta = beta_in << (bit_depth - 8);
    const int p3 = vv[0], p2 = vv[1], p1 = vv[2], p0 = vv[3], q0 = vv[4], q1 = vv[5], q2 = vv[6], q3 = vv[7];
    // (every exit below is taken by whole segments or writes nothing; the vector store sits in flush())
    auto flush = [&]() {
        if (!vec || !dirty) return;
        constexpr unsigned M = sizeof(Pixel) == 1 ? 0xffu : 0xffffu;        // the (Pixel) cast of the sample-wise stores
        if (sizeof(Pixel) == 1)
            *reinterpret_cast<u32x2 *>(pix - 4) = u32x2{ (vv[0] & M) | ((vv[1] & M) << 8) | ((vv[2] & M) << 16) | ((vv[3] & M) << 24),
                                                         (vv[4] & M) | ((vv[5] & M) << 8) | ((vv[6] & M) << 16) | ((vv[7] & M) << 24) };
        else
            *reinterpret_cast<u32x4 *>(pix - 8) = u32x4{ (vv[0] & M) | ((vv[1] & M) << 16), (vv[2] & M) | ((vv[3] & M) << 16),
                                                         (vv[4] & M) | ((vv[5] & M) << 16), (vv[6] & M) | ((vv[7] & M) << 16) };
    };
    const int dp = iabs(p2 - 2 * p1 + p0), dq = iabs(q2 - 2 * q1 + q0);
    const int flat = iabs(p3 - p0) + iabs(q3 - q0), step = iabs(p0 - q0);
    const int dp0 = __shfl(dp, l0), dp3 = __shfl(dp, l3), dq0 = __shfl(dq, l0), dq3 = __shfl(dq, l3);
    const int flat0 = __shfl(flat, l0), flat3 = __shfl(flat, l3), step0 = __shfl(step, l0), step3 = __shfl(step, l3);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int tc25 = (tc * 5 + 1) >> 1;
    const bool strong = flat0 < (beta >> 3) && step0 < tc25 && flat3 < (beta >> 3) && step3 < tc25 &&
                        (d0 << 1) < (beta >> 2) && (d3 << 1) < (beta >> 2);
    if (strong) {
        const int tc2 = tc << 1;
        if (!no_p) {
            ST(-1, p0 + iclip(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc2, tc2));
            ST(-2, p1 + iclip(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2));
            ST(-3, p2 + iclip(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc2, tc2));
        }
        if (!no_q) {
            ST(0, q0 + iclip(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc2, tc2));
            ST(1, q1 + iclip(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2));
            ST(2, q2 + iclip(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc2, tc2));
        }
    } else {
        const int side = (beta + (beta >> 1)) >> 3, tc_2 = tc >> 1;
        const bool two_p = dp0 + dp3 < side, two_q = dq0 + dq3 < side;
        int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (iabs(delta) >= 10 * tc) return;
        delta = iclip(delta, -tc, tc);
        if (!no_p) ST(-1, iclip(p0 + delta, 0, maxv));
        if (!no_q) ST(0, iclip(q0 - delta, 0, maxv));
        if (!no_p && two_p) ST(-2, iclip(p1 + iclip((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2), 0, maxv));
        if (!no_q && two_q) ST(1, iclip(q1 + iclip((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2), 0, maxv));
    }
    flush();
#undef ST
}

template <typename Pixel>
__device__ __forceinline__ void deblock_line(unsigned char *plane_base, int stride, int jx, int jy, int jplane, int flags, int beta_in, int tc_in, int line, int bit_depth, int l0, int l3)
{
    const DbkLine<Pixel> s = deblock_load<Pixel>(plane_base, stride, jx, jy, jplane, (flags & OHEVC_DBK_VERTICAL_EDGE) != 0, line, jplane == 0 || tc_in > 0);
    deblock_filter<Pixel>(s, jplane, flags, beta_in, tc_in, line, bit_depth, l0, l3);
}

template <typename Pixel>
__global__ __launch_bounds__(256) void deblock_kernel(PlaneSet planes, const ohevc_dbk_job *__restrict__ jobs, int njobs, int bit_depth)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int job = tid >> 3, line = tid & 7, seg = line >> 2;
    if (job >= njobs) return;                        // whole 8-lane groups leave together
    const u32x4 jraw = reinterpret_cast<const u32x4 *>(jobs)[job];
    const int jx = jraw.x & 0xffff, jy = jraw.x >> 16, jplane = jraw.y & 0xff, flags = (jraw.y >> 8) & 0xff;
    const int beta_in = (jraw.y >> 16) & 0xff;
    const int tc_in = seg ? (int)jraw.z >> 16 : (int)(short)(jraw.z & 0xffff);
    const int l0 = (threadIdx.x & 63) & ~3;
    deblock_line<Pixel>(PLANE_PTR3(planes, jplane), PLANE_STRIDE3(planes, jplane), jx, jy, jplane, flags, beta_in, tc_in, line, bit_depth, l0, l0 | 3);
}

// ------------------------------------------------------------------ deblocking straight from the decoder's maps
// The parameter derivation of deblocking_filter_CTB (hevc_filter.c:345-581) per edge instead of per CTB loop: which 8-sample edges
// exist (boundary-strength maps), QP average of the two sides (get_qPy, :144-150), beta / tc from H.265 table 8-12 (:50-60, TC_CALC
// :340-343, chroma_tc :62-89), pcm / bypass flags (get_pcm, :325-338).  The reference's loops give an edge next to a CTB boundary
// offsets of a particular neighbour CTB; restated per edge (the loops: hevc_filter.c:385-579):
//   vertical edges            beta_offset, tc_offset of the CTB the edge lies in
//   horizontal luma edge      beta_offset of the CTB holding x, tc_offset of the CTB holding x + 8 (the run of a CTB starts 8 samples
//                             inside its left neighbour; the last CTB column runs to the picture edge)
//   horizontal chroma edge    segment 0: tc_offset of the CTB holding x, segment 1: of the CTB holding x + 8 * h
// One 8-lane group per edge position of the 8x8 luma grid (luma) / the (8h)x(8v) grid (each chroma plane); all lanes of a group derive
// the same parameters (a handful of byte loads that hit the same cache line) and groups without an edge leave at once.
OHEVC_CONST_TABLE unsigned char kDbkTc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
                                               4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
OHEVC_CONST_TABLE unsigned char kDbkQpC[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };

// `wave`: which run of 8 units this wavefront takes (the kernels below number their wavefronts differently)
template <typename Pixel, bool CHROMA_ONLY = false>
__device__ __forceinline__ void deblock_maps_lines(unsigned char *base0, unsigned char *base1, unsigned char *base2, int stride0, int stride1, int stride2,
                                                   const ohevc_dbk_maps &m, int vertical, int bit_depth, int luma_units, int chroma_units,
                                                   int luma_uw, int chroma_uw, int wave)
{      // (the planes arrive as scalars: a PlaneSet behind a reference becomes a scratch copy of the kernel arguments, indexed at run time)
    // lane -> (edge, line) inside a wavefront of 8 edges.  Horizontal edges: edge-major (the 8 lanes of an edge are 8 neighbouring samples of
    // a row).  Vertical edges: LINE-major - the lanes of one line across 8 neighbouring edges then read one contiguous 64- / 128-byte
    // piece of a row; edge-major, every lane of a load would sit in its own row.
    const int lane = threadIdx.x & 63;
    const int line = vertical ? lane >> 3 : lane & 7;
    int unit = wave * 8 + (vertical ? lane & 7 : lane >> 3);
    const int l0 = vertical ? ((line & 4) << 3) | (lane & 7) : lane & ~3, l3 = vertical ? l0 + 24 : l0 | 3;
    // every plane's units start at a multiple of 8 (one wavefront = 8 units): the plane is wave-uniform
    const int luma_pad = (luma_units + 7) & ~7, chroma_pad = (chroma_units + 7) & ~7;
    int plane = 0;
    if (unit >= luma_pad) { unit -= luma_pad; plane = 1; if (unit >= chroma_pad) { unit -= chroma_pad; plane = 2; } }
    plane = __builtin_amdgcn_readfirstlane(plane);
    if (unit >= (plane ? chroma_units : luma_units)) return;
    const int hs = plane && (m.chroma_format_idc == 1 || m.chroma_format_idc == 2), vs = plane && m.chroma_format_idc == 1;
    const int uw = plane ? chroma_uw : luma_uw;
    const int x = (unit % uw) << (3 + hs), y = (unit / uw) << (3 + vs);          // luma coordinates of the edge's first sample
    if (vertical ? x == 0 : y == 0) return;
    const int dx = vertical ? 0 : 4 << hs, dy = vertical ? 4 << vs : 0;           // second segment
    const unsigned char *bsm = vertical ? m.vertical_bs : m.horizontal_bs;
    // (map indices as 32-bit unsigned offsets from a scalar base: one VGPR of address per load)
    const int bs0 = bsm[(unsigned)(x + y * m.bs_width) >> 2], bs1 = bsm[(unsigned)(x + dx + (y + dy) * m.bs_width) >> 2];
    if (plane ? !(bs0 == 2 || bs1 == 2) : !(bs0 || bs1)) return;
    const int seg = line >> 2;
    int bs = seg ? bs1 : bs0;
#ifndef OHEVC_HIPEMU
    asm volatile("" : "+v"(bs));                     // selected HERE: sunk to its first use, the select waits for every load issued in between
#endif
    const int log2_ctb = m.log2_ctb_size, ctb_w = (m.width + (1 << log2_ctb) - 1) >> log2_ctb;
    auto qpy = [&](int xx, int yy) { return (int)m.qp_y_tab[(unsigned)((xx >> m.log2_min_cb_size) + (yy >> m.log2_min_cb_size) * m.min_cb_width)]; };
    auto ctb_param = [&](int xx, int k) {                        // DBParams of the CTB holding (xx, y), xx clamped to the last column
        int cx = xx >> log2_ctb;
        cx = cx < ctb_w - 1 ? cx : ctb_w - 1;
        return (int)m.deblock[(unsigned)((cx + (y >> log2_ctb) * ctb_w) * m.deblock_stride + k)];
    };
    auto clipi = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
    const int px = vertical ? x - 1 : x, py = vertical ? y : y - 1;               // the P side of segment 0
    // Second round of loads, all independent of each other and issued together: the line's samples, the four pcm / bypass flags, the two
    // QPs, the CTB's offsets.  (First version: every one of them behind its own branch and wait - nine dependent round trips per edge.)
    unsigned char *const pbase = plane == 0 ? base0 : plane == 1 ? base1 : base2;
    const int pstride = plane == 0 ? stride0 : plane == 1 ? stride1 : stride2;
    const bool vec = vertical && plane == 0 && ((reinterpret_cast<uintptr_t>(pbase) | (unsigned)pstride) & 3) == 0;      // x is a multiple of 8 samples
    auto tail = [&](auto form_tag) {
    constexpr int FORM = decltype(form_tag)::value;
    const DbkLine<Pixel> smp = deblock_load<Pixel, FORM>(pbase, pstride, x >> hs, y >> vs, plane, vertical != 0, line, plane == 0 || bs == 2);
    // get_pcm of the four sides (2 outside the picture).  Loaded unconditionally - without a map, byte 0 of the QP map, not used - so that
    // these loads, the QP and the offset loads below sit in one basic block and leave together; behind `if (m.is_pcm)` the block waited
    // for its own four before the next four were issued.
    const bool has_pcm = m.is_pcm != nullptr;
    const unsigned char *const pcm_map = has_pcm ? m.is_pcm : reinterpret_cast<const unsigned char *>(m.qp_y_tab);
    int pcm_v[4];
    bool pcm_out[4];
    {
        const int xs4[4] = { px, px + dx, x, x + dx }, ys4[4] = { py, py + dy, y, y + dy };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xp = xs4[k] >> m.log2_min_pu_size, yp = ys4[k] >> m.log2_min_pu_size;
            pcm_out[k] = xs4[k] < 0 || ys4[k] < 0 || xp >= m.min_pu_width || yp >= m.min_pu_height;
            const unsigned idx = (unsigned)(yp * m.min_pu_width + xp);
            pcm_v[k] = (int)pcm_map[(pcm_out[k] || !has_pcm) ? 0u : idx];
        }
    }
    // luma: both segments share the QP of segment 0's sides; chroma: each segment its own (and only where bs == 2; elsewhere the address
    // of segment 0 again - the value is not used)
    const int sx = plane && seg && bs == 2 ? dx : 0, sy = plane && seg && bs == 2 ? dy : 0;
    const int qp_p = qpy(px + sx, py + sy), qp_q = qpy(x + sx, y + sy);
    const int beta_offset = ctb_param(x, 0);
    const int tc_offset = ctb_param(plane == 0 ? (vertical ? x : x + 8) : (vertical || !seg ? x : x + (8 << hs)), 1);
    int flags = vertical ? OHEVC_DBK_VERTICAL_EDGE : 0;
    if (has_pcm)
        flags |= ((pcm_out[0] || pcm_v[0]) ? OHEVC_DBK_NO_P0 : 0) | ((pcm_out[1] || pcm_v[1]) ? OHEVC_DBK_NO_P1 : 0) |
                 ((pcm_out[2] || pcm_v[2]) ? OHEVC_DBK_NO_Q0 : 0) | ((pcm_out[3] || pcm_v[3]) ? OHEVC_DBK_NO_Q1 : 0);
    const int qp_y = (qp_p + qp_q + 1) >> 1;
    int beta = 0, tc = 0;
    if (plane == 0) {
        const int qb = clipi(qp_y + beta_offset, 0, 51);
        beta = qb < 16 ? 0 : qb < 29 ? qb - 10 : 2 * qb - 38;                    // H.265 table 8-12, the beta' column in closed form
        tc = bs ? kDbkTc[clipi(qp_y + 2 * (bs - 1) + (tc_offset >> 1 << 1), 0, 53)] : 0;
    } else if (bs == 2) {
        const int qp_i = clipi(qp_y + (plane == 1 ? m.cb_qp_offset : m.cr_qp_offset), 0, 57);
        const int qp = m.chroma_format_idc == 1 ? (qp_i < 30 ? qp_i : qp_i > 43 ? qp_i - 6 : (int)kDbkQpC[qp_i - 30]) : clipi(qp_i, 0, 51);
        tc = kDbkTc[clipi(qp + 2 + tc_offset, 0, 53)];
    }
    deblock_filter<Pixel, FORM>(smp, plane, flags, beta, tc, line, bit_depth, l0, l3);
    };
    if (CHROMA_ONLY || plane != 0) tail(std::integral_constant<int, DBK_FORM_CHROMA>{});          // wave-uniform: scalar branches
    else if (vec)   tail(std::integral_constant<int, DBK_FORM_VECTOR>{});
    else            tail(std::integral_constant<int, DBK_FORM_SAMPLES>{});
}

template <typename Pixel>
__global__ __launch_bounds__(256) void deblock_maps_kernel(PlaneSet planes, ohevc_dbk_maps m, int vertical, int bit_depth, int luma_units, int chroma_units,
                                                           int luma_uw, int chroma_uw)
{
    deblock_maps_lines<Pixel>(planes.data[0], planes.data[1], planes.data[2], planes.stride[0], planes.stride[1], planes.stride[2], m, vertical, bit_depth,
                              luma_units, chroma_units, luma_uw, chroma_uw, (int)((blockIdx.x * 256 + threadIdx.x) >> 6));
}

}  // namespace ohevc
#include "dbk4_kernel.hpp"
namespace ohevc {

// The shipped form up to 10 bit: workgroups [0, luma_groups) take the luma segments, one per lane (dbk4_kernel.hpp); the rest run the chroma
// edges line by line (deblock_maps_lines with no luma units).
template <typename Pixel>
__global__ __launch_bounds__(256) void deblock_maps_segments_kernel(PlaneSet planes, ohevc_dbk_maps m, int vertical, int bit_depth, int luma_groups, int luma_segments,
                                                                    int segments_per_row, int chroma_units, int chroma_uw)
{
    __shared__ unsigned char tc_tab[64];
    if (threadIdx.x < 54) tc_tab[threadIdx.x] = kDbkTc[threadIdx.x];
    __syncthreads();
    if ((int)blockIdx.x >= luma_groups) {
        deblock_maps_lines<Pixel, true>(planes.data[0], planes.data[1], planes.data[2], planes.stride[0], planes.stride[1], planes.stride[2], m, vertical, bit_depth,
                                        0, chroma_units, 0, chroma_uw, (int)(((blockIdx.x - luma_groups) * 256 + threadIdx.x) >> 6));
        return;
    }
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= luma_segments) return;
    const int ux = u % segments_per_row, uy = u / segments_per_row;
    if (vertical) { if (ux) deblock_maps_luma4<Pixel, true>(planes.data[0], planes.stride[0], m, bit_depth, ux << 3, uy << 2, tc_tab); }
    else          { if (uy) deblock_maps_luma4<Pixel, false>(planes.data[0], planes.stride[0], m, bit_depth, ux << 2, uy << 3, tc_tab); }
}

template <typename Pixel>
__device__ __forceinline__ bool sao_wide_ok(const ohevc_sao_job &jb, const unsigned char *sbase, const unsigned char *dbase, int sstride, int dstride, int pw, int bit_depth);

// One workgroup per SAO block.  Fast form (block width a multiple of 4 samples, dword-aligned rows - every block the decoder makes):
// the (w + 2) x (h + 2) source window goes to LDS with dword loads (edge columns and the lagged samples patched in), every lane then
// produces 4 neighbouring samples and stores them with one 4- / 8-byte access.  Anything else takes the sample-at-a-time form.
// SPLIT (ohevc_debug_set_sao_variant(1); A/B pending): the edge classes run the block's interior - samples no border / restore rule can
// touch - through a short form, and the outer ring (rows 0, h-2, h-1; the first and last quad of every row) through the full one,
// enumerated so that all but one wavefront take a single form.
// Band SAO of a sample above the bit depth's range: the reference's sao_band_filter indexes its 32-entry offset table with src >> (BIT_DEPTH - 5)
// (hevcdsp_template.c:340-365) - past the table for such a sample, i.e. it adds whatever lies on its stack.  Constrained intra prediction
// above 8 bit produces them (0x8080, hevcpred_template.c:159-161).  The kernels wrap the band index (a defined result) and COUNT the event:
// a stream that never triggers it is decoded bit-identically with the reference, one that does has no reference output to compare with
// (ohevc_debug_sao_band_above_range; the stream fuzzer asks before it compares instead of switching SAO off for such streams).
__device__ unsigned g_sao_band_above_range;
__global__ void sao_band_above_range_fetch(unsigned *out, int reset)
{
    *out = g_sao_band_above_range;
    if (reset) g_sao_band_above_range = 0u;
}

template <typename Pixel, bool SPLIT>
__global__ __launch_bounds__(256) void sao_kernel(PlaneSet dst, PlaneSet src, PlaneSet lag, const ohevc_sao_job *__restrict__ jobs, int njobs, int bit_depth,
                                                  ohevc_sao_bypass bp, int g_variant)
{
    constexpr int PXB = (int)sizeof(Pixel), PPD = 4 / PXB, OFF = 4, PITCH = 64 + 2 * OFF;     // window: x = -1 sits at column OFF - 1
    __shared__ __attribute__((aligned(16))) Pixel win[66][PITCH];
    const int ji = (g_variant & 4) ? blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));      // see sao_wide_kernel
    if (ji >= njobs) return;
    const ohevc_sao_job jb = jobs[ji];
    const int w = jb.w, h = jb.h, eo = jb.klass, maxv = (1 << bit_depth) - 1;
    const int ov0 = jb.offset_val[0], ov1 = jb.offset_val[1], ov2 = jb.offset_val[2], ov3 = jb.offset_val[3], ov4 = jb.offset_val[4];
    const int sstride = PLANE_STRIDE3(src, jb.plane), dstride = PLANE_STRIDE3(dst, jb.plane);
    const unsigned char *sbase = PLANE_PTR3(src, jb.plane) + (size_t)jb.y * sstride + (size_t)jb.x * sizeof(Pixel);
    unsigned char *dbase = PLANE_PTR3(dst, jb.plane) + (size_t)jb.y * dstride + (size_t)jb.x * sizeof(Pixel);
    // neighbours outside the plane are never allowed to matter (picture-border samples get offset_val[0], :419-455; the
    // reference reads its frame padding there): clamp the coordinate instead of reading out of bounds
    const int pw = PLANE_WIDTH3(src, jb.plane), ph = PLANE_HEIGHT3(src, jb.plane);
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; };
    const int bx = jb.x, by = jb.y;
    // restore_tqb_pixels (hevc_filter.c:163-193): samples of min-PU blocks flagged in the reference's is_pcm map (PCM CUs with
    // pcm_loop_filter_disabled, cu_transquant_bypass CUs) keep their deblocked value.  The reference walks the PUs of
    // [x0, x0 + width) x [y0, y0 + height) with x0/y0 in LUMA samples but width/height in samples of THIS plane, so for
    // subsampled chroma only the PUs of the CTB's first half are restored; reproduced here (xlim / ylim).
    const unsigned char *const bmap = bp.map;
    const int b_hs = jb.plane ? bp.chroma_hshift : 0, b_vs = jb.plane ? bp.chroma_vshift : 0, b_l2 = bp.log2_min_pu_size;
    const int b_xlim = bp.exact_reference ? ((bx << b_hs) + w) >> b_l2 : 0x7fffffff;
    const int b_ylim = bp.exact_reference ? ((by << b_vs) + h) >> b_l2 : 0x7fffffff;
    // ... and copies (min_pu_size >> hshift) BYTES per row: half a PU row of 16-bit samples (:176,:184)
    const int b_len = (bp.exact_reference && sizeof(Pixel) == 2) ? ((1 << b_l2) >> b_hs) >> 1 : 0x7fffffff;
    auto bypassed = [&](int x, int y) {
        const int xpu = ((bx + x) << b_hs) >> b_l2, ypu = ((by + y) << b_vs) >> b_l2;
        return xpu < b_xlim && ypu < b_ylim && (bx + x) - ((xpu << b_l2) >> b_hs) < b_len && bmap[(size_t)ypu * bp.stride + xpu] != 0;
    };
    const bool fast = (w & 3) == 0 && w <= 64 && h <= 64 &&
                      ((reinterpret_cast<uintptr_t>(sbase) | reinterpret_cast<uintptr_t>(dbase) | (unsigned)sstride | (unsigned)dstride) & 3) == 0;
#define SRC(px_, py_) ((int)*reinterpret_cast<const Pixel *>(sbase + (ptrdiff_t)(clampi(by + (py_), ph - 1) - by) * sstride + \
                                                              (ptrdiff_t)(clampi(bx + (px_), pw - 1) - bx) * (int)sizeof(Pixel)))
    const int quads = w >> 2;                         // fast form: 4 samples per lane and step
    auto store4 = [&](int x0, int y, const int *v) {
        if (sizeof(Pixel) == 1)
            *reinterpret_cast<unsigned *>(dbase + (size_t)y * dstride + (size_t)x0) = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
        else
            *reinterpret_cast<u32x2 *>(dbase + (size_t)y * dstride + (size_t)x0 * 2) = u32x2{ (unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16) };
    };
    if (!(g_variant & 2) && sao_wide_ok<Pixel>(jb, sbase, dbase, sstride, dstride, pw, bit_depth)) return;       // sao_wide_kernel's job
    if (jb.type == OHEVC_SAO_BAND) {                 // sao_band_filter_0, :340-365
        const int shift = bit_depth - 5;
        auto band = [&](int c, int x, int y) {
            if (sizeof(Pixel) == 2 && c > maxv) atomicAdd(&g_sao_band_above_range, 1u);
            const int k = ((c >> shift) - jb.klass) & 31;
            const int off = k == 0 ? ov1 : k == 1 ? ov2 : k == 2 ? ov3 : k == 3 ? ov4 : 0;
            int v = iclip(c + off, 0, maxv);
            if (bmap && bypassed(x, y)) v = c;
            return v;
        };
        if (fast) {
            for (int idx = threadIdx.x; idx < quads * h; idx += 256) {
                const int y = idx / quads, x0 = (idx - y * quads) * 4;
                int c[4], v[4];
                if (sizeof(Pixel) == 1) {
                    const unsigned raw = *reinterpret_cast<const unsigned *>(sbase + (size_t)y * sstride + (size_t)x0);
                    c[0] = raw & 0xff; c[1] = (raw >> 8) & 0xff; c[2] = (raw >> 16) & 0xff; c[3] = raw >> 24;
                } else {
                    const u32x2 raw = *reinterpret_cast<const u32x2 *>(sbase + (size_t)y * sstride + (size_t)x0 * 2);
                    c[0] = raw.x & 0xffff; c[1] = raw.x >> 16; c[2] = raw.y & 0xffff; c[3] = raw.y >> 16;
                }
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = band(c[e], x0 + e, y);
                store4(x0, y, v);
            }
            return;
        }
        for (int idx = threadIdx.x; idx < w * h; idx += 256) {
            const int y = idx / w, x = idx - y * w;
            *reinterpret_cast<Pixel *>(dbase + (size_t)y * dstride + (size_t)x * sizeof(Pixel)) = (Pixel)band(SRC(x, y), x, y);
        }
        return;
    }
    // sao_edge_filter_{0,1}, :372-567
    const int dxa = eo == 1 ? 0 : eo == 3 ? 1 : -1, dya = eo == 0 ? 0 : -1;       // first neighbour; second is its mirror
    const bool b0 = jb.borders & 1, b1 = jb.borders & 2, b2 = jb.borders & 4, b3 = jb.borders & 8;
    const int init_x = (eo != 1 && b0) ? 1 : 0;
    const int w2 = w - ((eo != 1 && b2) ? 1 : 0), h2 = h - ((eo != 0 && b3) ? 1 : 0);
    const bool ve0 = jb.edges & 1, ve1 = jb.edges & 2, he0 = jb.edges & 4, he1 = jb.edges & 8;
    const bool de0 = jb.edges & 16, de1 = jb.edges & 32, de2 = jb.edges & 64, de3 = jb.edges & 128;
    const int sul = !de0 && eo == 2 && !b0 && !b1, sur = !de1 && eo == 3 && !b1 && !b2;
    const int slr = !de2 && eo == 2 && !b2 && !b3, sll = !de3 && eo == 3 && !b0 && !b3;
    const bool lag_below = (jb.quirks & OHEVC_SAO_LAG_BELOW) != 0, lag_above = (jb.quirks & OHEVC_SAO_LAG_ABOVE) != 0;
    const bool lag_mid = (jb.quirks & OHEVC_SAO_LAG_MID) != 0;
    const bool lag_any = (lag_below || lag_above || lag_mid) && eo != 1 && bx + w < pw;
    const unsigned char *lbase = PLANE_PTR3(lag, jb.plane) + (size_t)(bx + w) * sizeof(Pixel);
    const int lstride = PLANE_STRIDE3(lag, jb.plane);
    // samples of the column right of the block that the reference copies too early (ohevc_hip.h): rows ny of the block
    auto stale = [&](int ny) {
        return by + ny >= 0 && by + ny < ph && ((lag_below && (ny == h - 1 || ny == h)) || (lag_above && (ny == -1 || ny == 0)) ||
                                             (lag_mid && (ny == 7 || ny == 8)));
    };
    // one output sample from its three inputs: everything the reference's border / restore rules say about position (x, y)
    auto edge_px = [&](int c, int a, int b, int x, int y) {
        const int s_ = (c > a) - (c < a) + (c > b) - (c < b);                      // -2..2
        int off = s_ == -2 ? ov1 : s_ == -1 ? ov2 : s_ == 0 ? ov0 : s_ == 1 ? ov3 : ov4;  // offset_val[edge_idx[2 + s]], edge_idx = {1,2,0,3,4}
        const bool on_border = (eo != 1 && ((b0 && x == 0) || (b2 && x == w - 1))) ||
                               (eo != 0 && x >= init_x && x < w2 && ((b1 && y == 0) || (b3 && y == h - 1)));
        if (on_border) off = ov0;
        int v = iclip(c + off, 0, maxv);
        if (jb.restore) {
            const bool r = (ve0 && eo != 1 && x == 0 && y >= sul && y < h2 - sll) ||
                           (ve1 && eo != 1 && x == w2 - 1 && y >= sur && y < h2 - slr) ||
                           (he0 && eo != 0 && y == 0 && x >= init_x + sul && x < w2 - sur) ||
                           (he1 && eo != 0 && y == h2 - 1 && x >= init_x + sll && x < w2 - slr) ||
                           (de0 && eo == 2 && x == 0 && y == 0) || (de1 && eo == 3 && x == w2 - 1 && y == 0) ||
                           (de2 && eo == 2 && x == w2 - 1 && y == h2 - 1) || (de3 && eo == 3 && x == 0 && y == h2 - 1);
            if (r) v = c;
        }
        if (bmap && bypassed(x, y)) v = c;
        return v;
    };
    if (fast) {
        // ---- window rows -1 .. h (clamped to the plane), columns 0 .. w - 1 with dword loads ...
        const int dpr = w / PPD;                          // dwords per row
        for (int idx = threadIdx.x; idx < (h + 2) * dpr; idx += 256) {
            const int r = idx / dpr, d = idx - r * dpr;
            const int yy = clampi(by + r - 1, ph - 1) - by;
            const unsigned raw = *reinterpret_cast<const unsigned *>(sbase + (ptrdiff_t)yy * sstride + (size_t)d * 4);
            *reinterpret_cast<unsigned *>(&win[r][OFF + d * PPD]) = raw;
        }
        // ... and the columns left and right of the block sample by sample (clamped; the lagged ones from the early copy)
        for (int idx = threadIdx.x; idx < (h + 2) * 2; idx += 256) {
            const int r = idx >> 1, right = idx & 1, ny = r - 1;
            const int x = right ? w : -1;
            int v = SRC(x, ny);
            if (right && lag_any && stale(ny)) v = (int)*reinterpret_cast<const Pixel *>(lbase + (ptrdiff_t)(by + ny) * lstride);
            win[r][OFF + x] = (Pixel)v;
        }
        __syncthreads();
        if (SPLIT && quads >= 3 && h >= 4) {
            // Which samples can a position rule touch?  x in {0, w-1, w2-1} with w2 in {w-1, w}; y in {0, h-1, h2-1} with h2 in {h-1, h}
            // (on_border and every term of the restore test name one of them): none inside x in [4, w-5], y in [1, h-3].
            const int iq = quads - 2, ih = h - 3, n_int = iq * ih, n_all = n_int + 3 * quads + 2 * ih;
            for (int idx = threadIdx.x; idx < n_all; idx += 256) {
                int v[4], x0, y;
                if (idx < n_int) {
                    const int r = idx / iq;
                    y = 1 + r; x0 = 4 * (1 + idx - r * iq);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int x = x0 + e;
                        const int c = (int)win[y + 1][OFF + x], a = (int)win[y + 1 + dya][OFF + x + dxa], b = (int)win[y + 1 - dya][OFF + x - dxa];
                        const int s_ = (c > a) - (c < a) + (c > b) - (c < b);
                        const int off = s_ == -2 ? ov1 : s_ == -1 ? ov2 : s_ == 0 ? ov0 : s_ == 1 ? ov3 : ov4;
                        v[e] = iclip(c + off, 0, maxv);
                        if (bmap && bypassed(x, y)) v[e] = c;
                    }
                } else {
                    const int k = idx - n_int;
                    if (k < 3 * quads) {
                        const int r = k / quads;
                        y = r == 0 ? 0 : h - 3 + r; x0 = 4 * (k - r * quads);
                    } else {
                        const int k2 = k - 3 * quads;
                        y = 1 + (k2 >> 1); x0 = (k2 & 1) ? w - 4 : 0;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int x = x0 + e;
                        v[e] = edge_px((int)win[y + 1][OFF + x], (int)win[y + 1 + dya][OFF + x + dxa], (int)win[y + 1 - dya][OFF + x - dxa], x, y);
                    }
                }
                store4(x0, y, v);
            }
            return;
        }
        for (int idx = threadIdx.x; idx < quads * h; idx += 256) {
            const int y = idx / quads, x0 = (idx - y * quads) * 4;
            int v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int x = x0 + e;
                v[e] = edge_px((int)win[y + 1][OFF + x], (int)win[y + 1 + dya][OFF + x + dxa], (int)win[y + 1 - dya][OFF + x - dxa], x, y);
            }
            store4(x0, y, v);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < w * h; idx += 256) {
        const int y = idx / w, x = idx - y * w;
        const int c = SRC(x, y);
        int a = SRC(x + dxa, y + dya), b = SRC(x - dxa, y - dya);
        if (lag_any && x == w - 1) {
            if (dxa == 1 && stale(y + dya)) a = (int)*reinterpret_cast<const Pixel *>(lbase + (ptrdiff_t)(by + y + dya) * lstride);
            if (dxa == -1 && stale(y - dya)) b = (int)*reinterpret_cast<const Pixel *>(lbase + (ptrdiff_t)(by + y - dya) * lstride);
        }
        *reinterpret_cast<Pixel *>(dbase + (size_t)y * dstride + (size_t)x * sizeof(Pixel)) = (Pixel)edge_px(c, a, b, x, y);
    }
#undef SRC
}

// ------------------------------------------------------------------ SAO, wide form
// 16 bytes of one row per lane, no LDS, no barrier.  Lane l of the workgroup takes piece l % pieces of row l / pieces (+ passes of 128 / pieces
// rows).  The class offsets come out of a 5-entry byte table through v_perm_b32; what the reference's position rules change (picture-border
// samples take offset_val[0], restored samples and bypassed PUs keep the deblocked value: hevcdsp_template.c:419-566, hevc_filter.c:163-193)
// is decided per lane as byte masks - only in lanes that can hold such a sample - and merged bitwise.  Takes the blocks sao_wide_ok accepts
// (16-byte row pieces, a power-of-two number of them, offsets that fit a byte, no filter-lag patch); sao_kernel takes the others.
template <typename Pixel>
__device__ __forceinline__ bool sao_wide_ok(const ohevc_sao_job &jb, const unsigned char *sbase, const unsigned char *dbase, int sstride, int dstride, int pw, int bit_depth)
{
    constexpr int PPL = 16 / (int)sizeof(Pixel);
    const int w = jb.w, h = jb.h, np = w / PPL;
    const bool band = jb.type == OHEVC_SAO_BAND;
    const bool lagq = (jb.quirks & (OHEVC_SAO_LAG_BELOW | OHEVC_SAO_LAG_ABOVE | OHEVC_SAO_LAG_MID)) != 0 && !band && jb.klass != 1 && jb.x + w < pw;
    bool fits = true;
    for (int k = 0; k < 5; k++) fits = fits && jb.offset_val[k] >= -128 && jb.offset_val[k] < 128;
    return (w % PPL) == 0 && np >= 1 && np <= 8 && (np & (np - 1)) == 0 && h <= 64 && !lagq && fits &&
           ((reinterpret_cast<uintptr_t>(sbase) | reinterpret_cast<uintptr_t>(dbase) | (unsigned)sstride | (unsigned)dstride) & 15) == 0;
}

// sign of d as -1 / 0 / 1 (the compiler prefers two compares and two selects)
__device__ __forceinline__ int sao_sign(int d)
{
#ifdef OHEVC_HIPEMU
    return d < -1 ? -1 : d > 1 ? 1 : d;
#else
    int r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(d));
    return r;
#endif
}

// the same for two 16-bit lanes (left to the compiler, the two clamps become four compares and four selects)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 sao_sign2(s16x2 d)
{
#ifdef OHEVC_HIPEMU
    return s16x2{ (short)(d.x < -1 ? -1 : d.x > 1 ? 1 : d.x), (short)(d.y < -1 ? -1 : d.y > 1 ? 1 : d.y) };
#else
    unsigned r;
    asm("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, %3" : "=&v"(r) : "v"(__builtin_bit_cast(unsigned, d)), "s"(0x00010001u), "s"(0xffffffffu));
    return __builtin_bit_cast(s16x2, r);
#endif
}

// The position rules of one 16-byte row piece as byte masks: bord = samples that take offset_val[0] (picture borders), keep = samples that
// keep the deblocked value (restored slice / tile edges, bypassed PUs).  Out of line: its two dozen wave-uniform flags would otherwise
// crowd the scalar registers of the path every lane runs (measured: 850 SGPR spill moves in the kernel).
struct SaoMasks { u32x4 keep, bord; };
template <typename Pixel>
__device__ __forceinline__ SaoMasks sao_rule_masks(u32x4 j0, u32x4 j1, ohevc_sao_bypass bp, int x0, int y)
{
    constexpr int PPL = 16 / (int)sizeof(Pixel), SB = 8 * (int)sizeof(Pixel), SPD = 4 / (int)sizeof(Pixel);
    constexpr unsigned M = sizeof(Pixel) == 1 ? 0xffu : 0xffffu;
#ifndef OHEVC_HIPEMU
    // The job record and the bypass description are wave-uniform; passed through vector registers here, the two dozen flags derived from
    // them live in vector registers too.  As scalars they crowded the registers of the path every lane runs (850 spill moves when this
    // was part of the row loop), and as an out-of-line call the function gave the kernel a stack - scratch memory per wave at launch.
    asm volatile("" : "+v"(j0.x), "+v"(j0.y), "+v"(j0.z), "+v"(j0.w), "+v"(j1.x), "+v"(j1.y), "+v"(j1.z), "+v"(j1.w));
    asm volatile("" : "+v"(bp.stride), "+v"(bp.log2_min_pu_size), "+v"(bp.chroma_hshift), "+v"(bp.chroma_vshift), "+v"(bp.exact_reference));
#endif
    ohevc_sao_job jb;
    const unsigned words[8] = { j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w };
    __builtin_memcpy(&jb, words, sizeof(jb));
    const int w = jb.w, h = jb.h, eo = jb.klass, bx = jb.x, by = jb.y;
    const bool rules = jb.type != OHEVC_SAO_BAND && (jb.borders != 0 || jb.restore != 0);
    const bool b0 = jb.borders & 1, b1 = jb.borders & 2, b2 = jb.borders & 4, b3 = jb.borders & 8;
    const int init_x = (eo != 1 && b0) ? 1 : 0;
    const int w2 = w - ((eo != 1 && b2) ? 1 : 0), h2 = h - ((eo != 0 && b3) ? 1 : 0);
    const bool ve0 = jb.edges & 1, ve1 = jb.edges & 2, he0 = jb.edges & 4, he1 = jb.edges & 8;
    const bool de0 = jb.edges & 16, de1 = jb.edges & 32, de2 = jb.edges & 64, de3 = jb.edges & 128;
    const int sul = !de0 && eo == 2 && !b0 && !b1, sur = !de1 && eo == 3 && !b1 && !b2;
    const int slr = !de2 && eo == 2 && !b2 && !b3, sll = !de3 && eo == 3 && !b0 && !b3;
    const unsigned char *const bmap = bp.map;        // restore_tqb_pixels: see sao_kernel
    const int b_hs = jb.plane ? bp.chroma_hshift : 0, b_vs = jb.plane ? bp.chroma_vshift : 0, b_l2 = bp.log2_min_pu_size;
    const int b_xlim = bp.exact_reference ? ((bx << b_hs) + w) >> b_l2 : 0x7fffffff;
    const int b_ylim = bp.exact_reference ? ((by << b_vs) + h) >> b_l2 : 0x7fffffff;
    const int b_len = (bp.exact_reference && sizeof(Pixel) == 2) ? ((1 << b_l2) >> b_hs) >> 1 : 0x7fffffff;
    unsigned keep[4] = { 0, 0, 0, 0 }, bord[4] = { 0, 0, 0, 0 };
    for (int e = 0; e < PPL; e++) {
        const int x = x0 + e, d = e / SPD, k = e % SPD;
        const bool on_border = rules && ((eo != 1 && ((b0 && x == 0) || (b2 && x == w - 1))) ||
                                         (eo != 0 && x >= init_x && x < w2 && ((b1 && y == 0) || (b3 && y == h - 1))));
        bool kp = rules && jb.restore &&
                  ((ve0 && eo != 1 && x == 0 && y >= sul && y < h2 - sll) || (ve1 && eo != 1 && x == w2 - 1 && y >= sur && y < h2 - slr) ||
                   (he0 && eo != 0 && y == 0 && x >= init_x + sul && x < w2 - sur) || (he1 && eo != 0 && y == h2 - 1 && x >= init_x + sll && x < w2 - slr) ||
                   (de0 && eo == 2 && x == 0 && y == 0) || (de1 && eo == 3 && x == w2 - 1 && y == 0) ||
                   (de2 && eo == 2 && x == w2 - 1 && y == h2 - 1) || (de3 && eo == 3 && x == 0 && y == h2 - 1));
        if (bmap) {
            const int xpu = ((bx + x) << b_hs) >> b_l2, ypu = ((by + y) << b_vs) >> b_l2;
            kp = kp || (xpu < b_xlim && ypu < b_ylim && (bx + x) - ((xpu << b_l2) >> b_hs) < b_len && bmap[(size_t)ypu * bp.stride + xpu] != 0);
        }
        const unsigned bit = M << (SB * k);
        keep[0] |= d == 0 && kp ? bit : 0; keep[1] |= d == 1 && kp ? bit : 0; keep[2] |= d == 2 && kp ? bit : 0; keep[3] |= d == 3 && kp ? bit : 0;
        bord[0] |= d == 0 && on_border ? bit : 0; bord[1] |= d == 1 && on_border ? bit : 0; bord[2] |= d == 2 && on_border ? bit : 0; bord[3] |= d == 3 && on_border ? bit : 0;
    }
    return SaoMasks{ u32x4{ keep[0], keep[1], keep[2], keep[3] }, u32x4{ bord[0], bord[1], bord[2], bord[3] } };
}

// An edge-class block that no position rule can touch - not at a picture border (so every neighbour of every sample exists), no restored
// slice / tile edge, no bypass map.  Interior CTBs are most of a picture.  What the short form changes against sao_wide_kernel's general loop
// (profiles/r5u_*; fractions of the HBM peak in tools/bench_kernels.py's measure, 8 / 10 bit):
// * ONE wavefront (two at 16 bit) takes the block, kSaoPlainRows row pieces per lane with every load in flight before the first use, no
//   rule flags, no clamped addresses: ~150 vector and ~330 scalar instructions per wavefront instead of 4 x (330 + 300).  By itself: nothing
//   (0.288 -> 0.288) - the kernel is not bound by instruction issue.
// * every load ALIGNED: the classes with a horizontal component (0, 2, 3) read their neighbours one sample to the left / right, and a
//   16-byte load at an address that is not a multiple of 16 costs the memory pipeline more than twice an aligned one - the vertical class
//   ran at 0.344 / 0.508, the other three at 0.272 / 0.389.  Here a lane loads the aligned 16-byte piece of the neighbour row, takes the
//   dword before / after it from the neighbour lane (from memory at the block's own edges) and shifts (v_alignbyte_b32); class 0 needs
//   no second row at all.
constexpr int kSaoPlainRows = 4;
// the value the previous / next lane holds (within a row of 16 lanes; v_mov_b32_dpp row_shr:1 / row_shl:1)
__device__ __forceinline__ unsigned from_lane_before(unsigned v)
{
#ifdef OHEVC_HIPEMU
    return __shfl_up(v, 1);
#else
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
#endif
}
__device__ __forceinline__ unsigned from_lane_after(unsigned v)
{
#ifdef OHEVC_HIPEMU
    return __shfl_down(v, 1);
#else
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, false);
#endif
}
template <typename Pixel>
__device__ __forceinline__ void sao_edge_plain(const ohevc_sao_job &jb, const unsigned char *sbase, unsigned char *dbase, int sstride, int dstride, int maxv)
{
    constexpr int PPL = 16 / (int)sizeof(Pixel), U = kSaoPlainRows, LANES = 64 * (int)sizeof(Pixel), P = (int)sizeof(Pixel);
    const int w = jb.w, h = jb.h, eo = jb.klass;
    const int pieces = w / PPL, lp = pieces == 1 ? 0 : pieces == 2 ? 1 : pieces == 4 ? 2 : 3, rows_per_pass = LANES >> lp;
    const int piece = threadIdx.x & (pieces - 1), row0 = threadIdx.x >> lp;
    const int trips = h > rows_per_pass ? h / rows_per_pass : 1;     // (h is a power of two: the caller's predicate)
    // The two neighbours of a sample are mirror images and enter the class symmetrically: call L the one at x - 1 (x for the vertical
    // class) and R the one at x + 1.  Row of L relative to the sample's: class 0 the same, 1 and 2 the row above, 3 the row below.
    const int off_l = eo == 0 ? 0 : eo == 3 ? sstride : -sstride;
    // class -> 128 + offset through v_perm_b32; class = 2 + sign(c - a) + sign(c - b), which indexes offset_val as {1, 2, 0, 3, 4}
    // (edge_idx, hevcdsp_template.c:372-378)
    const unsigned tab_lo = ((unsigned)(jb.offset_val[1] + 128) & 0xff) | (((unsigned)(jb.offset_val[2] + 128) & 0xff) << 8) |
                            (((unsigned)(jb.offset_val[0] + 128) & 0xff) << 16) | (((unsigned)(jb.offset_val[3] + 128) & 0xff) << 24);
    const unsigned tab_hi = (unsigned)(jb.offset_val[4] + 128) & 0xff;
    const u16x2 maxv2 = { (unsigned short)maxv, (unsigned short)maxv };
    auto edge2 = [&](unsigned c, unsigned a, unsigned b) {          // two samples in 16-bit lanes
        const u16x2 c1 = __builtin_bit_cast(u16x2, c) + u16x2{ 1, 1 };
        // min(max(c + 1 - n, 0), 2) = 0 / 1 / 2 for c < n / c == n / c > n: a saturating subtraction and a minimum
        const unsigned ka = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_sub_sat(c1, __builtin_bit_cast(u16x2, a)), u16x2{ 2, 2 }));
        const unsigned kb = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_sub_sat(c1, __builtin_bit_cast(u16x2, b)), u16x2{ 2, 2 }));
        const unsigned offb = __builtin_amdgcn_perm(tab_hi, tab_lo, ka + kb + 0x0c000c00u);      // (no carry between the lanes: one 32-bit add3)
        u16x2 t = __builtin_bit_cast(u16x2, c) + __builtin_bit_cast(u16x2, offb);
        t = __builtin_elementwise_sub_sat(t, u16x2{ 128, 128 });
        return __builtin_bit_cast(unsigned, __builtin_elementwise_min(t, maxv2));
    };
    auto lo2 = [](unsigned v) { return __builtin_amdgcn_perm(0u, v, 0x0c010c00u); };
    auto hi2 = [](unsigned v) { return __builtin_amdgcn_perm(0u, v, 0x0c030c02u); };
    auto pack4 = [](unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); };
    if (row0 >= h) return;
    auto pass = [&](auto horizontal_tag, auto second_row_tag, auto one_piece_tag) {
        // a horizontal component; neighbours in other rows; the block is one piece wide
        constexpr bool HOR = decltype(horizontal_tag)::value, ROWS = decltype(second_row_tag)::value, ONE = decltype(one_piece_tag)::value;
        for (int t0 = 0; t0 < trips; t0 += U) {
            unsigned cv[U][4], lv[U][4], rv[U][4], le[U], re[U];
#pragma unroll
            for (int u = 0; u < U; u++) {                            // every load of the pass before the first use; no branch among them - with
                {                                                    // one the compiler's wait counts are the minimum over the paths, i.e. "all"
                    const int t = t0 + u < trips ? t0 + u : trips - 1;                     // (a pass a small block does not have: reloaded, never stored)
                    const unsigned char *pc = sbase + ((unsigned)(row0 + t * rows_per_pass) * (unsigned)sstride + (unsigned)piece * 16u);
                    __builtin_memcpy(cv[u], pc, 16);
                    if (ROWS) { __builtin_memcpy(lv[u], pc + off_l, 16); __builtin_memcpy(rv[u], pc - off_l, 16); }
                    // the dword before / after the piece: the neighbour lane holds it, except at the block's own edges
                    // (ONE load for both: the first piece's lane asks for the dword before, the last piece's for the one after; a block of a
                    //  single piece needs both)
                    //  single piece needs both.  Every lane loads - inside a branch the compiler waits for the load before it issues the next row's)
                    if (HOR) __builtin_memcpy(&le[u], piece == 0 ? pc + off_l - 4 : pc - off_l + 16, 4);
                    if (HOR && ONE) __builtin_memcpy(&re[u], pc - off_l + 16, 4);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (t0 + u < trips) {
                    unsigned o[4];
                    const unsigned *lr = ROWS ? lv[u] : cv[u], *rr = ROWS ? rv[u] : cv[u];
                    unsigned before = 0, after = 0;
                    if (HOR) {
                        before = from_lane_before(lr[3]); after = from_lane_after(rr[0]);
                        before = piece == 0 ? le[u] : before; after = piece == pieces - 1 ? (ONE ? re[u] : le[u]) : after;
                    }
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const unsigned a = HOR ? align_bytes(lr[d], d ? lr[d - 1] : before, 4 - P) : lr[d];       // the sample at x - 1 of every sample of dword d
                        const unsigned b = HOR ? align_bytes(d < 3 ? rr[d + 1] : after, rr[d], P) : rr[d];       // ... at x + 1
                        o[d] = sizeof(Pixel) == 2 ? edge2(cv[u][d], a, b) : pack4(edge2(lo2(cv[u][d]), lo2(a), lo2(b)), edge2(hi2(cv[u][d]), hi2(a), hi2(b)));
                    }
                    __builtin_memcpy(dbase + ((unsigned)(row0 + (t0 + u) * rows_per_pass) * (unsigned)dstride + (unsigned)piece * 16u), o, 16);
                }
        }
    };
    if (eo == 1)          pass(std::false_type{}, std::true_type{}, std::false_type{});
    else if (pieces == 1) { if (eo == 0) pass(std::true_type{}, std::false_type{}, std::true_type{}); else pass(std::true_type{}, std::true_type{}, std::true_type{}); }
    else if (eo == 0)     pass(std::true_type{}, std::false_type{}, std::false_type{});
    else                  pass(std::true_type{}, std::true_type{}, std::false_type{});
}

constexpr int kSaoWideThreads = 256, kSaoWideRows = 2;       // lanes per block of the list (band filter: half of them); rows a lane of the band filter has in flight at a time
#ifdef OHEVC_HIPEMU
#define SAO_WIDE_OCCUPANCY
#else
#define SAO_WIDE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(7)))      // <= 72 vector registers: sao_edge_plain's four rows in flight must not cost the general loop a wavefront
#endif
template <typename Pixel>
__global__ __launch_bounds__(kSaoWideThreads) SAO_WIDE_OCCUPANCY void sao_wide_kernel(PlaneSet dst, PlaneSet src, const ohevc_sao_job *__restrict__ jobs, int njobs, int bit_depth, ohevc_sao_bypass bp, int xcd_spread)
{
    constexpr int PPL = 16 / (int)sizeof(Pixel), SB = 8 * (int)sizeof(Pixel), SPD = 4 / (int)sizeof(Pixel);     // samples per lane / dword
    constexpr unsigned M = sizeof(Pixel) == 1 ? 0xffu : 0xffffu;
    typedef const OHEVC_CONST_AS u32x4 *cptr;
    // Workgroups go to the 8 XCDs round-robin by number and every XCD has an L2 of its own: in list order, the two 64-byte halves of a
    // 128-byte line of an 8-bit picture (two neighbouring CTBs) and the rows a block shares with the CTBs above and below would be fetched
    // by several XCDs (FETCH_SIZE: 4.0x the algorithmic bytes for the 8-bit edge classes, 2.0x for the band filter, profiles/r03l_pmc_sao_*).
    // Renumbered, an XCD takes runs of 16 consecutive list entries (1.02x; 5-10 % faster on the edge classes, profiles/r03m_*); one contiguous
    // eighth of the list per XCD reads as little but runs slower on the band filter.  (gridDim.x is a multiple of 128 / 8.)
    const int vb = blockIdx.x >> 3;
    const int ji = (xcd_spread & 15) == 1 ? (int)blockIdx.x : (xcd_spread & 15) == 2 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3)) + vb
                                                      : (vb >> 4) * 128 + (int)(blockIdx.x & 7) * 16 + (vb & 15);       // runs of 16 list entries per XCD
    if (ji >= njobs) return;
    const u32x4 j0 = ((cptr)(jobs + ji))[0], j1 = ((cptr)(jobs + ji))[1];
    ohevc_sao_job jb;
    {
        const unsigned words[8] = { j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w };
        static_assert(sizeof(ohevc_sao_job) == 32, "SAO job record");
        __builtin_memcpy(&jb, words, sizeof(jb));
    }
    const int w = jb.w, h = jb.h, eo = jb.klass, maxv = (1 << bit_depth) - 1;
    const int pw = PLANE_WIDTH3(src, jb.plane), ph = PLANE_HEIGHT3(src, jb.plane);
    const bool is_band = jb.type == OHEVC_SAO_BAND;
    // sao_edge_plain's blocks (ohevc_debug_set_sao_variant(16): without that form); all but its wavefronts leave before any other work
    // (the diagonal classes of 8-bit samples stay with the general loop: the short form is no faster there - the halo of a 64-byte row piece -
    //  and on some boxes 10 % slower)
    const bool plain = !is_band && jb.borders == 0 && jb.restore == 0 && bp.map == nullptr && !(xcd_spread & 16) && (h & (h - 1)) == 0 &&
                       jb.x > 0 && jb.y > 0 && jb.x + w < pw && jb.y + h < ph && (sizeof(Pixel) == 2 || eo < 2);
    if (plain && (int)threadIdx.x >= 64 * (int)sizeof(Pixel)) return;
    const int sstride = PLANE_STRIDE3(src, jb.plane), dstride = PLANE_STRIDE3(dst, jb.plane);
    const unsigned char *splane = PLANE_PTR3(src, jb.plane);
    const unsigned char *sbase = splane + (size_t)jb.y * sstride + (size_t)jb.x * sizeof(Pixel);
    unsigned char *dbase = PLANE_PTR3(dst, jb.plane) + (size_t)jb.y * dstride + (size_t)jb.x * sizeof(Pixel);
    if (!sao_wide_ok<Pixel>(jb, sbase, dbase, sstride, dstride, pw, bit_depth)) return;
    if (plain) {
        sao_edge_plain<Pixel>(jb, sbase, dbase, sstride, dstride, maxv);
        return;
    }
    // The edge classes take all 256 lanes of the workgroup - a 64x64 block of 8-bit samples is then ONE row per lane, every load of the block
    // in flight at once (r4z2: +7-8 % at both depths over 128 lanes; the kernel waits for memory more than it computes since the arithmetic
    // went to packed 16-bit, profiles/r4y_sq_counters_sao_wide.txt).  The band filter keeps 128 lanes with two rows each in flight (-12 % at
    // 8 bit with 256, r4z3): its other two wavefronts leave.
    const int nthreads = is_band ? 128 : kSaoWideThreads;
    if ((int)threadIdx.x >= nthreads) return;
    const int pieces = w / PPL, lp = pieces == 1 ? 0 : pieces == 2 ? 1 : pieces == 4 ? 2 : 3, rows_per_pass = nthreads >> lp;      // a power of two (sao_wide_ok)
    const int piece = threadIdx.x & (pieces - 1), row0 = threadIdx.x >> lp, x0 = piece * PPL;
    const int dxa = eo == 1 ? 0 : eo == 3 ? 1 : -1, dya = eo == 0 ? 0 : -1;       // first neighbour; the second is its mirror
    // class -> offset table for v_perm_b32: byte k of (tab_lo, tab_hi) = 128 + offset of class k.  Edge: k = sign + sign + 2 ->
    // offset_val[{1,2,0,3,4}] (edge_idx, hevcdsp_template.c:372-378); band: k = band index 0..3 -> offset_val[k + 1], 4 = outside the four bands.
    const int ov0 = jb.offset_val[0];
    const int t0 = jb.offset_val[1], t1 = jb.offset_val[2], t2 = is_band ? jb.offset_val[3] : ov0, t3 = is_band ? jb.offset_val[4] : jb.offset_val[3], t4 = is_band ? 0 : jb.offset_val[4];
    const unsigned tab_lo = ((unsigned)(t0 + 128) & 0xff) | (((unsigned)(t1 + 128) & 0xff) << 8) | (((unsigned)(t2 + 128) & 0xff) << 16) | (((unsigned)(t3 + 128) & 0xff) << 24);
    const unsigned tab_hi = (unsigned)(t4 + 128) & 0xff;
    const int shift = bit_depth - 5, band_pos = jb.klass;
    const bool rules = !is_band && (jb.borders != 0 || jb.restore != 0);
    // The arithmetic runs on pairs of 16-bit lanes (v_pk_*_i16 / _u16; 8-bit samples are widened with v_perm_b32 first): the kernel's VALU
    // work, not its memory traffic, was what kept the SIMDs busy (profiles/r02s4_sq_counters_sao_wide.txt: VALU active 70 % of the launch).
    const u16x2 maxv2 = { (unsigned short)maxv, (unsigned short)maxv };
    auto finish2 = [&](unsigned c, s16x2 cls) {       // two samples: class -> 128 + offset (one v_perm_b32 for both), add, clip
        const unsigned offb = __builtin_amdgcn_perm(tab_hi, tab_lo, __builtin_bit_cast(unsigned, cls) | 0x0c000c00u);
        u16x2 t = __builtin_bit_cast(u16x2, c) + __builtin_bit_cast(u16x2, offb);
        t = __builtin_elementwise_sub_sat(t, u16x2{ 128, 128 });
        return __builtin_bit_cast(unsigned, __builtin_elementwise_min(t, maxv2));
    };
    auto edge2 = [&](unsigned c, unsigned a, unsigned b) {
        const s16x2 cc = __builtin_bit_cast(s16x2, c);
        const s16x2 s1 = sao_sign2(cc - __builtin_bit_cast(s16x2, a)), s2 = sao_sign2(cc - __builtin_bit_cast(s16x2, b));
        return finish2(c, s1 + s2 + s16x2{ 2, 2 });
    };
    auto band2 = [&](unsigned c) {
        if (sizeof(Pixel) == 2 && __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, c), maxv2)) != __builtin_bit_cast(unsigned, maxv2))
            atomicAdd(&g_sao_band_above_range, 1u);
        const u16x2 k = ((__builtin_bit_cast(u16x2, c) >> (unsigned short)shift) - u16x2{ (unsigned short)band_pos, (unsigned short)band_pos }) & u16x2{ 31, 31 };
        return finish2(c, __builtin_bit_cast(s16x2, __builtin_elementwise_min(k, u16x2{ 4, 4 })));
    };
    // 8-bit samples: bytes 0, 1 / 2, 3 of a dword as two 16-bit lanes, and back
    auto lo2 = [](unsigned v) { return __builtin_amdgcn_perm(0u, v, 0x0c010c00u); };
    auto hi2 = [](unsigned v) { return __builtin_amdgcn_perm(0u, v, 0x0c030c02u); };
    auto pack4 = [](unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); };
    // A block lives as long as one lane's chain job record -> samples -> store, and the device holds a fixed number of lanes: what a lane has
    // in flight decides the rate (profiles/r03l_*: 4x less traffic changed nothing).  So a lane takes kSaoWideRows rows at a time, all their
    // loads issued before the first is used (the band filter; the edge classes: see `nthreads` above).
    // (one copy of the loop per filter type: with `is_band` tested inside, the branch around the neighbour loads made the compiler wait for
    //  the first row's samples before it issued the second row's loads)
    auto rows = [&](auto band_tag) {
    constexpr bool is_band = decltype(band_tag)::value;
    // (the edge classes are bound by their arithmetic - VALU busy 70 % of the launch, profiles/r02s4_* - and ran 8 % slower with two rows of
    //  registers per lane: they keep one row; the band filter gained 15 % at 8 bit, profiles/r03n_*)
    constexpr int U = is_band ? kSaoWideRows : 1;
    for (int yb = row0; yb < h; yb += U * rows_per_pass) {
        unsigned cv[U][4], av[U][4], bv[U][4];
        int ea[U], eb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int y = yb + u * rows_per_pass < h ? yb + u * rows_per_pass : yb;      // (a row that does not exist: reloaded, never stored)
            __builtin_memcpy(cv[u], sbase + ((unsigned)y * (unsigned)sstride + (unsigned)x0 * (unsigned)sizeof(Pixel)), 16);
            ea[u] = eb[u] = 0;
            if (!is_band) {
                // the two neighbours of every sample: 16 bytes one sample to the left / right of the piece, in the row above / below, clamped to
                // the plane (what lies beyond only reaches samples that take offset_val[0]).  Both loads are unconditional and issued together
                // with the piece's own; a piece that touches the picture's left / right edge loaded itself and is shifted by one sample afterwards
                // (with the loads inside the branches of that case the compiler serialised them: three memory round trips per row instead of one)
                auto fetch = [&](int dx, int dy, unsigned *o) {
                    int yy = jb.y + y + dy;
                    yy = yy < 0 ? 0 : yy > ph - 1 ? ph - 1 : yy;
                    const int xs = jb.x + x0 + dx, xc = xs < 0 ? 0 : xs + PPL > pw ? pw - PPL : xs;
                    __builtin_memcpy(o, splane + ((unsigned)yy * (unsigned)sstride + (unsigned)xc * (unsigned)sizeof(Pixel)), 16);
                    return xs - xc;                          // -1 / +1 at the left / right picture edge
                };
                ea[u] = fetch(dxa, dya, av[u]); eb[u] = fetch(-dxa, -dya, bv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int y = yb + u * rows_per_pass;
            if (y >= h) break;
            unsigned ov_[4];
            if (is_band) {
#pragma unroll
                for (int d = 0; d < 4; d++)
                    ov_[d] = sizeof(Pixel) == 2 ? band2(cv[u][d]) : pack4(band2(lo2(cv[u][d])), band2(hi2(cv[u][d])));
            } else {
                auto shift = [&](unsigned *o, int e) {       // the outermost sample repeated
                    if (e < 0) { o[3] = (o[3] << SB) | (o[2] >> (32 - SB)); o[2] = (o[2] << SB) | (o[1] >> (32 - SB)); o[1] = (o[1] << SB) | (o[0] >> (32 - SB)); o[0] = (o[0] << SB) | (o[0] & M); }
                    if (e > 0) { o[0] = (o[0] >> SB) | (o[1] << (32 - SB)); o[1] = (o[1] >> SB) | (o[2] << (32 - SB)); o[2] = (o[2] >> SB) | (o[3] << (32 - SB)); o[3] = (o[3] >> SB) | (o[3] & ~(0xffffffffu >> SB)); }
                };
                if ((ea[u] | eb[u]) != 0) { shift(av[u], ea[u]); shift(bv[u], eb[u]); }
#pragma unroll
                for (int d = 0; d < 4; d++)
                    ov_[d] = sizeof(Pixel) == 2 ? edge2(cv[u][d], av[u][d], bv[u][d])
                                                : pack4(edge2(lo2(cv[u][d]), lo2(av[u][d]), lo2(bv[u][d])), edge2(hi2(cv[u][d]), hi2(av[u][d]), hi2(bv[u][d])));
            }
            // position rules: only where a lane can hold such a sample (first / last piece of a row, rows 0, h - 2, h - 1), or with a bypass map
            if ((rules && (y == 0 || y >= h - 2 || piece == 0 || piece == pieces - 1)) || bp.map != nullptr) {
                const SaoMasks m = sao_rule_masks<Pixel>(j0, j1, bp, x0, y);
                const unsigned keep[4] = { m.keep.x, m.keep.y, m.keep.z, m.keep.w }, bord[4] = { m.bord.x, m.bord.y, m.bord.z, m.bord.w };
                if ((bord[0] | bord[1] | bord[2] | bord[3]) != 0) {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        unsigned acc = 0;
#pragma unroll
                        for (int k = 0; k < SPD; k++) {
                            int v = (int)((cv[u][d] >> (SB * k)) & M) + ov0;
                            v = v < 0 ? 0 : v > maxv ? maxv : v;
                            acc |= (unsigned)v << (SB * k);
                        }
                        ov_[d] = (ov_[d] & ~bord[d]) | (acc & bord[d]);
                    }
                }
#pragma unroll
                for (int d = 0; d < 4; d++) ov_[d] = (ov_[d] & ~keep[d]) | (cv[u][d] & keep[d]);
            }
            __builtin_memcpy(dbase + ((unsigned)y * (unsigned)dstride + (unsigned)x0 * (unsigned)sizeof(Pixel)), ov_, 16);
        }
    }
    };
    if (is_band) rows(std::true_type{}); else rows(std::false_type{});
}

int g_sao_variant = 0;     // ohevc_debug_set_sao_variant
}  // namespace ohevc

// ------------------------------------------------------------------ boundary strengths (hevc_filter.c:584-700, 805-941)
namespace ohevc {

struct BsField { int mv0, mv1, poc0, poc1; unsigned pf; };       // mv: x in the low, y in the high 16 bits

__device__ __forceinline__ BsField bs_load(const ohevc_bs_maps &m, int x_pu, int y_pu)
{
    const unsigned char *e = m.mvf + ((size_t)y_pu * m.min_pu_width + x_pu) * (size_t)m.mvf_stride;
    BsField f;
    f.mv0 = *reinterpret_cast<const int *>(e + m.off_mv); f.mv1 = *reinterpret_cast<const int *>(e + m.off_mv + 4);
    f.poc0 = *reinterpret_cast<const int *>(e + m.off_poc); f.poc1 = *reinterpret_cast<const int *>(e + m.off_poc + 4);
    f.pf = m.pred_flag_bytes == 4 ? *reinterpret_cast<const unsigned *>(e + m.off_pred_flag) : (unsigned)e[m.off_pred_flag];
    return f;
}
// |a.x - b.x| >= 4 || |a.y - b.y| >= 4 on packed int16 pairs
__device__ __forceinline__ bool bs_far(int a, int b)
{
    const int dx = (int)(short)(a & 0xffff) - (int)(short)(b & 0xffff), dy = (a >> 16) - (b >> 16);
    return (dx < 0 ? -dx : dx) >= 4 || (dy < 0 ? -dy : dy) >= 4;
}
// boundary_strength(), :584-700 (its memcmp shortcut returns what the rules below return for identical entries)
__device__ __forceinline__ int bs_motion(const BsField &c, const BsField &n)
{
    if (c.pf == 3u && n.pf == 3u) {
        if (c.poc0 == n.poc0 && c.poc0 == c.poc1 && n.poc0 == n.poc1)
            return ((bs_far(n.mv0, c.mv0) || bs_far(n.mv1, c.mv1)) && (bs_far(n.mv1, c.mv0) || bs_far(n.mv0, c.mv1))) ? 1 : 0;
        if (n.poc0 == c.poc0 && n.poc1 == c.poc1) return (bs_far(n.mv0, c.mv0) || bs_far(n.mv1, c.mv1)) ? 1 : 0;
        if (n.poc1 == c.poc0 && n.poc0 == c.poc1) return (bs_far(n.mv1, c.mv0) || bs_far(n.mv0, c.mv1)) ? 1 : 0;
        return 1;
    }
    if (c.pf != 3u && n.pf != 3u) {
        const int a = (c.pf & 1u) ? c.mv0 : c.mv1, ra = (c.pf & 1u) ? c.poc0 : c.poc1;
        const int b = (n.pf & 1u) ? n.mv0 : n.mv1, rb = (n.pf & 1u) ? n.poc0 : n.poc1;
        return ra == rb ? (bs_far(a, b) ? 1 : 0) : 1;
    }
    return 1;
}

// 16 lanes per recorded call of ff_hevc_deblocking_boundary_strengths; a lane takes the (edge, 4-sample segment) items k, k + 16, ... of its
// call: the block's top edge, its left edge, the prediction-block edges inside it (a 64x64 coding block has 256 items, an 8x8 transform block 4).
// Every array entry is written by exactly one item of one call.
__global__ __launch_bounds__(256) void boundary_strength_kernel(ohevc_bs_maps m, const ohevc_bs_call *__restrict__ calls, int ncalls,
                                                                unsigned char *__restrict__ vbs, unsigned char *__restrict__ hbs)
{
    const int ci = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (ci >= ncalls) return;
    const ohevc_bs_call cl = calls[ci];
    const int x0 = cl.x0, y0 = cl.y0, n = 1 << cl.log2_size, l2pu = m.log2_min_pu_size, l2tu = m.log2_min_tb_size;
    const int ctb_mask = (1 << m.log2_ctb_size) - 1;
    auto cbf = [&](int x, int y) { return m.cbf_luma[(size_t)(y >> l2tu) * m.min_tb_width + (x >> l2tu)]; };
    auto edge = [&](int xc, int yc, int xn, int yn) {        // a transform-block edge: current block at (xc, yc), its neighbour at (xn, yn)
        const BsField c = bs_load(m, xc >> l2pu, yc >> l2pu), nb = bs_load(m, xn >> l2pu, yn >> l2pu);
        if (c.pf == 0u || nb.pf == 0u) return 2;
        if (cbf(xc, yc) || cbf(xn, yn)) return 1;
        return bs_motion(c, nb);
    };
    bool top = false, left = false;
    if (y0 > 0 && (y0 & 7) == 0) {                            // the block's top edge, :821-858
        const bool bd_ctby = (y0 & ctb_mask) != 0;
        const bool bd_slice = (cl.flags & OHEVC_BS_ACROSS_SLICES) || !(cl.flags & OHEVC_BS_SLICE_UP);
        const bool bd_tiles = m.loop_filter_across_tiles || !(cl.flags & OHEVC_BS_TILE_UP);
        top = (bd_slice && bd_tiles) || bd_ctby;
    }
    if (x0 > 0 && (x0 & 7) == 0) {                            // its left edge, :861-898
        const bool bd_ctbx = (x0 & ctb_mask) != 0;
        const bool bd_slice = (cl.flags & OHEVC_BS_ACROSS_SLICES) || !(cl.flags & OHEVC_BS_SLICE_LEFT);
        const bool bd_tiles = m.loop_filter_across_tiles || !(cl.flags & OHEVC_BS_TILE_LEFT);
        left = (bd_slice && bd_tiles) || bd_ctbx;
    }
    // prediction-block edges inside the block: motion only, :900-940 (the neighbour of the edge at offset j is the entry at j - 8 - `top = curr` -
    // except the first, j = 8, whose neighbour is the entry at 7)
    const bool inner = cl.log2_size > l2pu && n > 8 && bs_load(m, x0 >> l2pu, y0 >> l2pu).pf != 0u;
    const int n4 = n >> 2, per_dir = inner ? n4 * ((n >> 3) - 1) : 0, total = 2 * n4 + 2 * per_dir;
    for (int k = sub; k < total; k += 16) {
        if (k < n4) {
            if (top) hbs[((x0 + 4 * k) + y0 * m.bs_width) >> 2] = (unsigned char)edge(x0 + 4 * k, y0, x0 + 4 * k, y0 - 1);
        } else if (k < 2 * n4) {
            const int i = 4 * (k - n4);
            if (left) vbs[(x0 + (y0 + i) * m.bs_width) >> 2] = (unsigned char)edge(x0, y0 + i, x0 - 1, y0 + i);
        } else if (k < 2 * n4 + per_dir) {                   // horizontal edges: column segment i, edge row j
            const int q = k - 2 * n4, i = 4 * (q % n4), j = 8 * (1 + q / n4);
            const BsField cur = bs_load(m, (x0 + i) >> l2pu, (y0 + j) >> l2pu), nb = bs_load(m, (x0 + i) >> l2pu, (y0 + (j == 8 ? 7 : j - 8)) >> l2pu);
            hbs[((x0 + i) + (y0 + j) * m.bs_width) >> 2] = (unsigned char)bs_motion(cur, nb);
        } else {                                             // vertical edges: row segment j, edge column i
            const int q = k - 2 * n4 - per_dir, j = 4 * (q % n4), i = 8 * (1 + q / n4);
            const BsField cur = bs_load(m, (x0 + i) >> l2pu, (y0 + j) >> l2pu), nb = bs_load(m, (x0 + (i == 8 ? 7 : i - 8)) >> l2pu, (y0 + j) >> l2pu);
            vbs[((x0 + i) + (y0 + j) * m.bs_width) >> 2] = (unsigned char)bs_motion(cur, nb);
        }
    }
}

// The motion field rebuilt from the luma motion-compensation jobs (ohevc_dev_motion_grid): one lane per (job, 4x4 unit of its at most
// 16x16 tile).  A job's integer source position and phase ARE its motion vector: sx = x + (mv.x >> 2), mx = mv.x & 3 (luma_mc_uni, hevc.c:1693-1703).
__global__ __launch_bounds__(256) void motion_grid_kernel(const ohevc_mc_job *__restrict__ jobs, int njobs, const ohevc_mc_job *__restrict__ more, int nmore,
                                                          unsigned char *__restrict__ grid, int grid_w, int grid_h, int log2_unit)
{
    const int t = blockIdx.x * 256 + threadIdx.x, ji = t >> 4, ux = (t & 3) << 2, uy = ((t >> 2) & 3) << 2;
    if (ji >= njobs + nmore) return;
    const ohevc_mc_job j = ji < njobs ? jobs[ji] : more[ji - njobs];      // (the ctx layer keeps its tiles in two arrays: one launch for both)
    const int um = (1 << log2_unit) - 1;
    if (j.plane != 0 || ux >= j.w || uy >= j.h || (((j.x + ux) | (j.y + uy)) & um)) return;
    const int gx = (j.x + ux) >> log2_unit, gy = (j.y + uy) >> log2_unit;
    if (gx >= grid_w || gy >= grid_h) return;
    auto mv = [](int s, int p, int m) { return (((s - p) << 2) | m) & 0xffff; };
    int *e = reinterpret_cast<int *>(grid + ((size_t)gy * grid_w + gx) * OHEVC_MOTION_GRID_ENTRY);
    const bool bi = (j.flags & OHEVC_MC_BI) != 0;
    e[0] = mv(j.sx0, j.x, j.mx0) | (mv(j.sy0, j.y, j.my0) << 16);
    e[1] = bi ? mv(j.sx1, j.x, j.mx1) | (mv(j.sy1, j.y, j.my1) << 16) : 0;
    e[2] = j.ref0; e[3] = bi ? j.ref1 : -1;
    e[4] = bi ? 3 : 1;
}

}  // namespace ohevc

// A device-to-device copy as a plain kernel: the picture's deblocked copy is made once per picture on the critical path of every frame end,
// and a launch is the cheapest thing to put there (no copy-engine hand-over, 3-4 us of host time).
namespace ohevc {
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, unsigned n16)
{
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
}  // namespace ohevc
extern "C" int ohevc_dev_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    using namespace ohevc;
    if (bytes == 0) return OHEVC_OK;
    OHEVC_REQUIRE(dst != nullptr && src != nullptr && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15) == 0 && (bytes >> 4) < 0xffffffffull,
                  "16-byte aligned buffers and size");
    const unsigned n16 = (unsigned)(bytes >> 4);
    const unsigned grid = std::min<unsigned>((n16 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u32x4 *>(src), static_cast<u32x4 *>(dst), n16);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

// Zeros as a plain kernel, for the same reason - and one more: on this runtime hipMemsetAsync (like hipMemcpyAsync) does not RETURN while the
// stream still waits for an event of another stream (ctx.hip, up_stream): the frame end's first clearing sat behind the picture's upload
// and, in inter pictures, behind the reference pictures' frame ends - 0.3 ms of a decoding thread per picture at 16 frame threads where a
// launch takes 4 us (OHEVC_TRACE=timing, profiles/r5z_timing_intra_only_16.txt).
namespace ohevc {
__global__ __launch_bounds__(256) void zero16_kernel(u32x4 *__restrict__ dst, unsigned n16)
{
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = u32x4{ 0u, 0u, 0u, 0u };
}
}  // namespace ohevc
extern "C" int ohevc_dev_zero(void *dst, size_t bytes, void *stream)
{
    using namespace ohevc;
    if (bytes == 0) return OHEVC_OK;
    OHEVC_REQUIRE(dst != nullptr && ((reinterpret_cast<uintptr_t>(dst) | bytes) & 15) == 0 && (bytes >> 4) < 0xffffffffull, "16-byte aligned buffer and size");
    const unsigned n16 = (unsigned)(bytes >> 4);
    const unsigned grid = std::min<unsigned>((n16 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(zero16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<u32x4 *>(dst), n16);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_dev_motion_grid2(const ohevc_mc_job *jobs, int njobs, const ohevc_mc_job *more, int nmore, uint8_t *grid, int grid_width, int grid_height,
                                      int log2_unit, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(njobs >= 0 && nmore >= 0 && grid_width > 0 && grid_height > 0 && log2_unit >= 2 && log2_unit <= 5, "grid geometry");
    if (njobs + nmore == 0) return OHEVC_OK;
    OHEVC_REQUIRE((njobs == 0 || jobs != nullptr) && (nmore == 0 || more != nullptr) && grid != nullptr && (reinterpret_cast<uintptr_t>(grid) & 3) == 0, "null / misaligned array");
    hipLaunchKernelGGL(motion_grid_kernel, dim3((unsigned)(((long long)(njobs + nmore) * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), jobs, njobs, more, nmore,
                       grid, grid_width, grid_height, log2_unit);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}
extern "C" int ohevc_dev_motion_grid(const ohevc_mc_job *jobs, int njobs, uint8_t *grid, int grid_width, int grid_height, int log2_unit, void *stream)
{
    return ohevc_dev_motion_grid2(jobs, njobs, nullptr, 0, grid, grid_width, grid_height, log2_unit, stream);
}

extern "C" int ohevc_dev_boundary_strengths(const ohevc_bs_maps *maps, const ohevc_bs_call *calls, int ncalls, uint8_t *vertical_bs, uint8_t *horizontal_bs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(maps != nullptr && ncalls >= 0, "null argument");
    if (ncalls == 0) return OHEVC_OK;
    OHEVC_REQUIRE(calls != nullptr && vertical_bs != nullptr && horizontal_bs != nullptr && maps->mvf != nullptr && maps->cbf_luma != nullptr, "null array");
    OHEVC_REQUIRE(maps->mvf_stride >= 20 && (maps->mvf_stride & 3) == 0 && (maps->off_mv & 3) == 0 && (maps->off_poc & 3) == 0 &&
                  (maps->pred_flag_bytes == 1 || (maps->pred_flag_bytes == 4 && (maps->off_pred_flag & 3) == 0)), "motion-field entry layout");
    OHEVC_REQUIRE(maps->log2_min_pu_size >= 2 && maps->log2_min_tb_size >= 2 && maps->log2_ctb_size >= 4 && maps->log2_ctb_size <= 6 && maps->bs_width > 0 &&
                  maps->min_pu_width > 0 && maps->min_tb_width > 0, "picture geometry");
    hipLaunchKernelGGL(boundary_strength_kernel, dim3((ncalls + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), *maps, calls, ncalls, vertical_bs, horizontal_bs);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" long ohevc_debug_sao_band_above_range(int reset)
{
    unsigned *d = nullptr, h = 0;
    if (hipMalloc((void **)&d, sizeof(unsigned)) != hipSuccess) return -1;
    (void)hipDeviceSynchronize();                     // every stream's SAO launches have finished
    hipLaunchKernelGGL(ohevc::sao_band_above_range_fetch, dim3(1), dim3(1), 0, nullptr, d, reset);
    const bool ok = hipMemcpy(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? (long)h : -1;
}

extern "C" int ohevc_debug_set_sao_variant(int variant)
{
    const int old = ohevc::g_sao_variant;
    ohevc::g_sao_variant = variant;
    return old;
}

extern "C" int ohevc_dev_deblock_batch(const ohevc_plane planes[3], int bit_depth, const ohevc_dbk_job *jobs, int njobs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(jobs != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0, "jobs must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (int)(((long long)njobs * 8 + 255) / 256);
    if (bit_depth == 8) hipLaunchKernelGGL((deblock_kernel<uint8_t>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, bit_depth);
    else                hipLaunchKernelGGL((deblock_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, bit_depth);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

static int g_deblock_variant = 0;      // 0: a lane per luma segment (shipped), 1: a lane per line (rounds 2-3)
static long long g_deblock_segment_launches;
extern "C" int ohevc_debug_set_deblock_variant(int v) { const int prev = g_deblock_variant; g_deblock_variant = v; return prev; }
extern "C" long long ohevc_debug_deblock_segment_launches(void) { return g_deblock_segment_launches; }

extern "C" int ohevc_dev_deblock_maps(const ohevc_plane planes[3], int bit_depth, const ohevc_dbk_maps *m, int vertical, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr && m != nullptr, "null argument");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(m->width > 0 && m->height > 0 && m->log2_ctb_size >= 4 && m->log2_ctb_size <= 6 && m->log2_min_cb_size >= 3 &&
                  m->chroma_format_idc >= 0 && m->chroma_format_idc <= 3, "picture geometry");
    OHEVC_REQUIRE(m->horizontal_bs && m->vertical_bs && m->bs_width > 0 && m->qp_y_tab && m->min_cb_width > 0 && m->deblock && m->deblock_stride >= 2,
                  "deblocking maps");
    OHEVC_REQUIRE(!m->is_pcm || (m->min_pu_width > 0 && m->min_pu_height > 0 && m->log2_min_pu_size >= 2), "pcm map");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    const int hs = m->chroma_format_idc == 1 || m->chroma_format_idc == 2, vs = m->chroma_format_idc == 1;
    // the loops of deblocking_filter_CTB step 8 (luma) / 8h, 8v (chroma) samples from 0 while below the picture size
    const int luma_uw = (m->width + 7) >> 3, luma_uh = (m->height + 7) >> 3;
    const int chroma_uw = m->chroma_format_idc ? (m->width + (8 << hs) - 1) >> (3 + hs) : 0, chroma_uh = m->chroma_format_idc ? (m->height + (8 << vs) - 1) >> (3 + vs) : 0;
    const int luma_units = luma_uw * luma_uh, chroma_units = chroma_uw * chroma_uh;
    const long long threads = ((long long)((luma_units + 7) & ~7) + 2ll * ((chroma_units + 7) & ~7)) * 8;      // each plane's units padded to whole wavefronts
    const int grid = (int)((threads + 255) / 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one lane per luma segment (dbk4_kernel.hpp): 16-bit arithmetic holds up to 10-bit samples; whole 8x8 blocks (every picture the decoder
    // makes: the minimum coding block is 8x8) and rows that can be read a dword at a time
    const bool segments = g_deblock_variant == 0 && bit_depth <= 10 && ((m->width | m->height) & 7) == 0 &&
                          ((reinterpret_cast<uintptr_t>(ps.data[0]) | (unsigned)ps.stride[0]) & 3) == 0;
    if (segments) {
        const int per_row = vertical ? m->width >> 3 : m->width >> 2;
        const int nseg = per_row * (vertical ? m->height >> 2 : m->height >> 3);
        const int luma_groups = (nseg + 255) / 256;
        const int chroma_groups = (int)((2ll * ((chroma_units + 7) & ~7) * 8 + 255) / 256);
        if (bit_depth == 8) hipLaunchKernelGGL((deblock_maps_segments_kernel<uint8_t>), dim3(luma_groups + chroma_groups), dim3(256), 0, st, ps, *m, vertical != 0, bit_depth, luma_groups, nseg, per_row, chroma_units, chroma_uw);
        else                hipLaunchKernelGGL((deblock_maps_segments_kernel<uint16_t>), dim3(luma_groups + chroma_groups), dim3(256), 0, st, ps, *m, vertical != 0, bit_depth, luma_groups, nseg, per_row, chroma_units, chroma_uw);
        OHEVC_HIP_TRY(hipGetLastError());
        g_deblock_segment_launches++;
        return OHEVC_OK;
    }
    if (bit_depth == 8) hipLaunchKernelGGL((deblock_maps_kernel<uint8_t>), dim3(grid), dim3(256), 0, st, ps, *m, vertical != 0, bit_depth, luma_units, chroma_units, luma_uw, chroma_uw);
    else                hipLaunchKernelGGL((deblock_maps_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, ps, *m, vertical != 0, bit_depth, luma_units, chroma_units, luma_uw, chroma_uw);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

static int sao_launch(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3],
                      int bit_depth, const ohevc_sao_job *jobs, int njobs, const ohevc_sao_bypass *bypass, void *stream, int n_wide)
{
    using namespace ohevc;
    OHEVC_REQUIRE(dst != nullptr && src != nullptr && lagged != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(jobs != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0, "jobs must be 16-byte aligned");
    ohevc_sao_bypass bp = {};
    if (bypass && bypass->map) {
        bp = *bypass;
        OHEVC_REQUIRE(bp.stride > 0 && bp.log2_min_pu_size >= 2 && bp.log2_min_pu_size <= 6 && (bp.chroma_hshift | 1) == 1 &&
                      (bp.chroma_vshift | 1) == 1, "bad bypass map description");
    }
    PlaneSet pd, psrc, plag;
    int rc = make_plane_set(dst, pd, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    rc = make_plane_set(src, psrc, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    rc = make_plane_set(lagged, plag, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // every block goes through exactly one of the two kernels (sao_wide_ok, evaluated by both on the job record): the jobs live in
    // device memory, the host cannot sort them.  Blocks are disjoint and both kernels read `src` only, so the order does not matter.
    // n_wide >= 0: the caller sorted the jobs (ohevc_dev_sao_batch_sorted) - the first n_wide go to the wide kernel, the rest to the other
    const int nw = n_wide >= 0 ? n_wide : njobs;
    if (!(g_sao_variant & 2) && nw > 0) {
        // ohevc_debug_set_sao_variant(4): list order; (8): one contiguous eighth of the list per XCD; default: runs of 16 list entries per XCD
        const int spread = (g_sao_variant & 4) ? 1 : (g_sao_variant & 8) ? 2 : 0, gw = spread == 1 ? nw : spread == 2 ? (nw + 7) & ~7 : (nw + 127) & ~127;
        const int sp = spread | (g_sao_variant & 16);       // (16): interior edge-class blocks through the general loop too (the A/B of sao_edge_plain)
        if (bit_depth == 8) hipLaunchKernelGGL((sao_wide_kernel<uint8_t>), dim3(gw), dim3(kSaoWideThreads), 0, st, pd, psrc, jobs, nw, bit_depth, bp, sp);
        else                hipLaunchKernelGGL((sao_wide_kernel<uint16_t>), dim3(gw), dim3(kSaoWideThreads), 0, st, pd, psrc, jobs, nw, bit_depth, bp, sp);
    }
    if (n_wide >= 0 && !(g_sao_variant & 2)) { jobs += n_wide; njobs -= n_wide; }
    if (njobs <= 0) { OHEVC_HIP_TRY(hipGetLastError()); return OHEVC_OK; }
    const int gs = (g_sao_variant & 4) ? njobs : (njobs + 7) & ~7;
    if (g_sao_variant & 1) {
        if (bit_depth == 8) hipLaunchKernelGGL((sao_kernel<uint8_t, true>), dim3(gs), dim3(256), 0, st, pd, psrc, plag, jobs, njobs, bit_depth, bp, g_sao_variant);
        else                hipLaunchKernelGGL((sao_kernel<uint16_t, true>), dim3(gs), dim3(256), 0, st, pd, psrc, plag, jobs, njobs, bit_depth, bp, g_sao_variant);
    } else {
        if (bit_depth == 8) hipLaunchKernelGGL((sao_kernel<uint8_t, false>), dim3(gs), dim3(256), 0, st, pd, psrc, plag, jobs, njobs, bit_depth, bp, g_sao_variant);
        else                hipLaunchKernelGGL((sao_kernel<uint16_t, false>), dim3(gs), dim3(256), 0, st, pd, psrc, plag, jobs, njobs, bit_depth, bp, g_sao_variant);
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_dev_sao_batch_bypass(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3],
                                          int bit_depth, const ohevc_sao_job *jobs, int njobs, const ohevc_sao_bypass *bypass, void *stream)
{
    return sao_launch(dst, src, lagged, bit_depth, jobs, njobs, bypass, stream, -1);
}

extern "C" int ohevc_dev_sao_batch_sorted(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3], int bit_depth,
                                          const ohevc_sao_job *jobs, int n_wide, int n_other, const ohevc_sao_bypass *bypass, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(n_wide >= 0 && n_other >= 0, "job counts");
    return sao_launch(dst, src, lagged, bit_depth, jobs, n_wide + n_other, bypass, stream, n_wide);
}

// host twin of sao_wide_ok (same conditions, same order)
extern "C" int ohevc_sao_job_is_wide(const ohevc_sao_job *jb, const ohevc_plane dst[3], const ohevc_plane src[3], int bit_depth)
{
    if (!jb || !dst || !src || jb->plane > 2) return 0;
    const int ps = bit_depth > 8 ? 2 : 1, ppl = 16 / ps, w = jb->w, h = jb->h, np = w / ppl;
    const bool band = jb->type == OHEVC_SAO_BAND;
    const ohevc_plane &sp = src[jb->plane], &dp = dst[jb->plane];
    const bool lagq = (jb->quirks & (OHEVC_SAO_LAG_BELOW | OHEVC_SAO_LAG_ABOVE | OHEVC_SAO_LAG_MID)) != 0 && !band && jb->klass != 1 && jb->x + w < sp.width;
    bool fits = true;
    for (int k = 0; k < 5; k++) fits = fits && jb->offset_val[k] >= -128 && jb->offset_val[k] < 128;
    const uintptr_t sbase = reinterpret_cast<uintptr_t>(sp.data) + (size_t)jb->y * sp.stride + (size_t)jb->x * ps;
    const uintptr_t dbase = reinterpret_cast<uintptr_t>(dp.data) + (size_t)jb->y * dp.stride + (size_t)jb->x * ps;
    return (w % ppl) == 0 && np >= 1 && np <= 8 && (np & (np - 1)) == 0 && h <= 64 && !lagq && fits &&
           ((sbase | dbase | (unsigned)sp.stride | (unsigned)dp.stride) & 15) == 0;
}

extern "C" int ohevc_dev_sao_batch_lagged(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3],
                                          int bit_depth, const ohevc_sao_job *jobs, int njobs, void *stream)
{
    return ohevc_dev_sao_batch_bypass(dst, src, lagged, bit_depth, jobs, njobs, nullptr, stream);
}

extern "C" int ohevc_dev_sao_batch(const ohevc_plane dst[3], const ohevc_plane src[3], int bit_depth,
                                   const ohevc_sao_job *jobs, int njobs, void *stream)
{
    // without a lagged picture the flag has nothing to read from: jobs carrying it read the deblocked copy
    return ohevc_dev_sao_batch_bypass(dst, src, src, bit_depth, jobs, njobs, nullptr, stream);
}
