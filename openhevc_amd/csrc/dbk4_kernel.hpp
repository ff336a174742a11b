// dbk4_kernel.hpp -- luma deblocking from the decoder's maps, ONE LANE PER 4-LINE SEGMENT, packed 16-bit arithmetic.  Included by
// filter_kernels.hip (after kDbkTc).
//
// What hevc_{h,v}_loop_filter_luma (hevcdsp_template.c:1629-1723) decides, it decides per 4-line segment (dE, dEp, dEq from lines 0 and 3,
// the tc of the segment, the pcm / bypass flags of the segment's two sides) - and what deblocking_filter_CTB (hevc_filter.c:345-581)
// derives per edge is per segment too (bS per 4 samples, the QP / offsets of the 8x8 block around it).  The first form (deblock_maps_kernel)
// gives every LINE a lane: eight lanes repeat one derivation, the decisions travel through eight cross-lane moves per line, and every line
// unpacks and repacks its own eight samples - about 400 instructions per line, the kernel is bound by instruction issue (r4q: the same
// 0.82 Tpixel/s at 8 and at 10 bit).  Here a lane owns the segment:
//   - one derivation per segment, no cross-lane traffic: the decisions are lane-local;
//   - the four lines are two pairs, each pair one register per sample position (line a in the low, line b in the high 16 bits): the
//     filters run on v_pk_{add,sub,mul_lo,max,min,ashrrev}_i16 - one instruction per two lines.  Every intermediate of the luma filters
//     fits 16 bits up to 10-bit samples (the largest: 9 * (q0 - p0) - 3 * (q1 - p1) + 8 <= 12284); deeper pictures keep the first form;
//   - vertical edges: a lane loads 4 rows x 8 samples (8 / 16 bytes each; the lanes of a wavefront are 64 neighbouring edges: 512 / 1024
//     contiguous bytes per row), and byte permutes (v_perm_b32) transpose them into the pairs; horizontal edges: 8 rows x 4 samples,
//     already one sample position per row.
// Chroma keeps the first form (two of three chroma edges are not filtered at all: bS < 2), in the same launch.
#pragma once

namespace ohevc {

typedef short pk16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pk16 pk_splat(int v) { return pk16{ (short)v, (short)v }; }
__device__ __forceinline__ pk16 pk_abs(pk16 v) { return __builtin_elementwise_max(v, pk_splat(0) - v); }
__device__ __forceinline__ pk16 pk_clip(pk16 v, pk16 lo, pk16 hi) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }
__device__ __forceinline__ pk16 pk_select(pk16 mask, pk16 yes, pk16 no) { return (yes & mask) | (no & ~mask); }
__device__ __forceinline__ pk16 pk_from(unsigned v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ unsigned pk_bits(pk16 v) { return __builtin_bit_cast(unsigned, v); }

// The two luma filters on one pair of lines.  s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3; everything lane-uniform arrives as a scalar.  SIDE_P / SIDE_Q:
// which side this call writes (the caller branches once per side for both pairs).
template <bool SIDE_P>
__device__ __forceinline__ void dbk4_strong(const pk16 in[8], pk16 s[8], int tc)
{      // in: the samples before the filter (the other side's call may have written s already)
    const pk16 p3 = in[0], p2 = in[1], p1 = in[2], p0 = in[3], q0 = in[4], q1 = in[5], q2 = in[6], q3 = in[7];
    const pk16 hi = pk_splat(2 * tc), lo = pk_splat(-2 * tc), k2 = pk_splat(2), k3 = pk_splat(3), k4 = pk_splat(4);
    const pk16 mid = p0 + q0;                                        // shared by all six sums
    if (SIDE_P) {
        s[3] = p0 + pk_clip(((p2 + p1 + p1 + mid + mid + q1 + k4) >> k3) - p0, lo, hi);          // (p2 + 2 p1 + 2 p0 + 2 q0 + q1 + 4) >> 3
        s[2] = p1 + pk_clip(((p2 + p1 + mid + k2) >> k2) - p1, lo, hi);
        s[1] = p2 + pk_clip(((p3 + p3 + p2 + p2 + p2 + p1 + mid + k4) >> k3) - p2, lo, hi);      // (2 p3 + 3 p2 + p1 + p0 + q0 + 4) >> 3
    } else {
        s[4] = q0 + pk_clip(((p1 + mid + mid + q1 + q1 + q2 + k4) >> k3) - q0, lo, hi);
        s[5] = q1 + pk_clip(((mid + q1 + q2 + k2) >> k2) - q1, lo, hi);
        s[6] = q2 + pk_clip(((q3 + q3 + q2 + q2 + q2 + q1 + mid + k4) >> k3) - q2, lo, hi);
    }
}

// The normal filter's per-line part: delta (clipped) and the line mask - all ones in the half whose line is filtered, |delta0| < 10 tc
// (hevcdsp_template.c:1691).
struct Dbk4Delta { pk16 delta, line; };
__device__ __forceinline__ Dbk4Delta dbk4_delta(const pk16 s[8], int tc)
{
    const pk16 p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5];
    const pk16 delta0 = (pk_splat(9) * (q0 - p0) - pk_splat(3) * (q1 - p1) + pk_splat(8)) >> pk_splat(4);
    return Dbk4Delta{ pk_clip(delta0, pk_splat(-tc), pk_splat(tc)), (pk_abs(delta0) - pk_splat(10 * tc)) >> pk_splat(15) };
}
// sample 0 of a side (p0: SIGN +1, q0: SIGN -1), sample 1 of a side (n2 n1 n0 = p2 p1 p0 / q2 q1 q0)
template <int SIGN>
__device__ __forceinline__ pk16 dbk4_normal0(pk16 n0, const Dbk4Delta &d, int maxv)
{
    return pk_select(d.line, pk_clip(SIGN > 0 ? n0 + d.delta : n0 - d.delta, pk_splat(0), pk_splat(maxv)), n0);
}
template <int SIGN>
__device__ __forceinline__ pk16 dbk4_normal1(pk16 n2, pk16 n1, pk16 n0, const Dbk4Delta &d, int tc, int maxv)
{
    const pk16 k1 = pk_splat(1), half = ((n2 + n0 + k1) >> k1) - n1;
    const pk16 step = pk_clip((SIGN > 0 ? half + d.delta : half - d.delta) >> k1, pk_splat(-(tc >> 1)), pk_splat(tc >> 1));
    return pk_select(d.line, pk_clip(n1 + step, pk_splat(0), pk_splat(maxv)), n1);
}

// One segment.  VERTICAL: the edge at column x (a multiple of 8), lines y .. y + 3; else the edge at row y (a multiple of 8), columns
// x .. x + 3.  tc_tab: H.265 table 8-12's tc' column in LDS (the kernel copies it there: a lookup that costs an LDS read instead of a third
// round trip to memory behind the QP loads).
template <typename Pixel, bool VERTICAL>
__device__ __forceinline__ void deblock_maps_luma4(unsigned char *__restrict__ pbase, int pstride, const ohevc_dbk_maps &m, int bit_depth, int x, int y,
                                                   const unsigned char *tc_tab)
{
    constexpr bool WIDE = sizeof(Pixel) == 2;
    const unsigned char *const bsm = VERTICAL ? m.vertical_bs : m.horizontal_bs;
    const int bs = bsm[(unsigned)(x + y * m.bs_width) >> 2];
    if (!bs) return;
    // ---- second round: samples, the two pcm / bypass flags, the two QPs, the CTB's offsets - independent, issued together
    unsigned char *const pix = pbase + (size_t)(VERTICAL ? y : y - 4) * pstride + (size_t)(VERTICAL ? x - 4 : x) * sizeof(Pixel);
    u32x4 r16[4] = {};                                              // 16-bit samples, vertical: 4 rows x 8 samples
    u32x2 r8[8];                                                    // 8-bit vertical: 4 rows x 8 samples; horizontal (either depth): 8 rows x 4 samples
#pragma unroll
    for (int k = 0; k < 8; k++) r8[k] = u32x2{ 0, 0 };
    if constexpr (VERTICAL) {
        if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < 4; k++) r16[k] = *reinterpret_cast<const u32x4 *>(pix + (size_t)k * pstride);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) r8[k] = *reinterpret_cast<const u32x2 *>(pix + (size_t)k * pstride);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if constexpr (WIDE) r8[k] = *reinterpret_cast<const u32x2 *>(pix + (size_t)k * pstride);
            else                r8[k].x = *reinterpret_cast<const unsigned *>(pix + (size_t)k * pstride);
        }
    }
    const int px = VERTICAL ? x - 1 : x, py = VERTICAL ? y : y - 1;
    const bool has_pcm = m.is_pcm != nullptr;
    const unsigned char *const pcm_map = has_pcm ? m.is_pcm : reinterpret_cast<const unsigned char *>(m.qp_y_tab);
    // get_pcm (hevc_filter.c:325-338): the Q side lies inside the picture, the P side too (x >= 8 / y >= 8)
    const int pcm_p = pcm_map[has_pcm ? (unsigned)((py >> m.log2_min_pu_size) * m.min_pu_width + (px >> m.log2_min_pu_size)) : 0u];
    const int pcm_q = pcm_map[has_pcm ? (unsigned)((y >> m.log2_min_pu_size) * m.min_pu_width + (x >> m.log2_min_pu_size)) : 0u];
    // the QP, the offsets: those of the 8-sample edge the segment belongs to (its first segment's, hevc_filter.c:396-399,492-495)
    const int ex = VERTICAL ? x : x & ~7, ey = VERTICAL ? y & ~7 : y;
    const int epx = VERTICAL ? ex - 1 : ex, epy = VERTICAL ? ey : ey - 1;
    const int qp_p = (int)m.qp_y_tab[(unsigned)((epx >> m.log2_min_cb_size) + (epy >> m.log2_min_cb_size) * m.min_cb_width)];
    const int qp_q = (int)m.qp_y_tab[(unsigned)((ex >> m.log2_min_cb_size) + (ey >> m.log2_min_cb_size) * m.min_cb_width)];
    const int log2_ctb = m.log2_ctb_size, ctb_w = (m.width + (1 << log2_ctb) - 1) >> log2_ctb;
    const int ctb_row = (y >> log2_ctb) * ctb_w;
    int cx_beta = ex >> log2_ctb, cx_tc = (VERTICAL ? ex : ex + 8) >> log2_ctb;          // (see deblock_maps_kernel: which CTB's offsets an edge gets)
    cx_beta = cx_beta < ctb_w - 1 ? cx_beta : ctb_w - 1;
    cx_tc = cx_tc < ctb_w - 1 ? cx_tc : ctb_w - 1;
    const int beta_offset = (int)m.deblock[(unsigned)((cx_beta + ctb_row) * m.deblock_stride)];
    const int tc_offset = (int)m.deblock[(unsigned)((cx_tc + ctb_row) * m.deblock_stride + 1)];
    const bool no_p = has_pcm && pcm_p, no_q = has_pcm && pcm_q;
    const int qp_y = (qp_p + qp_q + 1) >> 1;
    int qb = qp_y + beta_offset, qt = qp_y + 2 * (bs - 1) + (tc_offset >> 1 << 1);
    qb = qb < 0 ? 0 : qb > 51 ? 51 : qb;
    qt = qt < 0 ? 0 : qt > 53 ? 53 : qt;
    const int shift = bit_depth - 8;
    const int beta = (qb < 16 ? 0 : qb < 29 ? qb - 10 : 2 * qb - 38) << shift;
    const int tc = (int)tc_tab[qt] << shift;
    const int maxv = (1 << bit_depth) - 1;

    // ---- the two pairs: a[k] = lines 0 | 1, b[k] = lines 2 | 3 of sample position k (p3 p2 p1 p0 q0 q1 q2 q3)
    pk16 a[8], b[8];
    if constexpr (VERTICAL) {
        if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < 4; k++) {           // dword k of a row holds positions 2k (low half) and 2k + 1
                const unsigned w0 = k == 0 ? r16[0].x : k == 1 ? r16[0].y : k == 2 ? r16[0].z : r16[0].w, w1 = k == 0 ? r16[1].x : k == 1 ? r16[1].y : k == 2 ? r16[1].z : r16[1].w;
                const unsigned w2 = k == 0 ? r16[2].x : k == 1 ? r16[2].y : k == 2 ? r16[2].z : r16[2].w, w3 = k == 0 ? r16[3].x : k == 1 ? r16[3].y : k == 2 ? r16[3].z : r16[3].w;
                a[2 * k] = pk_from(perm_b32(w1, w0, 0x05040100u)); a[2 * k + 1] = pk_from(perm_b32(w1, w0, 0x07060302u));
                b[2 * k] = pk_from(perm_b32(w3, w2, 0x05040100u)); b[2 * k + 1] = pk_from(perm_b32(w3, w2, 0x07060302u));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {           // byte k & 3 of dword k >> 2 of a row
                const unsigned sel = 0x0c000c00u | (unsigned)(k & 3) | ((unsigned)(4 + (k & 3)) << 16);
                a[k] = pk_from(perm_b32(k < 4 ? r8[1].x : r8[1].y, k < 4 ? r8[0].x : r8[0].y, sel));
                b[k] = pk_from(perm_b32(k < 4 ? r8[3].x : r8[3].y, k < 4 ? r8[2].x : r8[2].y, sel));
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {               // row k = position k, four columns
            if constexpr (WIDE) { a[k] = pk_from(r8[k].x); b[k] = pk_from(r8[k].y); }
            else                { a[k] = pk_from(perm_b32(0u, r8[k].x, 0x0c010c00u)); b[k] = pk_from(perm_b32(0u, r8[k].x, 0x0c030c02u)); }
        }
    }
    // ---- the segment's decisions, from lines 0 and 3 (hevcdsp_template.c:1646-1665): low half = line 0, high half = line 3
    pk16 e[8];
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = pk_from(perm_b32(pk_bits(b[k]), pk_bits(a[k]), 0x07060100u));
    const pk16 dp = pk_abs(e[1] - e[2] - e[2] + e[3]), dq = pk_abs(e[6] - e[5] - e[5] + e[4]);
    const int dp0 = dp.x, dp3 = dp.y, dq0 = dq.x, dq3 = dq.y;
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const pk16 flat = pk_abs(e[0] - e[3]) + pk_abs(e[7] - e[4]), step = pk_abs(e[3] - e[4]);
    const int tc25 = (tc * 5 + 1) >> 1, beta3 = beta >> 3, beta2 = beta >> 2;
    const bool strong = flat.x < beta3 && flat.y < beta3 && step.x < tc25 && step.y < tc25 && (d0 << 1) < beta2 && (d3 << 1) < beta2;
    if (strong) {
        pk16 a0[8], b0[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { a0[k] = a[k]; b0[k] = b[k]; }
        if (!no_p) { dbk4_strong<true>(a0, a, tc); dbk4_strong<true>(b0, b, tc); }
        if (!no_q) { dbk4_strong<false>(a0, a, tc); dbk4_strong<false>(b0, b, tc); }
    } else {
        const int side = (beta + (beta >> 1)) >> 3;
        const bool two_p = dp0 + dp3 < side, two_q = dq0 + dq3 < side;
        const Dbk4Delta da = dbk4_delta(a, tc), db = dbk4_delta(b, tc);
        const pk16 ap0 = a[3], aq0 = a[4], bp0 = b[3], bq0 = b[4];              // (sample 1 of a side is computed from the UNFILTERED sample 0)
        if (!no_p) {
            a[3] = dbk4_normal0<1>(ap0, da, maxv); b[3] = dbk4_normal0<1>(bp0, db, maxv);
            if (two_p) { a[2] = dbk4_normal1<1>(a[1], a[2], ap0, da, tc, maxv); b[2] = dbk4_normal1<1>(b[1], b[2], bp0, db, tc, maxv); }
        }
        if (!no_q) {
            a[4] = dbk4_normal0<-1>(aq0, da, maxv); b[4] = dbk4_normal0<-1>(bq0, db, maxv);
            if (two_q) { a[5] = dbk4_normal1<-1>(a[6], a[5], aq0, da, tc, maxv); b[5] = dbk4_normal1<-1>(b[6], b[5], bq0, db, tc, maxv); }
        }
    }
    // ---- back: positions 1 .. 6 can have changed; vertical edges store whole rows (samples -4 and +3 unchanged: no other edge of the
    // pass comes within 4 samples), horizontal edges the six rows
    if constexpr (VERTICAL) {
        if constexpr (WIDE) {
            u32x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned l0 = perm_b32(pk_bits(a[2 * k + 1]), pk_bits(a[2 * k]), 0x05040100u), l1 = perm_b32(pk_bits(a[2 * k + 1]), pk_bits(a[2 * k]), 0x07060302u);
                const unsigned l2 = perm_b32(pk_bits(b[2 * k + 1]), pk_bits(b[2 * k]), 0x05040100u), l3 = perm_b32(pk_bits(b[2 * k + 1]), pk_bits(b[2 * k]), 0x07060302u);
                if (k == 0) { o[0].x = l0; o[1].x = l1; o[2].x = l2; o[3].x = l3; }
                if (k == 1) { o[0].y = l0; o[1].y = l1; o[2].y = l2; o[3].y = l3; }
                if (k == 2) { o[0].z = l0; o[1].z = l1; o[2].z = l2; o[3].z = l3; }
                if (k == 3) { o[0].w = l0; o[1].w = l1; o[2].w = l2; o[3].w = l3; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) *reinterpret_cast<u32x4 *>(pix + (size_t)k * pstride) = o[k];
        } else {
            // bytes: first the two positions of a dword half side by side for both lines of the pair, then the halves of one line together
            u32x2 o[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {          // h: the dword of the row (positions 4h .. 4h + 3)
                const unsigned a01 = perm_b32(pk_bits(a[4 * h + 1]), pk_bits(a[4 * h]), 0x06020400u), a23 = perm_b32(pk_bits(a[4 * h + 3]), pk_bits(a[4 * h + 2]), 0x06020400u);
                const unsigned b01 = perm_b32(pk_bits(b[4 * h + 1]), pk_bits(b[4 * h]), 0x06020400u), b23 = perm_b32(pk_bits(b[4 * h + 3]), pk_bits(b[4 * h + 2]), 0x06020400u);
                const unsigned l0 = perm_b32(a23, a01, 0x05040100u), l1 = perm_b32(a23, a01, 0x07060302u);
                const unsigned l2 = perm_b32(b23, b01, 0x05040100u), l3 = perm_b32(b23, b01, 0x07060302u);
                if (h == 0) { o[0].x = l0; o[1].x = l1; o[2].x = l2; o[3].x = l3; }
                else        { o[0].y = l0; o[1].y = l1; o[2].y = l2; o[3].y = l3; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) *reinterpret_cast<u32x2 *>(pix + (size_t)k * pstride) = o[k];
        }
    } else {
#pragma unroll
        for (int k = 1; k < 7; k++) {
            if constexpr (WIDE) *reinterpret_cast<u32x2 *>(pix + (size_t)k * pstride) = u32x2{ pk_bits(a[k]), pk_bits(b[k]) };
            else                *reinterpret_cast<unsigned *>(pix + (size_t)k * pstride) = perm_b32(pk_bits(b[k]), pk_bits(a[k]), 0x06040200u);
        }
    }
}

}  // namespace ohevc
