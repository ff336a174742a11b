// intra_pack.hpp -- intra prediction + the block's own residual, N lanes per N x N block: 16 / 8 / 4 / 2 blocks share a wavefront.
// (included by tu_kernels.hip behind the residual bodies it reuses)
//
// The first intra kernel (intra_body.hpp) gives a block a whole wavefront: a 4x4 block is 16 samples on 64 lanes, the availability cascade of
// hevcpred_template.c:251-286 is a chain of LDS hand-offs (~10 barriers), and the block's residual - the other half of what
// hls_transform_unit does per transform block (hevc.c:1214-1215, 1260-1290) - reads the prediction back from memory behind a cache
// invalidate.  A dependency level of a picture lasts as long as one block, so that chain of 7-9 dependent memory round trips was the
// frame-end hook.  Here:
//   * lane (g, i) = row i of block g.  It loads top[i], top[N + i], left[i], left[N + i] and the corner - five loads, issued together,
//     whose ADDRESSES already contain the substitution rules: an unavailable group (below-left, left, corner, above, above-right in the
//     reference's scan order) takes the last sample of the nearest available group before it, else the first sample of the first
//     available group after it, else 1 << (bit_depth - 1): hevcpred_template.c:251-286 in closed form, no hand-off;
//   * the coefficient block is requested in the same round (its address comes from the residual record, read next to the job record);
//   * smoothing (:289-327) is one pass over the block's LDS arrays, the predictors (:359-537) produce row i in registers;
//   * the residual row comes out of the same lane (the batched bodies' lane map IS lane = (block, row): tu_idct_add_body, tu_rows_body; the
//     4x4 transforms in a row form), is added in registers and the row is stored once: no prediction store, no read-back.
// Three dependent memory round trips per level (records -> samples + coefficients -> store).  Constrained-intra jobs keep the first kernel
// (their substitution walk is sequential, hevcpred_template.c:185-249).
#pragma once

namespace ohevc {

template <int LOG2N> struct IntraPackLayout {
    static constexpr int N = 1 << LOG2N, G = 64 / N;       // lanes per block, blocks per wavefront
    static constexpr int ARR = 2 * N + 2;                  // top / left: element k (-1 .. 2N-1) at [k + 1]; one pad
    static constexpr int REF = 3 * N + 2;                  // angular reference: element k (-N .. 2N+1) at [k + N]
    static constexpr int INTS = 4 * ARR + REF;             // top, left, filtered top, filtered left, ref
};
constexpr int kIntraPackInts = 16 * IntraPackLayout<2>::INTS;      // the 4x4 form is the largest: 864 ints per wavefront
static_assert(IntraPackLayout<3>::G * IntraPackLayout<3>::INTS <= kIntraPackInts && IntraPackLayout<4>::G * IntraPackLayout<4>::INTS <= kIntraPackInts &&
              IntraPackLayout<5>::G * IntraPackLayout<5>::INTS <= kIntraPackInts, "LDS arrays of one wavefront");

#define PACK_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// row i of the 4x4 inverse DCT / DST-VII (hevcdsp_template.c:170-222): pass 1 produces row i of the intermediate from all four
// columns (the weights T[k][i] depend on the lane), pass 2 is the 4-point transform of that row
template <bool DST>
__device__ __forceinline__ void tu4_row(const u32x4 a, const u32x4 b, const int i, const int bit_depth, int *res)
{
    const unsigned raw[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    int c[4][4];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c[k / 2][(k % 2) * 2]     = (int)(short)(raw[k] & 0xffffu);
        c[k / 2][(k % 2) * 2 + 1] = (int)raw[k] >> 16;
    }
    int w0, w1, w2, w3;                                   // T[k][i], k = 0..3
    if constexpr (DST) {
        w0 = i == 0 ? 29 : i == 1 ? 55 : i == 2 ? 74 : 84;
        w1 = i == 0 ? 74 : i == 1 ? 74 : i == 2 ? 0 : -74;
        w2 = i == 0 ? 84 : i == 1 ? -29 : i == 2 ? -74 : 55;
        w3 = i == 0 ? 55 : i == 1 ? -84 : i == 2 ? 74 : -29;
    } else {
        w0 = 64;
        w1 = i == 0 ? 83 : i == 1 ? 36 : i == 2 ? -36 : -83;
        w2 = (i == 0 || i == 3) ? 64 : -64;
        w3 = i == 0 ? 36 : i == 1 ? -83 : i == 2 ? 83 : -36;
    }
    auto clip16 = [](int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; };
    int m[4];
#pragma unroll
    for (int col = 0; col < 4; col++) m[col] = clip16((w0 * c[0][col] + w1 * c[1][col] + w2 * c[2][col] + w3 * c[3][col] + 64) >> 7);
    const int shift2 = 20 - bit_depth, add = 1 << (shift2 - 1);
    if constexpr (DST) {
        res[0] = 29 * m[0] + 74 * m[1] + 84 * m[2] + 55 * m[3] + add;
        res[1] = 55 * m[0] + 74 * m[1] - 29 * m[2] - 84 * m[3] + add;
        res[2] = 74 * (m[0] - m[2] + m[3]) + add;
        res[3] = 84 * m[0] - 74 * m[1] + 55 * m[2] - 29 * m[3] + add;
    } else {
        const int e0 = 64 * (m[0] + m[2]) + add, e1 = 64 * (m[0] - m[2]) + add;
        const int o0 = 83 * m[1] + 36 * m[3], o1 = 36 * m[1] - 83 * m[3];
        res[0] = e0 + o0; res[1] = e1 + o1; res[2] = e1 - o1; res[3] = e0 - o0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) res[k] >>= shift2;
}

// row i of an 8x8 / 16x16 / 32x32 inverse DCT: steps A-C of tu_idct_add_body on the block's wave-private LDS tile, with the 16-byte
// coefficient chunks of this lane (chunk q * N + i) already in registers
template <int LOG2N, bool ASM>
__device__ __forceinline__ void idct_row(unsigned char *blk, const int i, const u32x4 *cq, const int bit_depth, int *t)
{
    using L = TuLayout<LOG2N>;
    constexpr int N = L::N, RS = L::RS;
    constexpr PairTab<N> pt{};
#pragma unroll
    for (int q = 0; q < N / 8; q++) {
        const int c = q * N + i;
        *reinterpret_cast<u32x4 *>(blk + (c / (N / 8)) * RS + (c % (N / 8)) * 16) = cq[q];
    }
    __builtin_amdgcn_wave_barrier();
    unsigned p[N / 2];
    {
        const unsigned short *col = reinterpret_cast<const unsigned short *>(blk) + i;
#pragma unroll
        for (int m = 0; m < N / 2; m++)
            p[m] = (unsigned)col[pt.lo[m] * (RS / 2)] | ((unsigned)col[pt.hi[m] * (RS / 2)] << 16);
    }
    __builtin_amdgcn_wave_barrier();
    Idct1D<N, ASM>::run(p, t, 64);
    {
        unsigned short *dst = reinterpret_cast<unsigned short *>(blk) + slot_of_rt(N, i);
#pragma unroll
        for (int r = 0; r < N; r += 2) {
            const unsigned pk = sat_pack_i16(t[r] >> 7, t[r + 1] >> 7);
            dst[r * (RS / 2)]       = (unsigned short)(pk & 0xffffu);
            dst[(r + 1) * (RS / 2)] = (unsigned short)(pk >> 16);
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const u32x4 *rowp = reinterpret_cast<const u32x4 *>(blk + i * RS);
#pragma unroll
        for (int q = 0; q < N / 8; q++) {
            const u32x4 v = rowp[q];
            p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
        }
    }
    const int shift2 = 20 - bit_depth;
    Idct1D<N, ASM>::run(p, t, 1 << (shift2 - 1));
#pragma unroll
    for (int k = 0; k < N; k++) t[k] >>= shift2;
}

// The work of lane (g, i) comes in three stages, so that the chain kernel below can have the records and coefficients of LATER levels in
// flight while a level computes: (R) the two 16-byte records of the block, (C) the coefficient chunks the residual's transform needs
// (addresses from the residual record), (X) samples -> prediction -> residual -> store.
struct PackRecs { u32x4 jw, rw; bool valid; bool has_res = true; };      // has_res false: rw is not a residual record (the chain's branch-free loader)

template <int LOG2N>
__device__ __forceinline__ PackRecs pack_load_recs(const int lane, const int job0, const int njobs, const ohevc_intra_job *__restrict__ jobs,
                                                   const ohevc_tu_job *__restrict__ residuals)
{
    constexpr int N = 1 << LOG2N;
    const int g = lane / N;
    PackRecs r;
    r.valid = job0 + g < njobs;
    const int ji = r.valid ? job0 + g : njobs - 1;          // lanes behind the last job repeat it and do not store
    r.jw = reinterpret_cast<const u32x4 *>(jobs)[ji];
    r.rw = u32x4{ 0u, 0u, 0u, 0u };
    if (residuals != nullptr) r.rw = reinterpret_cast<const u32x4 *>(residuals)[ji];
    return r;
}

__device__ __forceinline__ int pack_kind(const PackRecs &r) { return r.has_res ? (int)((r.rw.y >> 8) & 0xff) - 1 : -1; }      // -1: no residual

// READY: the arena holds the block's RESIDUAL (row-major int16, written in place over the coefficients by intra_chain_residual_kernel): the
// lane takes row i of it instead of its coefficient chunks
template <int LOG2N, bool READY = false>
__device__ __forceinline__ void pack_load_coeffs(const PackRecs &r, const int lane, const int16_t *__restrict__ coeffs, u32x4 (&cq)[4])
{
    constexpr int N = 1 << LOG2N, NCQ = LOG2N == 2 ? 2 : N / 8;
    const int i = lane % N, kind = pack_kind(r);
    const bool is_idct = kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + (is_idct ? r.rw.z : 0u));
    if constexpr (READY) {
        constexpr int NR = LOG2N == 2 ? 1 : N / 8;             // 16-byte pieces of a row of N int16 (a 4x4 row is half a piece)
#pragma unroll
        for (int q = 0; q < NR; q++) {
            cq[q] = u32x4{ 0u, 0u, 0u, 0u };
            if (is_idct) cq[q] = src[LOG2N == 2 ? i / 2 : i * NR + q];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < NCQ; q++) {
        cq[q] = u32x4{ 0u, 0u, 0u, 0u };
        if (is_idct) cq[q] = src[LOG2N == 2 ? q : q * N + i];
    }
}

// The five neighbour samples of lane (g, i) - top[i], top[N + i], left[i], left[N + i], the corner - as loaded; bit k of `none` set: sample k
// has no source (no neighbour group is available) and takes 1 << (bit_depth - 1).  Loading them is a stage of its own so that the chain
// kernel can issue the loads of a level FIRST behind the level's barrier and its prefetches for later levels behind them (vector memory
// returns in order: a prefetch from HBM in front of them would delay every level by an HBM round trip).
struct PackSamples { int v[5]; int none; unsigned char *blk; int stride; };      // blk / stride: the block's first sample and its plane's pitch, worked out once

template <int LOG2N, typename Pixel>
__device__ __forceinline__ PackSamples pack_load_samples(const int lane, const PlaneSet planes, const PackRecs &recs)
{
    constexpr int N = 1 << LOG2N;
    const int i = lane % N;
    const u32x4 jw = recs.jw;
    const int jx = jw.x & 0xffff, jy = jw.x >> 16, jplane = jw.y & 0xff, flags = jw.y >> 24;
    const int bl_size = jw.z & 0xff, tr_size = (jw.z >> 8) & 0xff;
    const int stride = PLANE_STRIDE3(planes, jplane);
    const unsigned char *blk = PLANE_PTR3(planes, jplane) + (__umul24((unsigned)jy, (unsigned)stride) + (unsigned)jx * (unsigned)sizeof(Pixel));      // (a plane is < 4 GiB: 32-bit offsets, 24-bit multiplies)
    const bool c_bl = flags & OHEVC_INTRA_BOTTOM_LEFT, c_l = flags & OHEVC_INTRA_LEFT, c_ul = flags & OHEVC_INTRA_UP_LEFT;
    const bool c_u = flags & OHEVC_INTRA_UP, c_ur = flags & OHEVC_INTRA_UP_RIGHT;
    // Positions relative to the block's first sample, as byte offsets.
    // Substitutes (:251-286): the last sample of a group in scan order is left[N] / left[0] / corner / top[N-1], the first left[N-1] / top[0] / top[N]
    constexpr int P = (int)sizeof(Pixel), NONE = -0x40000000;
    const int o_c = -stride - P, o_l0 = -P, o_ln = N * stride - P, o_lm = (N - 1) * stride - P, o_t0 = -stride, o_tn = N * P - stride, o_tm = (N - 1) * P - stride;
    const int a = c_u ? o_t0 : c_ur ? o_tn : NONE;           // first available group after the corner: its first sample
    const int b = c_l ? o_l0 : c_bl ? o_ln : NONE;           // nearest available group before the corner: its last sample
    const int ul_or_after = c_ul ? o_c : a;
    const int p_bl = c_l ? o_lm : ul_or_after;               // nothing lies before below-left: the first available group after it
    const int p_l = c_bl ? o_ln : ul_or_after;
    const int p_ul = b != NONE ? b : a;
    const int p_u = c_ul ? o_c : b != NONE ? b : (c_ur ? o_tn : NONE);
    const int p_ur = c_u ? o_tm : c_ul ? o_c : b;            // nothing lies after above-right
    auto REC = [&](const int off) -> int { return (int)*reinterpret_cast<const Pixel *>(blk + (off == NONE ? 0 : off)); };
    const int kt = i < tr_size ? i : tr_size - 1, kb = i < bl_size ? i : bl_size - 1;        // beyond the picture: the last valid sample (:111-114, 164-183)
    const int q_t0 = c_u ? i * P - stride : p_u, q_t1 = c_ur ? (N + kt) * P - stride : p_ur;
    const int q_l0 = c_l ? i * stride - P : p_l, q_l1 = c_bl ? (N + kb) * stride - P : p_bl;
    const int q_c = c_ul ? o_c : p_ul;
    PackSamples sm;
    sm.blk = const_cast<unsigned char *>(blk); sm.stride = stride;
    sm.v[0] = REC(q_t0); sm.v[1] = REC(q_t1); sm.v[2] = REC(q_l0); sm.v[3] = REC(q_l1); sm.v[4] = REC(q_c);
    sm.none = (q_t0 == NONE ? 1 : 0) | (q_t1 == NONE ? 2 : 0) | (q_l0 == NONE ? 4 : 0) | (q_l1 == NONE ? 8 : 0) | (q_c == NONE ? 16 : 0);
    return sm;
}

template <int LOG2N, typename Pixel, bool READY = false>
__device__ __forceinline__ void pack_finish(int *ish, unsigned char *tu_lds, const int lane, const PlaneSet planes, const PackRecs &recs, const PackSamples &sm,
                                            const u32x4 (&cq)[4], const int16_t *__restrict__ coeffs, const int bit_depth)
{
    using IL = IntraPackLayout<LOG2N>;
    constexpr int N = IL::N;
    const int g = lane / N, i = lane % N;
    const bool valid = recs.valid;
    const u32x4 jw = recs.jw, rw = recs.rw;
    const int jx = jw.x & 0xffff, jy = jw.x >> 16, jplane = jw.y & 0xff, mode = (jw.y >> 16) & 0xff, flags = jw.y >> 24;
    const int kind = pack_kind(recs);
    const int stride = sm.stride;
    unsigned char *blk = sm.blk;                             // (pack_load_samples worked them out: a level is one wavefront's instruction stream)
    (void)jx; (void)jy; (void)jplane;
    const bool is_idct = kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4;
    const int dflt = 1 << (bit_depth - 1);
    const int v_t0 = (sm.none & 1) ? dflt : sm.v[0], v_t1 = (sm.none & 2) ? dflt : sm.v[1], v_l0 = (sm.none & 4) ? dflt : sm.v[2];
    const int v_l1 = (sm.none & 8) ? dflt : sm.v[3], v_c = (sm.none & 16) ? dflt : sm.v[4];

    int *top = ish + g * IL::INTS + 1, *left = top + IL::ARR, *ftop = left + IL::ARR, *fleft = ftop + IL::ARR, *ref = fleft + IL::ARR - 1 + N;
    top[i] = v_t0; top[N + i] = v_t1; left[i] = v_l0; left[N + i] = v_l1;
    if (i == 0) { top[-1] = v_c; left[-1] = v_c; }
    PACK_SYNC();

    // ---- reference smoothing (:289-327)
    const int *t = top, *l = left;
    if (LOG2N != 2 && !(flags & OHEVC_INTRA_NO_SMOOTHING) && mode != 1) {
        const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
        const int dist = dv < dh ? dv : dh, thresh = LOG2N == 3 ? 7 : LOG2N == 4 ? 1 : 0;
        if (dist > thresh) {
            bool strong = false;
            if constexpr (LOG2N == 5) {
                const int lim = 1 << (bit_depth - 5);
                int a = top[-1] + top[63] - 2 * top[31], b = left[-1] + left[63] - 2 * left[31];
                a = a < 0 ? -a : a; b = b < 0 ? -b : b;
                strong = (flags & OHEVC_INTRA_STRONG) && a < lim && b < lim;
            }
            if (strong) {
                const int t0 = top[-1], t63 = top[2 * N - 1], l0 = left[-1], l63 = left[2 * N - 1];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = h * N + i;
                    ftop[k]  = k < 2 * N - 1 ? (__mul24(2 * N - 1 - k, t0) + __mul24(k + 1, t63) + N) >> (LOG2N + 1) : t63;
                    fleft[k] = k < 2 * N - 1 ? (__mul24(2 * N - 1 - k, l0) + __mul24(k + 1, l63) + N) >> (LOG2N + 1) : l63;
                }
                if (i == 0) { ftop[-1] = t0; fleft[-1] = l0; }
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = h * N + i;
                    ftop[k]  = k < 2 * N - 1 ? (top[k + 1] + 2 * top[k] + top[k - 1] + 2) >> 2 : top[k];
                    fleft[k] = k < 2 * N - 1 ? (left[k + 1] + 2 * left[k] + left[k - 1] + 2) >> 2 : left[k];
                }
                if (i == 0) ftop[-1] = fleft[-1] = (left[0] + 2 * left[-1] + top[0] + 2) >> 2;
            }
            t = ftop; l = fleft;
            PACK_SYNC();
        }
    }

    // ---- prediction of row i
    int pred[N];
    const int maxv = (1 << bit_depth) - 1;
    const bool luma_edge = (flags & OHEVC_INTRA_LUMA_EDGE) && N < 32;
    if (mode == 0) {                                       // pred_planar, :359-372
        // (N-1-x) ly + (x+1) tn is linear in x: one add per sample; (N-1-i) t[x] a 24-bit multiply-add (v_mad_i32_i24, full rate - the
        // 32-bit v_mul_lo_u32 the plain expressions compile to runs at a quarter of it, and a level of the chain lasts as long as its
        // slowest wavefront's instruction stream)
        const int ly = l[i], tn = t[N], ln = l[N];
        const int step = tn - ly, wy = N - 1 - i;
        int acc = __mul24(N - 1, ly) + tn + __mul24(i + 1, ln) + N;
#pragma unroll
        for (int x = 0; x < N; x++) { pred[x] = (acc + __mul24(wy, t[x])) >> (LOG2N + 1); acc += step; }
    } else if (mode == 1) {                                // pred_dc, :388-417
        int part = l[i] + t[i];
#pragma unroll
        for (int o = N / 2; o >= 1; o >>= 1) part += __shfl_xor(part, o);
        const int dc = (part + N) >> (LOG2N + 1);
#pragma unroll
        for (int x = 0; x < N; x++) pred[x] = dc;
        if (luma_edge) {
            if (i == 0) {
#pragma unroll
                for (int x = 1; x < N; x++) pred[x] = (t[x] + 3 * dc + 2) >> 2;
                pred[0] = (l[0] + 2 * dc + t[0] + 2) >> 2;
            } else {
                pred[0] = (l[i] + 3 * dc + 2) >> 2;
            }
        }
    } else {                                               // pred_angular, :419-510
        const int angle = intra_pred_angle(mode), last = (N * angle) >> 5;
        const bool vertical = mode >= 18;
        const int *mainr = vertical ? t : l, *sider = vertical ? l : t;
        ref[i] = mainr[i - 1];
        ref[N + i] = mainr[N + i - 1];
        if (i == 0) { ref[2 * N] = mainr[2 * N - 1]; ref[2 * N + 1] = 0; }
        if (angle < 0 && last < -1) {
            const int inv = intra_inv_angle(mode);
            const int k = -1 - i;                           // k = -1 .. last (last >= -N)
            if (k >= last) ref[k] = sider[-1 + ((k * inv + 128) >> 8)];
        }
        PACK_SYNC();
        if (vertical) {
            const int pos = (i + 1) * angle, i2 = pos >> 5, fact = pos & 31;
            int r0 = ref[i2 + 1];
#pragma unroll
            for (int x = 0; x < N; x++) {
                const int r1 = ref[x + i2 + 2];
                pred[x] = (__mul24(32 - fact, r0) + __mul24(fact, r1) + 16) >> 5;
                r0 = r1;
            }
            if (luma_edge && mode == 26) { const int v = t[0] + ((l[i] - l[-1]) >> 1); pred[0] = v < 0 ? 0 : v > maxv ? maxv : v; }
        } else {
#pragma unroll
            for (int x = 0; x < N; x++) {
                const int pos = (x + 1) * angle, i2 = pos >> 5, fact = pos & 31;
                pred[x] = (__mul24(32 - fact, ref[i + i2 + 1]) + __mul24(fact, ref[i + i2 + 2]) + 16) >> 5;
            }
            if (luma_edge && mode == 10 && i == 0) {
#pragma unroll
                for (int x = 0; x < N; x++) { const int v = l[0] + ((t[x] - t[-1]) >> 1); pred[x] = v < 0 ? 0 : v > maxv ? maxv : v; }
            }
        }
    }

    unsigned char *row = blk + __umul24((unsigned)i, (unsigned)stride);
    if constexpr (READY) {
        // the chain's common case: the residual row is in `cq` as packed clip_int16 pairs (the pre-pass transformed in place) - added pairwise
        if (kind >= 0 && is_idct) {
            unsigned res2[N / 2];
            if constexpr (LOG2N == 2) {
                res2[0] = (i & 1) ? cq[0].z : cq[0].x; res2[1] = (i & 1) ? cq[0].w : cq[0].y;
            } else {
#pragma unroll
                for (int q = 0; q < N / 8; q++) { res2[4 * q] = cq[q].x; res2[4 * q + 1] = cq[q].y; res2[4 * q + 2] = cq[q].z; res2[4 * q + 3] = cq[q].w; }
            }
            finish_row_pairs<N, Pixel>(row, pred, res2, bit_depth, valid);
            return;
        }
    }
    // ---- the row as it will lie in memory
    constexpr int ROWDW = N * (int)sizeof(Pixel) / 4;
    unsigned px[ROWDW];
#pragma unroll
    for (int d = 0; d < ROWDW; d++) {
        if constexpr (sizeof(Pixel) == 1) px[d] = (unsigned)pred[4 * d] | ((unsigned)pred[4 * d + 1] << 8) | ((unsigned)pred[4 * d + 2] << 16) | ((unsigned)pred[4 * d + 3] << 24);
        else px[d] = (unsigned)pred[2 * d] | ((unsigned)pred[2 * d + 1] << 16);
    }

    // ---- the block's residual, row i (hevc_cabac.c:1868-1949), added in registers (transform_add, hevcdsp_template.c:45-111)
    if (kind >= 0) {
        int res[N];
        if (is_idct && READY) {                            // row i of the residual as the pre-pass left it in the arena
            if constexpr (LOG2N == 2) {
                const unsigned lo = (i & 1) ? cq[0].z : cq[0].x, hi = (i & 1) ? cq[0].w : cq[0].y;
                res[0] = (int)(short)(lo & 0xffffu); res[1] = (int)lo >> 16; res[2] = (int)(short)(hi & 0xffffu); res[3] = (int)hi >> 16;
            } else {
#pragma unroll
                for (int q = 0; q < N / 8; q++) {
                    const unsigned w4[4] = { cq[q].x, cq[q].y, cq[q].z, cq[q].w };
#pragma unroll
                    for (int k = 0; k < 4; k++) { res[8 * q + 2 * k] = (int)(short)(w4[k] & 0xffffu); res[8 * q + 2 * k + 1] = (int)w4[k] >> 16; }
                }
            }
        } else if (is_idct) {
            if constexpr (LOG2N == 2) {
                if (kind == OHEVC_TU_DST4) tu4_row<true>(cq[0], cq[1], i, bit_depth, res);
                else                       tu4_row<false>(cq[0], cq[1], i, bit_depth, res);
            } else {
                idct_row<LOG2N, LOG2N >= 4>(tu_lds + g * TuLayout<LOG2N>::BLK, i, cq, bit_depth, res);
            }
        } else {
            tu_rows_residual<LOG2N>(rw, i, coeffs, bit_depth, kind, res);
        }
        finish_row<N, Pixel>(row, px, res, bit_depth, valid);
    } else if (valid) {
        store_row<N, Pixel>(row, px);
    }
}

template <int LOG2N, typename Pixel>
__device__ __forceinline__ void pack_compute(int *ish, unsigned char *tu_lds, const int lane, const PlaneSet planes, const PackRecs &recs, const u32x4 (&cq)[4],
                                             const int16_t *__restrict__ coeffs, const int bit_depth)
{
    const PackSamples sm = pack_load_samples<LOG2N, Pixel>(lane, planes, recs);
    pack_finish<LOG2N, Pixel>(ish, tu_lds, lane, planes, recs, sm, cq, coeffs, bit_depth);
}

template <int LOG2N, typename Pixel>
__device__ __forceinline__ void intra_pack_body(int *ish, unsigned char *tu_lds, const int lane, const int job0, const int njobs, const PlaneSet planes,
                                                const ohevc_intra_job *__restrict__ jobs, const ohevc_tu_job *__restrict__ residuals,
                                                const int16_t *__restrict__ coeffs, const int bit_depth)
{
    const PackRecs r = pack_load_recs<LOG2N>(lane, job0, njobs, jobs, residuals);
    u32x4 cq[4];
    pack_load_coeffs<LOG2N>(r, lane, coeffs, cq);
    pack_compute<LOG2N, Pixel>(ish, tu_lds, lane, planes, r, cq, coeffs, bit_depth);
}

#undef PACK_SYNC

// jobs sorted by size: wavefront w serves 64 / N consecutive blocks of the segment it falls into
struct IntraPackSegs {
    int first_wave[5];               // first_wave[s] .. first_wave[s + 1]: the wavefronts of the (4 << s)-sample blocks
    int first_job[4], njobs[4];
};

// WITH_TU = false: prediction only (no residual records): no transform tile in LDS - 3.4 KB instead of 8.5 KB per wavefront, a third more
// wavefronts per CU
template <typename Pixel, bool WITH_TU>
__global__ __launch_bounds__(64) void intra_pack_kernel(PlaneSet planes, const ohevc_intra_job *__restrict__ jobs, const ohevc_tu_job *__restrict__ residuals,
                                                        IntraPackSegs segs, int bit_depth, const int16_t *__restrict__ coeffs)
{
    __shared__ int ish[kIntraPackInts];
    __shared__ __attribute__((aligned(16))) unsigned char tu_lds[WITH_TU ? TuLayout<5>::WAVE_BYTES : 16];
    static_assert(TuLayout<4>::WAVE_BYTES <= TuLayout<5>::WAVE_BYTES && TuLayout<3>::WAVE_BYTES <= TuLayout<5>::WAVE_BYTES, "transform tile");
    const int w = blockIdx.x, lane = threadIdx.x;
    const int s = w >= segs.first_wave[3] ? 3 : w >= segs.first_wave[2] ? 2 : w >= segs.first_wave[1] ? 1 : 0;       // wave-uniform
    const int local = w - segs.first_wave[s];
    const ohevc_intra_job *j = jobs + segs.first_job[s];
    const ohevc_tu_job *r = WITH_TU && residuals ? residuals + segs.first_job[s] : nullptr;
    const int n = segs.njobs[s];
    if (s == 0)      intra_pack_body<2, Pixel>(ish, tu_lds, lane, local * 16, n, planes, j, r, coeffs, bit_depth);
    else if (s == 1) intra_pack_body<3, Pixel>(ish, tu_lds, lane, local * 8, n, planes, j, r, coeffs, bit_depth);
    else if (s == 2) intra_pack_body<4, Pixel>(ish, tu_lds, lane, local * 4, n, planes, j, r, coeffs, bit_depth);
    else             intra_pack_body<5, Pixel>(ish, tu_lds, lane, local * 2, n, planes, j, r, coeffs, bit_depth);
}

// ------------------------------------------------------------------ a run of NARROW dependency levels in one launch
// Most dependency levels of a picture are narrow: the intra picture of a 1080p stream chains ~1200 levels of 1 .. 16 wavefronts, the tail
// of every inter picture a dozen of 1 .. 8 (OHEVC_TRACE_LEVELS).  As one launch each they cost a kernel boundary apiece - ~6 us of kernel
// (start, three dependent memory round trips, end-of-kernel write-back) plus 2 - 4 us until the next one starts.  A level that fits ONE
// workgroup (kChainWaves wavefronts) needs no kernel boundary to hand its samples to the next level: all of its wavefronts sit on one CU, so
// "stores complete in L2 (s_waitcnt vmcnt(0)) - workgroup barrier - drop this CU's L1 (buffer_inv sc1)" orders them.  The host cuts the
// chain of levels into runs of consecutive levels of at most kChainWaves wavefronts and launches one workgroup per run.
struct IntraChainLevel {             // 48 bytes; mirrors what ctx.hip stages
    int first_wave[5];               // as IntraPackSegs
    int njobs[4];
    unsigned jobs_off16, res_off16;  // the level's job / residual arrays, offsets from `base` in 16-byte units; res_off16 = 0xffffffff: none
    int reserved;
};

constexpr int kChainWaves = 8;       // wavefronts of the chain kernel's one workgroup; a wider level takes ceil(width / 8) steps of it (4 wavefronts - one per SIMD - measured in round 5: 8 030 clocks per level against 6 390, profiles/r5m_*)
constexpr int kChainMaxLevels_v = 1024;
constexpr bool kChainIdleSkips = false; // true: empty wavefront slots skip their loads behind a scalar branch.  Measured in round 6 (profiles/r6zd_chain_clocks_idle_slots_skip_slower.jsonl): a level 6350 -> 6920 clocks - the skipped path meets the loading one in front of the arithmetic and the wait counts get conservative again (the round-4 lesson); the dummy loads of an idle wavefront were not what an active one waits for
constexpr int kChainMaxSlots = kChainMaxLevels_v * 3;   // descriptors of 16 bytes in the LDS of the level records (48 bytes each)
constexpr int kChainMaxLevels = kChainMaxLevels_v;   // levels per launch: their records (48 bytes each) are copied to LDS at the kernel's start

// Round 4.  (a) A level may be WIDER than the workgroup: its wavefront slots 8, 9, ... are served by wavefronts 0, 1, ... in further passes
// behind the first (no barrier in between: the blocks of a level do not depend on each other), so a run no longer ends at every level of
// 9 .. 17 wavefronts - the intra picture of a 1080p stream was 15 chain launches and 205 single-level launches, 5.5 ms on the device, and
// every other picture of the GOP waits for it.  (b) The order of the memory operations inside a level: vector memory returns IN ORDER, so
// whatever is issued in front of a level's neighbour-sample loads delays them, and whatever is outstanding at the end of the level is
// waited for by the release.  Round 3 issued the prefetches (coefficients of level l + 1, records of level l + 2: HBM round trips, they
// were uploaded by DMA) behind the level's stores, i.e. every level waited a whole HBM latency for them before its barrier.  Now: barrier,
// the level's sample loads (L2 hits: the previous level has just written them), THEN the prefetches, then the arithmetic - by the time the
// rows are stored the prefetches have had the whole level to arrive.
// The residual of an intra-coded block (hevc_cabac.c:1868-1949: idct / idct_4x4_luma of its coefficients) does not depend on any sample: only
// its prediction does.  Inside the chain it was the larger part of a level's instructions - a lane of a 32x32 block runs two 32-point
// transforms, ~2500 instructions on one wavefront, 4 us of issue time - and a level lasts as long as its largest block.  So the transforms
// of ALL levels of a run are taken out of the chain: one launch in front of it, a wavefront per 16 / 8 / 4 / 2 blocks over the whole GPU,
// same lane map and same transform bodies as the packed kernel, writes every block's residual IN PLACE over its coefficients (row-major
// int16, what the reference's in-place idct leaves, hevcdsp_template.c:264-301); the chain then adds row i of it to the prediction
// (transform_add, :45-111).  Blocks with other residual kinds (DC, transform-skip, rdpcm, bypass: a few instructions per row) are left as
// they are.  grid = (wavefront slots, levels); a level wider than gridDim.x takes more than one slot per workgroup.
__global__ __launch_bounds__(64) void intra_chain_residual_kernel(const unsigned char *__restrict__ base, const IntraChainLevel *__restrict__ levels, int nlevels,
                                                                  int bit_depth, int16_t *__restrict__ coeffs)
{
    __shared__ __attribute__((aligned(16))) unsigned char tu_lds[TuLayout<5>::WAVE_BYTES];
    const int lane = threadIdx.x;
    const int l = blockIdx.y;
    if (l >= nlevels) return;
    const IntraChainLevel lv = levels[l];
    if (lv.res_off16 == 0xffffffffu) return;
    for (int w = blockIdx.x; w < lv.first_wave[4]; w += gridDim.x) {
        const int s = w >= lv.first_wave[3] ? 3 : w >= lv.first_wave[2] ? 2 : w >= lv.first_wave[1] ? 1 : 0;
        const int first_job = s == 0 ? 0 : s == 1 ? lv.njobs[0] : s == 2 ? lv.njobs[0] + lv.njobs[1] : lv.njobs[0] + lv.njobs[1] + lv.njobs[2];
        // (selects, not lv.first_wave[s] / lv.njobs[s]: an array indexed at run time lives in scratch memory)
        const int fws = s == 0 ? lv.first_wave[0] : s == 1 ? lv.first_wave[1] : s == 2 ? lv.first_wave[2] : lv.first_wave[3];
        const int job0 = (w - fws) * (16 >> s), n = s == 0 ? lv.njobs[0] : s == 1 ? lv.njobs[1] : s == 2 ? lv.njobs[2] : lv.njobs[3];
        const ohevc_intra_job *j = reinterpret_cast<const ohevc_intra_job *>(base + (size_t)lv.jobs_off16 * 16) + first_job;
        const ohevc_tu_job *r = reinterpret_cast<const ohevc_tu_job *>(base + (size_t)lv.res_off16 * 16) + first_job;
        auto one = [&](auto log2c) {
            constexpr int LOG2N = decltype(log2c)::value, N = 1 << LOG2N;
            const int g = lane / N, i = lane % N;
            const PackRecs rec = pack_load_recs<LOG2N>(lane, job0, n, j, r);
            const int kind = pack_kind(rec);
            const bool is_idct = kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4;
            u32x4 cq[4] = {};
            pack_load_coeffs<LOG2N>(rec, lane, coeffs, cq);
            int res[N];
            if constexpr (LOG2N == 2) {
                if (kind == OHEVC_TU_DST4) tu4_row<true>(cq[0], cq[1], i, bit_depth, res);
                else                       tu4_row<false>(cq[0], cq[1], i, bit_depth, res);
            } else {
                // (every lane takes the transform's wave-level scheduling points, also those of blocks without such a residual)
                idct_row<LOG2N, LOG2N >= 4>(tu_lds + g * TuLayout<LOG2N>::BLK, i, cq, bit_depth, res);
            }
            // in place: no lane stores before every lane of the wavefront has its coefficients (in lockstep anyway; said for the record)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (rec.valid && is_idct) {
                unsigned *dst = reinterpret_cast<unsigned *>(coeffs + rec.rw.z + i * N);
#pragma unroll
                for (int k = 0; k < N / 2; k++) {
                    const int a = res[2 * k] < -32768 ? -32768 : res[2 * k] > 32767 ? 32767 : res[2 * k];
                    const int b = res[2 * k + 1] < -32768 ? -32768 : res[2 * k + 1] > 32767 ? 32767 : res[2 * k + 1];
                    dst[k] = ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if (s == 0)      one(std::integral_constant<int, 2>{});
        else if (s == 1) one(std::integral_constant<int, 3>{});
        else if (s == 2) one(std::integral_constant<int, 4>{});
        else             one(std::integral_constant<int, 5>{});
    }
}

// CLOCKS: the diagnosis build of the kernel (ohevc_debug_intra_chain_clocks).  The counters used to be part of THE kernel behind `if
// (phase_clocks)`: eighteen 64-bit accumulators and a dozen scalar branches a step in a kernel that already spills scalar registers - and a
// wavefront's instruction stream is what a level costs (one instruction per ~4 clocks, whatever its kind: round 6, the listing between two
// time stamps).  The shipped instance carries none of it.
template <typename Pixel, bool CLOCKS>
__global__ __launch_bounds__(64 * kChainWaves) void intra_chain_kernel(PlaneSet planes, const unsigned char *__restrict__ base, const IntraChainLevel *__restrict__ levels,
                                                                       int nlevels, int bit_depth, const int16_t *__restrict__ coeffs, int agent_acquire,
                                                                       unsigned long long *__restrict__ phase_clocks_arg)
{
    unsigned long long *const phase_clocks = CLOCKS ? phase_clocks_arg : nullptr;
    __shared__ int ish[kChainWaves * kIntraPackInts];
    __shared__ __attribute__((aligned(16))) unsigned char tu_lds[kChainWaves * TuLayout<5>::WAVE_BYTES];
    // The level records, all of them, in LDS before the first level starts.  Read from memory as a level needs them (a scalar load whose
    // result the record loads of level l + 2 wait for) they cost every level an HBM round trip on its critical path - the records arrive
    // by DMA, nothing has them in a cache: 4500 of a level's 11 800 shader clocks (ohevc_debug_intra_chain_clocks, profiles/r4g_*).
    __shared__ __attribute__((aligned(16))) int slev[kChainMaxLevels * 12];
    static_assert(sizeof(IntraChainLevel) == 48, "12 dwords per level record");
    // Round 6: "what does wavefront slot w do at level l" took ~100 instructions a step (eleven fields of the level record out of LDS into
    // scalars, selects by size class, two 64-bit addresses) - a third of a step's 2500 issue clocks, for a value the records fix before the
    // first level runs.  The prologue now turns the level records into one 16-byte DESCRIPTOR per (level, wavefront slot) - first job record,
    // first residual record, blocks left, size class - in the LDS the records used to occupy, next to the levels' first-slot indices; a step
    // reads its descriptor and is done.  A run with more slots than fit (kChainMaxSlots) keeps the record form (slot_at below).
    __shared__ unsigned short lvl_off[kChainMaxLevels + 2];
    __shared__ int scan_tot[kChainWaves + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int *my_ish = ish + wave * kIntraPackInts;
    unsigned char *my_tu = tu_lds + wave * TuLayout<5>::WAVE_BYTES;
    static_assert(kChainMaxLevels == 2 * 64 * kChainWaves, "two levels per thread in the prologue's scan");
    bool use_desc;
    {
        const int l0 = 2 * (int)threadIdx.x, l1 = l0 + 1;
        const int nw0 = l0 < nlevels ? levels[l0].first_wave[4] : 0, nw1 = l1 < nlevels ? levels[l1].first_wave[4] : 0;
        int incl = nw0 + nw1;                                          // inclusive scan over the workgroup's threads (two levels each)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        if (lane == 63) scan_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kChainWaves; k++) { const int t = scan_tot[k]; if (k < wave) before += t; total += t; }
        use_desc = __builtin_amdgcn_readfirstlane((int)(total <= kChainMaxSlots && !(agent_acquire & 4))) != 0;      // (bit 2 of the hand-over switch: the record form, for A/B runs on one box)
        if (use_desc) {
            const int off0 = before + incl - nw0 - nw1, off1 = off0 + nw0;
            if (l0 < nlevels) lvl_off[l0] = (unsigned short)off0;
            if (l1 < nlevels) lvl_off[l1] = (unsigned short)off1;
            if (threadIdx.x == 0) lvl_off[nlevels] = (unsigned short)total;      // (the last level's slots end here - also when nlevels is the 1024 a launch takes: no thread owns "level 1024")
            u32x4 *desc = reinterpret_cast<u32x4 *>(slev);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int l = h ? l1 : l0, off = h ? off1 : off0;
                if (l >= nlevels) continue;
                const IntraChainLevel lv = levels[l];
                const int n0 = lv.njobs[0], n1 = lv.njobs[1], n2 = lv.njobs[2], n3 = lv.njobs[3];
                for (int w = 0; w < lv.first_wave[4]; w++) {
                    const int sc = w >= lv.first_wave[3] ? 3 : w >= lv.first_wave[2] ? 2 : w >= lv.first_wave[1] ? 1 : 0;
                    const int fws = sc == 0 ? 0 : sc == 1 ? lv.first_wave[1] : sc == 2 ? lv.first_wave[2] : lv.first_wave[3];
                    const int first_job = sc == 0 ? 0 : sc == 1 ? n0 : sc == 2 ? n0 + n1 : n0 + n1 + n2;
                    const int job0 = (w - fws) * (16 >> sc), n = sc == 0 ? n0 : sc == 1 ? n1 : sc == 2 ? n2 : n3;
                    desc[off + w] = u32x4{ lv.jobs_off16 + (unsigned)(first_job + job0), lv.res_off16 != 0xffffffffu ? lv.res_off16 + (unsigned)(first_job + job0) : 0xffffffffu,
                                           (unsigned)(n - job0), (unsigned)sc };
                }
            }
        } else {
            for (int i = threadIdx.x; i < nlevels * 12; i += 64 * kChainWaves) slev[i] = reinterpret_cast<const int *>(levels)[i];
        }
        __syncthreads();
    }

    // what wavefront slot w does at a level: the size class of its blocks (-1: nothing), its first block, the level's arrays.  All wave-uniform.
    struct Slot { int s, job0, n, nwaves; const ohevc_intra_job *j; const ohevc_tu_job *r; };
    auto slot_at = [&](const int l, const int w) -> Slot {
        Slot sl = { -1, 0, 0, 0, nullptr, nullptr };
        if (l >= nlevels) return sl;
        if (use_desc) {
            const int first = __builtin_amdgcn_readfirstlane((int)lvl_off[l]), nw = __builtin_amdgcn_readfirstlane((int)lvl_off[l + 1]) - first;
            sl.nwaves = nw;
            if (w >= nw) return sl;
            const u32x4 dsc = reinterpret_cast<const u32x4 *>(slev)[first + w];
            const unsigned j16 = (unsigned)__builtin_amdgcn_readfirstlane((int)dsc.x), r16 = (unsigned)__builtin_amdgcn_readfirstlane((int)dsc.y);
            sl.n = __builtin_amdgcn_readfirstlane((int)dsc.z);            // blocks from this slot's first one to the end of the size class
            sl.s = __builtin_amdgcn_readfirstlane((int)dsc.w);
            sl.j = reinterpret_cast<const ohevc_intra_job *>(base + (size_t)j16 * 16);
            sl.r = r16 != 0xffffffffu ? reinterpret_cast<const ohevc_tu_job *>(base + (size_t)r16 * 16) : nullptr;
            return sl;
        }
        // The record as eleven scalars, never as an array indexed by the (runtime) size class: such an array lives in scratch memory, and a
        // scratch load is a vector-memory operation - it returns IN ORDER behind the sample loads and the HBM prefetches issued just before,
        // so "which slot do I have at level l + 2" waited a whole HBM round trip on every level's critical path (s_waitcnt vmcnt behind
        // scratch_load_dword in the round-4 listing; the 3100 .. 4700 clocks of the "issue" phase in profiles/r4t_chain_clocks_mul24.jsonl).
        const int *rec = slev + l * 12;
        const int fw1 = __builtin_amdgcn_readfirstlane(rec[1]), fw2 = __builtin_amdgcn_readfirstlane(rec[2]), fw3 = __builtin_amdgcn_readfirstlane(rec[3]);
        const int fw4 = __builtin_amdgcn_readfirstlane(rec[4]);
        const int n0 = __builtin_amdgcn_readfirstlane(rec[5]), n1 = __builtin_amdgcn_readfirstlane(rec[6]), n2 = __builtin_amdgcn_readfirstlane(rec[7]);
        const int n3 = __builtin_amdgcn_readfirstlane(rec[8]);
        const unsigned jobs_off16 = (unsigned)__builtin_amdgcn_readfirstlane(rec[9]), res_off16 = (unsigned)__builtin_amdgcn_readfirstlane(rec[10]);
        sl.nwaves = fw4;
        if (w >= fw4) return sl;
        const int s = w >= fw3 ? 3 : w >= fw2 ? 2 : w >= fw1 ? 1 : 0;
        const int first_wave = s == 0 ? 0 : s == 1 ? fw1 : s == 2 ? fw2 : fw3;          // (first_wave[0] is 0 by construction)
        const int first_job = s == 0 ? 0 : s == 1 ? n0 : s == 2 ? n0 + n1 : n0 + n1 + n2;
        sl.s = s;
        sl.job0 = (w - first_wave) * (16 >> s);
        sl.n = s == 0 ? n0 : s == 1 ? n1 : s == 2 ? n2 : n3;
        sl.j = reinterpret_cast<const ohevc_intra_job *>(base + (size_t)jobs_off16 * 16) + first_job;
        sl.r = res_off16 != 0xffffffffu ? reinterpret_cast<const ohevc_tu_job *>(base + (size_t)res_off16 * 16) + first_job : nullptr;
        return sl;
    };
    // The two loaders that run AHEAD of a level (records of level l + 2, residual rows of level l + 1) are written without a branch: one
    // instruction stream for all four block sizes, every load unconditional, addresses picked by selects on wave-uniform values.  With a
    // copy of the loader per size class behind `if (sl.s == ...)` the loaded registers met in phi nodes, the compiler merged them with
    // copies, and a copy needs the value: s_waitcnt vmcnt(0) right behind the issue of the prefetch - the level waited for the loads it was
    // supposed to hide (the listing of round 4; "issue" = 4700 of a level's 10 500 clocks in profiles/r4t_chain_clocks_mul24.jsonl).
    // A slot without blocks (s < 0) loads from `base` (the upload buffer: always mapped) and its values are never looked at.
    // (kChainIdleSkips: a wavefront whose slot is EMPTY at a step skips that step's loads - tried in round 6, slower, off)
    auto load_recs = [&](const Slot &sl) -> PackRecs {
        if (kChainIdleSkips && sl.s < 0) { PackRecs r; r.jw = u32x4{ 0u, 0u, 0u, 0u }; r.rw = u32x4{ 0u, 0u, 0u, 0u }; r.valid = false; r.has_res = false; return r; }
        const int log2n = sl.s + 2;                                    // (s < 0: 1 - harmless)
        const int g = lane >> (log2n & 7);
        PackRecs r;
        r.valid = sl.s >= 0 && sl.job0 + g < sl.n;
        const int ji = sl.job0 + g < sl.n ? sl.job0 + g : sl.n - 1;   // lanes behind the last job repeat it and do not store
        const u32x4 *jp = sl.s >= 0 ? reinterpret_cast<const u32x4 *>(sl.j) + ji : reinterpret_cast<const u32x4 *>(base);
        const u32x4 *rp = sl.s >= 0 && sl.r != nullptr ? reinterpret_cast<const u32x4 *>(sl.r) + ji : reinterpret_cast<const u32x4 *>(base);
        r.jw = *jp;
        r.rw = *rp;
        r.has_res = sl.s >= 0 && sl.r != nullptr;                      // (applied where the record is READ, pack_kind: a select here would need the value)
        return r;
    };
    auto load_cq = [&](const Slot &sl, const PackRecs &r, u32x4 (&cq)[4]) {
        if (kChainIdleSkips && sl.s < 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) cq[q] = u32x4{ 0u, 0u, 0u, 0u };
            return;
        }
        // row i of the block's residual as the pre-pass left it in the arena (READY form of pack_load_coeffs): N / 8 pieces of 16 bytes
        // (4x4: the piece that holds rows i and i ^ 1); the remaining of the four loads repeat the last piece
        const int log2n = sl.s + 2, n = 1 << (log2n & 7), i = lane & (n - 1);
        const int kind = pack_kind(r);
        const bool is_idct = sl.s >= 0 && coeffs != nullptr && (kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4);
        const int nr = log2n <= 3 ? 1 : n >> 3;
        const unsigned row_bytes = log2n == 2 ? (unsigned)(i >> 1) * 16u : (unsigned)i * (unsigned)n * 2u;
        const unsigned char *rowp = is_idct ? reinterpret_cast<const unsigned char *>(coeffs) + ((size_t)r.rw.z * 2u + row_bytes) : base;
#pragma unroll
        for (int q = 0; q < 4; q++) cq[q] = *reinterpret_cast<const u32x4 *>(rowp + (is_idct ? 16 * (q < nr ? q : nr - 1) : 0));
    };
    auto load_samples = [&](const Slot &sl, const PackRecs &r) -> PackSamples {
        if (sl.s == 0) return pack_load_samples<2, Pixel>(lane, planes, r);
        if (sl.s == 1) return pack_load_samples<3, Pixel>(lane, planes, r);
        if (sl.s == 2) return pack_load_samples<4, Pixel>(lane, planes, r);
        if (sl.s == 3) return pack_load_samples<5, Pixel>(lane, planes, r);
        return PackSamples{ { 0, 0, 0, 0, 0 }, 0, nullptr, 0 };
    };
    auto finish = [&](const Slot &sl, const PackRecs &r, const PackSamples &sm, const u32x4 (&cq)[4]) {
        if (sl.s == 0)      pack_finish<2, Pixel, true>(my_ish, my_tu, lane, planes, r, sm, cq, coeffs, bit_depth);
        else if (sl.s == 1) pack_finish<3, Pixel, true>(my_ish, my_tu, lane, planes, r, sm, cq, coeffs, bit_depth);
        else if (sl.s == 2) pack_finish<4, Pixel, true>(my_ish, my_tu, lane, planes, r, sm, cq, coeffs, bit_depth);
        else if (sl.s == 3) pack_finish<5, Pixel, true>(my_ish, my_tu, lane, planes, r, sm, cq, coeffs, bit_depth);
    };

    // Software pipeline over STEPS: a step is one pass of the 8-wavefront workgroup over a level - a level of up to 8 wavefront slots is one
    // step, a wider one several (its blocks do not depend on each other: no release / acquire between its passes, only the barrier that keeps a
    // wavefront's LDS arrays from being rewritten while some of its lanes still read them).  Every step issues the SAME loads in the same
    // order - its neighbour samples (L2 hits: the previous level has just written them), then the residual rows of the next step and the
    // records of the step after that (from HBM: they arrived by DMA) - so the wait in front of the arithmetic is exactly "my five samples",
    // whatever was issued behind them.  Round 4 served the further passes of a wide level inside the level's loop body with loads of their own
    // behind `if (pass)`: two paths with different numbers of younger loads meet in front of the arithmetic, the compiler's s_waitcnt takes the
    // smaller count, and on the common path that count covered the next level's residual prefetch - every level waited for the loads it was
    // meant to hide (vmcnt(4) .. vmcnt(0) in front of the first use of a sample in the round-4 listing).
    struct Step { int l, p; };
    auto next_step = [&](const Step st) -> Step {
        const int nw = st.l >= nlevels ? 0 : use_desc ? __builtin_amdgcn_readfirstlane((int)lvl_off[st.l + 1]) - __builtin_amdgcn_readfirstlane((int)lvl_off[st.l])
                                                       : __builtin_amdgcn_readfirstlane(slev[st.l * 12 + 4]);
        return (st.p + 1) * kChainWaves < nw ? Step{ st.l, st.p + 1 } : Step{ st.l + 1, 0 };
    };
    auto slot_of = [&](const Step st) -> Slot { return slot_at(st.l, st.p * kChainWaves + wave); };
    Step st0 = { 0, 0 }, st1 = next_step(st0);
    Slot s0 = slot_of(st0), s1 = slot_of(st1);
    PackRecs r0 = load_recs(s0), r1 = load_recs(s1);
    u32x4 cq0[4], cq1[4];
    load_cq(s0, r0, cq0);
    // phase_clocks != NULL (diagnosis, ohevc_debug_intra_chain_clocks): wavefront 0 adds up, over the steps, the shader clocks it spends
    // [0] waiting for its stores + in the barrier, [1] issuing the step's loads and prefetches, [2] in the step's arithmetic (incl. the
    // wait for the samples), [3] unused; [4] = levels; [5..7] = the issue phase after the sample loads / the residual prefetch / the records
    unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pcx[3] = { 0, 0, 0 };
    unsigned long long cls_fin[4] = { 0, 0, 0, 0 }, cls_wait[4] = { 0, 0, 0, 0 }, cls_n[4] = { 0, 0, 0, 0 };
    while (st0.l < nlevels) {
        const unsigned long long tk0 = phase_clocks ? clock64() : 0;
        if (st0.p == 0) {
            if (st0.l) {
                // The hand-over between two levels is a release / acquire at WORKGROUP scope: every wavefront of the workgroup runs on one CU
                // and shares its vector L1 (no thread-group split), stores go through that L1, so "my stores have completed" + the barrier is
                // all it takes (the gfx942 memory model: workgroup-scope acquire needs no cache invalidate).  Rounds 2 - 3 issued the AGENT-scope
                // acquire here (buffer_inv sc1, what a hand-over between workgroups on different CUs needs): it costs every level microseconds
                // (agent_acquire != 0 keeps that form for A/B runs).
                // Round 6: not even the wait for the level's stores.  On gfx950 outside thread-group-split mode a workgroup-scope release is NO
                // instruction at all (what the compiler emits for fence(release, "workgroup") - s_barrier - fence(acquire, "workgroup") is the
                // barrier alone: stores and loads of one CU reach its L1 / the L2 in issue order) - the s_waitcnt vmcnt(0) of rounds 4-5 made
                // every level wait for the write acknowledgements of its rows (~1000 of a level's 6350 clocks, profiles/r5r_chain_clocks_*)
                // before the next level could even ISSUE its loads (agent_acquire bit 1 keeps that form for A/B runs).
                if (agent_acquire & 2) xcd_release();
                __syncthreads();
                if (agent_acquire & 1) xcd_acquire();
            }
        } else {
            __syncthreads();
        }
        const unsigned long long tk1 = phase_clocks ? clock64() : 0;
        const PackSamples sm = load_samples(s0, r0);         // first: they hit the L2 the previous level wrote
        issue_order_fence();
        if (phase_clocks) pcx[0] += clock64() - tk1;
        load_cq(s1, r1, cq1);                                // behind them: what later steps need, from HBM
        issue_order_fence();
        if (phase_clocks) pcx[1] += clock64() - tk1;
        const Step st2 = next_step(st1);
        const Slot s2 = slot_of(st2);
        const PackRecs r2 = load_recs(s2);
        issue_order_fence();
        const unsigned long long tk2 = phase_clocks ? clock64() : 0;
        if (phase_clocks) pcx[2] += tk2 - tk1;
        unsigned long long tkw = tk2;
        if (phase_clocks) { wait_all_but_6_loads(); tkw = clock64(); }      // (the five samples: six younger loads are in flight behind them)
        finish(s0, r0, sm, cq0);
        if (phase_clocks) {
            const unsigned long long tk3 = clock64();
            pc0 += tk1 - tk0; pc1 += tk2 - tk1; pc2 += tk3 - tk2;
            if (s0.s >= 0) { cls_fin[s0.s] += tk3 - tk2; cls_wait[s0.s] += tkw - tk2; cls_n[s0.s]++; }
            if (threadIdx.x == 0 && st0.p == 0) atomicAdd(&phase_clocks[20 + (s0.nwaves < 16 ? s0.nwaves : 16)], 1ull);
        }
        st0 = st1; st1 = st2;
        s0 = s1; r0 = r1;
#pragma unroll
        for (int q = 0; q < 4; q++) cq0[q] = cq1[q];
        s1 = s2; r1 = r2;
    }
    const unsigned long long pc3 = 0;
    if (phase_clocks && threadIdx.x == 0) {
        atomicAdd(&phase_clocks[0], pc0); atomicAdd(&phase_clocks[1], pc1); atomicAdd(&phase_clocks[2], pc2); atomicAdd(&phase_clocks[3], pc3);
        atomicAdd(&phase_clocks[4], (unsigned long long)nlevels);
        atomicAdd(&phase_clocks[5], pcx[0]); atomicAdd(&phase_clocks[6], pcx[1]); atomicAdd(&phase_clocks[7], pcx[2]);
    }
    if (phase_clocks && lane == 0)
        for (int s = 0; s < 4; s++) { atomicAdd(&phase_clocks[8 + s], cls_fin[s]); atomicAdd(&phase_clocks[12 + s], cls_wait[s]); atomicAdd(&phase_clocks[16 + s], cls_n[s]); }
}

}  // namespace ohevc
