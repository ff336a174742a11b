/*
 * hevc_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
 *
 * Plain-C restatement of the algorithms behind openHEVC's HEVCDSPContext / HEVCPredContext tables
 * (libavcodec/hevcdsp_template.c, libavcodec/hevcpred_template.c), written from their integer
 * semantics with one generic, run-time bit-depth code path (the reference instantiates a template
 * per BIT_DEPTH).  Each function cites the reference lines it restates.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every function here bit-for-bit
 * against the reference's own compiled C (oracle/_ref/libhevcref.so, built by oracle/Makefile from
 * /root/reference) on seeded random and corner-case inputs; tests/test_table_driver_cpu.py pins it at
 * whole-picture level against the reference's tables driven by a miniature front-end; and tests/golden/
 * holds 1652 output digests frozen from that reference build (tests/golden/make_golden.py, replayed by
 * tests/test_oracle_golden.py) so the pin also holds where /root/reference is absent.  The reference
 * ships no golden vectors of its own.  
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define OHX(name) ohor_##name
#include "oracle_api.h"

int ohor_available(void) { return 1; }

/* ------------------------------------------------------------------ helpers */
static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int clip_i16(int v) { return clip3(v, -32768, 32767); }              /* libavutil/common.h:139 */
static inline int clip_px(int v, int bd) { return clip3(v, 0, (1 << bd) - 1); }    /* bit_depth_template.c:67,86 */
static inline int psz(int bd) { return bd > 8 ? 2 : 1; }

static inline int ldpx(const uint8_t *p, int bd)
{
    return bd > 8 ? *(const uint16_t *)p : *p;
}
static inline void stpx(uint8_t *p, int bd, int v)
{
    if (bd > 8) *(uint16_t *)p = (uint16_t)v; else *p = (uint8_t)v;
}
/* pixel (x,y) relative to p, stride in bytes */
#define PX(p, stride, x, y) ((p) + (ptrdiff_t)(y) * (stride) + (ptrdiff_t)(x) * ps)

/* ------------------------------------------------------------------ transform matrix
 * The HEVC core transform is an integer approximation of 64*sqrt(2)*cos(k*pi/64); all 32x32 entries are
 * +-(one of 32 magnitudes) selected by the angle index (2c+1)*r mod 128.  This generates the same table the
 * reference spells out literally (libavcodec/hevcdsp.c:879-944). */
static const int8_t kCosMag[33] = {
    64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0
};
static int dct_coef(int r, int c)    /* row r (frequency), column c (sample), 32-point */
{
    int m = (r * (2 * c + 1)) & 127;
    if (m <= 32) return  kCosMag[m];
    if (m <= 64) return -kCosMag[64 - m];
    if (m <= 96) return -kCosMag[m - 64];
    return kCosMag[128 - m];
}

/* Which input index j does an N-point partial butterfly with bound `end` actually read?
 * TR_32/TR_16/TR_8/TR_4 (hevcdsp_template.c:210-262): the odd half of the outermost stage reads odd j < end;
 * TR_32 forwards end/2 to its inner TR_16; every deeper stage is called with its full length. */
static int tr_reads(int n, int j, int end)
{
    if (n == 4) return 1;
    if (j & 1) return j < end;
    if (n == 32 && (j & 3) == 2) return (j >> 1) < end / 2;
    return 1;
}

static void inv_dct_1d(int n, const int *src, int *dst, int end)
{
    int step = 32 / n;
    for (int k = 0; k < n; k++) {
        int acc = 0;
        for (int j = 0; j < n; j++)
            if (tr_reads(n, j, end))
                acc += dct_coef(j * step, k) * src[j];
        dst[k] = acc;
    }
}

/* idct_NxN: hevcdsp_template.c:264-301 */
static void idct_full(int bd, int log2, int16_t *c, int col_limit)
{
    int n = 1 << log2, in[32], out[32];
    int limit  = col_limit < n ? col_limit : n;
    int limit2 = col_limit + 4 < n ? col_limit + 4 : n;
    int shift = 7, add = 1 << (shift - 1);
    for (int i = 0; i < n; i++) {                   /* pass 1: columns, in place */
        for (int j = 0; j < n; j++) in[j] = c[j * n + i];
        inv_dct_1d(n, in, out, limit2);
        for (int j = 0; j < n; j++) c[j * n + i] = (int16_t)clip_i16((out[j] + add) >> shift);
        if (limit2 < n && i % 4 == 0 && i != 0)
            limit2 -= 4;
    }
    shift = 20 - bd; add = 1 << (shift - 1);
    for (int i = 0; i < n; i++) {                   /* pass 2: rows */
        for (int j = 0; j < n; j++) in[j] = c[i * n + j];
        inv_dct_1d(n, in, out, limit);
        for (int j = 0; j < n; j++) c[i * n + j] = (int16_t)clip_i16((out[j] + add) >> shift);
    }
}

/* idct_NxN_dc: hevcdsp_template.c:303-316 */
static void idct_dc(int bd, int log2, int16_t *c)
{
    /* BIT_DEPTH 14: the reference's `1 << (shift - 1)` (hevcdsp_template.c:308) has a negative count; its gcc build folds it to 0
     * (pinned against oracle/_ref) */
    int n2 = 1 << (2 * log2), shift = 14 - bd, add = shift > 0 ? 1 << (shift - 1) : 0;
    int v = (((c[0] + 1) >> 1) + add) >> shift;
    for (int i = 0; i < n2; i++) c[i] = (int16_t)v;
}

/* transform_4x4_luma (inverse DST-VII): hevcdsp_template.c:170-203 */
static void dst4_1d(const int *s, int *d)
{
    /* rows of the 4x4 DST matrix {29,55,74,84},{74,74,0,-74},{84,-29,-74,55},{55,-84,74,-29} applied transposed */
    d[0] = 29 * s[0] + 74 * s[1] + 84 * s[2] + 55 * s[3];
    d[1] = 55 * s[0] + 74 * s[1] - 29 * s[2] - 84 * s[3];
    d[2] = 74 * s[0]             - 74 * s[2] + 74 * s[3];
    d[3] = 84 * s[0] - 74 * s[1] + 55 * s[2] - 29 * s[3];
}
static void idct_dst4(int bd, int16_t *c)
{
    int in[4], out[4], shift = 7, add = 64;
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) in[j] = c[j * 4 + i];
        dst4_1d(in, out);
        for (int j = 0; j < 4; j++) c[j * 4 + i] = (int16_t)clip_i16((out[j] + add) >> shift);
    }
    shift = 20 - bd; add = 1 << (shift - 1);
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) in[j] = c[i * 4 + j];
        dst4_1d(in, out);
        for (int j = 0; j < 4; j++) c[i * 4 + j] = (int16_t)clip_i16((out[j] + add) >> shift);
    }
}

/* transform_skip: hevcdsp_template.c:139-163 (note: results wrap to int16 like the in-place reference) */
static void tr_skip(int bd, int log2, int16_t *c)
{
    int n2 = 1 << (2 * log2), shift = 15 - bd - log2;
    if (shift > 0) {
        int off = 1 << (shift - 1);
        for (int i = 0; i < n2; i++) c[i] = (int16_t)((c[i] + off) >> shift);
    } else {
        for (int i = 0; i < n2; i++) c[i] = (int16_t)(c[i] << -shift);
    }
}

/* transform_rdpcm: hevcdsp_template.c:114-136 (mode 0: running sum along rows, 1: along columns) */
static void tr_rdpcm(int log2, int16_t *c, int mode)
{
    int n = 1 << log2;
    if (mode) {
        for (int y = 1; y < n; y++)
            for (int x = 0; x < n; x++) c[y * n + x] = (int16_t)(c[y * n + x] + c[(y - 1) * n + x]);
    } else {
        for (int y = 0; y < n; y++)
            for (int x = 1; x < n; x++) c[y * n + x] = (int16_t)(c[y * n + x] + c[y * n + x - 1]);
    }
}

void ohor_tu_residual(int bd, int kind, int log2, int16_t *coeffs, int col_limit)
{
    switch (kind) {
    case OH_TU_IDCT: idct_full(bd, log2, coeffs, col_limit); break;
    case OH_TU_DC:   idct_dc(bd, log2, coeffs); break;
    case OH_TU_DST4: idct_dst4(bd, coeffs); break;
    case OH_TU_SKIP: tr_skip(bd, log2, coeffs); break;
    case OH_TU_SKIP_RDPCM_H: tr_skip(bd, log2, coeffs); tr_rdpcm(log2, coeffs, 0); break;
    case OH_TU_SKIP_RDPCM_V: tr_skip(bd, log2, coeffs); tr_rdpcm(log2, coeffs, 1); break;
    case OH_TU_BYPASS: break;
    case OH_TU_BYPASS_RDPCM_H: tr_rdpcm(log2, coeffs, 0); break;
    case OH_TU_BYPASS_RDPCM_V: tr_rdpcm(log2, coeffs, 1); break;
    }
}

/* transform_add: hevcdsp_template.c:45-111 */
void ohor_transform_add(int bd, int log2, uint8_t *dst, ptrdiff_t stride, int16_t *res)
{
    int n = 1 << log2, ps = psz(bd);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            uint8_t *p = PX(dst, stride, x, y);
            stpx(p, bd, clip_px(ldpx(p, bd) + res[y * n + x], bd));
        }
}

void ohor_tu_batch(int bd, int kind, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                   ptrdiff_t stride, const int32_t *xy, int col_limit)
{
    int nn = 1 << (2 * log2), ps = psz(bd);
    int16_t tmp[32 * 32];
    for (int i = 0; i < n; i++) {
        memcpy(tmp, coeffs + (size_t)i * nn, nn * sizeof(int16_t));
        ohor_tu_residual(bd, kind, log2, tmp, col_limit);
        ohor_transform_add(bd, log2, PX(plane, stride, xy[2 * i], xy[2 * i + 1]), stride, tmp);
    }
}

struct tu_mt_arg { int bd, kind, log2, n, col_limit; const int16_t *coeffs; uint8_t *plane; ptrdiff_t stride; const int32_t *xy; };
static void *tu_mt_worker(void *p)
{
    struct tu_mt_arg *a = p;
    ohor_tu_batch(a->bd, a->kind, a->log2, a->n, a->coeffs, a->plane, a->stride, a->xy, a->col_limit);
    return NULL;
}
void ohor_tu_batch_mt(int bd, int kind, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                      ptrdiff_t stride, const int32_t *xy, int col_limit, int threads)
{
    pthread_t th[256];
    struct tu_mt_arg a[256];
    int nn = 1 << (2 * log2);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    for (int t = 0; t < threads; t++) {
        int lo = (int)((long long)n * t / threads), hi = (int)((long long)n * (t + 1) / threads);
        a[t] = (struct tu_mt_arg){ bd, kind, log2, hi - lo, col_limit, coeffs + (size_t)lo * nn, plane, stride, xy + 2 * lo };
        pthread_create(&th[t], NULL, tu_mt_worker, &a[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------ motion compensation
 * Interpolation taps: the HEVC luma quarter-sample (8-tap) and chroma eighth-sample (4-tap) filters,
 * reference tables ff_hevc_qpel_filters / ff_hevc_epel_filters (libavcodec/hevcdsp.c:1028-1042). */
static const int8_t kLumaTaps[3][8] = {
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};
static const int8_t kChromaTaps[7][4] = {
    { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 },
    { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 },
};

/* 14-bit intermediate sample of put_hevc_{qpel,epel}_{pixels,h,v,hv}
 * (hevcdsp_template.c:610-624,731-794,1185-1247): src points at the block's (0,0) integer sample. */
static int mc_sample14(int bd, int luma, const uint8_t *src, ptrdiff_t stride, int x, int y, int mx, int my)
{
    int ps = psz(bd), taps = luma ? 8 : 4, before = luma ? 3 : 1;
    const int8_t *fh = mx ? (luma ? kLumaTaps[mx - 1] : kChromaTaps[mx - 1]) : NULL;
    const int8_t *fv = my ? (luma ? kLumaTaps[my - 1] : kChromaTaps[my - 1]) : NULL;
    if (!fh && !fv)
        return ldpx(PX(src, stride, x, y), bd) << (14 - bd);
    if (fh && !fv) {
        int s = 0;
        for (int k = 0; k < taps; k++) s += fh[k] * ldpx(PX(src, stride, x + k - before, y), bd);
        return s >> (bd - 8);
    }
    if (!fh && fv) {
        int s = 0;
        for (int k = 0; k < taps; k++) s += fv[k] * ldpx(PX(src, stride, x, y + k - before), bd);
        return s >> (bd - 8);
    }
    int acc = 0;
    for (int r = 0; r < taps; r++) {
        int s = 0;
        for (int k = 0; k < taps; k++) s += fh[k] * ldpx(PX(src, stride, x + k - before, y + r - before), bd);
        acc += fv[r] * (int16_t)(s >> (bd - 8));          /* the h-pass lands in an int16 tmp[] (:763-776) */
    }
    return acc >> 6;
}

void ohor_mc(int bd, int luma, int variant, uint8_t *dst, ptrdiff_t dststride,
             uint8_t *src, ptrdiff_t srcstride, int16_t *src2, ptrdiff_t src2stride,
             int height, int mx, int my, int width,
             int denom, int wx0, int wx1, int ox0, int ox1)
{
    int ps = psz(bd);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int v = mc_sample14(bd, luma, src, srcstride, x, y, mx, my);
            if (variant == OH_MC_PUT) {
                ((int16_t *)dst)[y * dststride + x] = (int16_t)v;
                continue;
            }
            int out;
            switch (variant) {
            case OH_MC_UNI: {           /* :626-640: the full-sample case is a memcpy -- a sample above the bit depth's range (the
                                         * constrained-intra 0x8080 samples above 8 bit) is carried over unclipped; :796-820,...: the
                                         * interpolating cases clip ((v + off) >> shift) */
                int shift = 14 - bd, off = bd < 14 ? 1 << (shift - 1) : 0;
                out = (mx || my) ? clip_px((v + off) >> shift, bd) : ldpx(PX(src, srcstride, x, y), bd);
                break;
            }
            case OH_MC_BI: {            /* :642-666,822-848,... */
                int shift = 15 - bd, off = bd < 14 ? 1 << (shift - 1) : 0;
                out = clip_px((v + src2[y * src2stride + x] + off) >> shift, bd);
                break;
            }
            case OH_MC_UNI_W: {         /* :668-690,985-1010,... */
                int shift = denom + 14 - bd, off = bd < 14 ? 1 << (shift - 1) : 0;
                out = clip_px(((v * wx0 + off) >> shift) + ox0 * (1 << (bd - 8)), bd);
                break;
            }
            default: {                  /* OH_MC_BI_W :692-716,1012-1038,... */
                int log2wd = denom + 14 - bd;
                int o0 = ox0 * (1 << (bd - 8)), o1 = ox1 * (1 << (bd - 8));
                out = clip_px((v * wx1 + src2[y * src2stride + x] * wx0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1), bd);
                break;
            }
            }
            stpx(PX(dst, dststride, x, y), bd, out);
        }
}

/* ------------------------------------------------------------------ deblocking
 * hevc_loop_filter_luma: hevcdsp_template.c:1629-1723.  `xs` steps ACROSS the edge, `ys` ALONG it (bytes). */
static void deblock_luma(int bd, uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int beta,
                         const int *tc_in, const uint8_t *no_p_in, const uint8_t *no_q_in)
{
#define S(i, line) ldpx(pix + (i) * xs + (line) * ys, bd)          /* i = -4..3 : p3..p0,q0..q3 */
#define W(i, line, v) stpx(pix + (i) * xs + (line) * ys, bd, (v))
    beta <<= bd - 8;
    for (int seg = 0; seg < 2; seg++, pix += 4 * ys) {
        int dp0 = abs(S(-3, 0) - 2 * S(-2, 0) + S(-1, 0)), dq0 = abs(S(2, 0) - 2 * S(1, 0) + S(0, 0));
        int dp3 = abs(S(-3, 3) - 2 * S(-2, 3) + S(-1, 3)), dq3 = abs(S(2, 3) - 2 * S(1, 3) + S(0, 3));
        int d0 = dp0 + dq0, d3 = dp3 + dq3;
        int tc = tc_in[seg] << (bd - 8), no_p = no_p_in[seg], no_q = no_q_in[seg];
        if (d0 + d3 >= beta)
            continue;
        int tc25 = (tc * 5 + 1) >> 1;
        int strong = abs(S(-4, 0) - S(-1, 0)) + abs(S(3, 0) - S(0, 0)) < (beta >> 3) && abs(S(-1, 0) - S(0, 0)) < tc25 &&
                     abs(S(-4, 3) - S(-1, 3)) + abs(S(3, 3) - S(0, 3)) < (beta >> 3) && abs(S(-1, 3) - S(0, 3)) < tc25 &&
                     (d0 << 1) < (beta >> 2) && (d3 << 1) < (beta >> 2);
        if (strong) {
            int tc2 = tc << 1;
            for (int l = 0; l < 4; l++) {
                int p3 = S(-4, l), p2 = S(-3, l), p1 = S(-2, l), p0 = S(-1, l);
                int q0 = S(0, l), q1 = S(1, l), q2 = S(2, l), q3 = S(3, l);
                if (!no_p) {
                    W(-1, l, p0 + clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc2, tc2));
                    W(-2, l, p1 + clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2));
                    W(-3, l, p2 + clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc2, tc2));
                }
                if (!no_q) {
                    W(0, l, q0 + clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc2, tc2));
                    W(1, l, q1 + clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2));
                    W(2, l, q2 + clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc2, tc2));
                }
            }
        } else {
            int side = (beta + (beta >> 1)) >> 3, tc_2 = tc >> 1;
            int nd_p = dp0 + dp3 < side ? 2 : 1, nd_q = dq0 + dq3 < side ? 2 : 1;
            for (int l = 0; l < 4; l++) {
                int p2 = S(-3, l), p1 = S(-2, l), p0 = S(-1, l), q0 = S(0, l), q1 = S(1, l), q2 = S(2, l);
                int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                if (abs(delta) >= 10 * tc)
                    continue;
                delta = clip3(delta, -tc, tc);
                if (!no_p) W(-1, l, clip_px(p0 + delta, bd));
                if (!no_q) W(0, l, clip_px(q0 - delta, bd));
                if (!no_p && nd_p > 1)
                    W(-2, l, clip_px(p1 + clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2), bd));
                if (!no_q && nd_q > 1)
                    W(1, l, clip_px(q1 + clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2), bd));
            }
        }
    }
}

/* hevc_loop_filter_chroma: hevcdsp_template.c:1725-1757 */
static void deblock_chroma(int bd, uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys,
                           const int *tc_in, const uint8_t *no_p_in, const uint8_t *no_q_in)
{
    for (int seg = 0; seg < 2; seg++, pix += 4 * ys) {
        int tc = tc_in[seg] << (bd - 8);
        if (tc <= 0)
            continue;
        for (int l = 0; l < 4; l++) {
            int p1 = S(-2, l), p0 = S(-1, l), q0 = S(0, l), q1 = S(1, l);
            int delta = clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
            if (!no_p_in[seg]) W(-1, l, clip_px(p0 + delta, bd));
            if (!no_q_in[seg]) W(0, l, clip_px(q0 - delta, bd));
        }
    }
#undef S
#undef W
}

/* wrappers hevc_{h,v}_loop_filter_*: hevcdsp_template.c:1759-1787.  "v" = vertical edge: across = 1 pixel */
void ohor_deblock_luma(int bd, int vertical_edge, uint8_t *pix, ptrdiff_t stride, int beta,
                       int *tc, uint8_t *no_p, uint8_t *no_q)
{
    if (vertical_edge) deblock_luma(bd, pix, psz(bd), stride, beta, tc, no_p, no_q);
    else               deblock_luma(bd, pix, stride, psz(bd), beta, tc, no_p, no_q);
}
void ohor_deblock_chroma(int bd, int vertical_edge, uint8_t *pix, ptrdiff_t stride,
                         int *tc, uint8_t *no_p, uint8_t *no_q)
{
    if (vertical_edge) deblock_chroma(bd, pix, psz(bd), stride, tc, no_p, no_q);
    else               deblock_chroma(bd, pix, stride, psz(bd), tc, no_p, no_q);
}

/* ------------------------------------------------------------------ SAO
 * sao_band_filter_0: hevcdsp_template.c:340-365 */
/* samples above the bit depth's range that went through the band filter since the last reset: each of them makes the REFERENCE read
 * past its offset table, i.e. its output for the stream is not defined (tools/fuzz_streams.py asks before it compares) */
static long ohor_band_above_range;
long ohor_sao_band_above_range(int reset)
{
    long v = __atomic_load_n(&ohor_band_above_range, __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(&ohor_band_above_range, 0, __ATOMIC_RELAXED);
    return v;
}

void ohor_sao_band(int bd, uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src,
                   const int16_t *offset_val, int band_position, int width, int height)
{
    int table[32] = { 0 }, ps = psz(bd), shift = bd - 5;
    for (int k = 0; k < 4; k++)
        table[(k + band_position) & 31] = offset_val[k + 1];
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int v = ldpx(PX(src, stride_src, x, y), bd);
            /* & 31: a sample above the bit depth's range (constrained intra prediction above 8 bit leaves 0x8080 samples,
             * hevcpred_template.c:117-141) indexes past the reference's 32-entry table -- whatever lies on ITS stack.  Here, and in
             * the kernel, the band index wraps instead; such streams cannot be compared with the reference */
            if (v >> bd) __atomic_fetch_add(&ohor_band_above_range, 1, __ATOMIC_RELAXED);
            stpx(PX(dst, stride_dst, x, y), bd, clip_px(v + table[(v >> shift) & 31], bd));
        }
}

/* sao_edge_filter_{0,1}: hevcdsp_template.c:372-567 */
void ohor_sao_edge(int bd, int restore, uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst,
                   ptrdiff_t stride_src, const int16_t *offset_val, int eo_class, int *borders,
                   int width, int height, uint8_t *vert_edge, uint8_t *horiz_edge, uint8_t *diag_edge)
{
    static const int8_t nb[4][2][2] = {        /* neighbour (dx,dy) pairs per eo_class (:378-383) */
        { { -1, 0 }, { 1, 0 } }, { { 0, -1 }, { 0, 1 } }, { { -1, -1 }, { 1, 1 } }, { { 1, -1 }, { -1, 1 } },
    };
    static const uint8_t edge_idx[5] = { 1, 2, 0, 3, 4 };
    int ps = psz(bd);
#define COPYPX(x, y, off) stpx(PX(dst, stride_dst, x, y), bd, clip_px(ldpx(PX(src, stride_src, x, y), bd) + (off), bd))
#define RESTOREPX(x, y)   stpx(PX(dst, stride_dst, x, y), bd, ldpx(PX(src, stride_src, x, y), bd))
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            int c = ldpx(PX(src, stride_src, x, y), bd);
            int a = ldpx(PX(src, stride_src, x + nb[eo_class][0][0], y + nb[eo_class][0][1]), bd);
            int b = ldpx(PX(src, stride_src, x + nb[eo_class][1][0], y + nb[eo_class][1][1]), bd);
            int s0 = (c > a) - (c < a), s1 = (c > b) - (c < b);
            stpx(PX(dst, stride_dst, x, y), bd, clip_px(c + offset_val[edge_idx[2 + s0 + s1]], bd));
        }
    /* picture-border rows/columns get offset_val[0] (:419-455); the reference's second column loop really
     * runs over `height` (it is a column), kept as is */
    int init_x = 0, init_y = 0, w = width, h = height;
    if (eo_class != 1) {
        if (borders[0]) { for (int y = 0; y < h; y++) COPYPX(0, y, offset_val[0]); init_x = 1; }
        if (borders[2]) { for (int y = 0; y < h; y++) COPYPX(w - 1, y, offset_val[0]); w--; }
    }
    if (eo_class != 0) {
        if (borders[1]) { for (int x = init_x; x < w; x++) COPYPX(x, 0, offset_val[0]); }
        if (borders[3]) { for (int x = init_x; x < w; x++) COPYPX(x, h - 1, offset_val[0]); h--; }
    }
    if (!restore)
        return;
    /* variant 1 (:533-566): undo the filter across slice/tile edges that may not be crossed */
    int sul = !diag_edge[0] && eo_class == 2 && !borders[0] && !borders[1];
    int sur = !diag_edge[1] && eo_class == 3 && !borders[1] && !borders[2];
    int slr = !diag_edge[2] && eo_class == 2 && !borders[2] && !borders[3];
    int sll = !diag_edge[3] && eo_class == 3 && !borders[0] && !borders[3];
    if (vert_edge[0] && eo_class != 1)
        for (int y = init_y + sul; y < h - sll; y++) RESTOREPX(0, y);
    if (vert_edge[1] && eo_class != 1)
        for (int y = init_y + sur; y < h - slr; y++) RESTOREPX(w - 1, y);
    if (horiz_edge[0] && eo_class != 0)
        for (int x = init_x + sul; x < w - sur; x++) RESTOREPX(x, 0);
    if (horiz_edge[1] && eo_class != 0)
        for (int x = init_x + sll; x < w - slr; x++) RESTOREPX(x, h - 1);
    if (diag_edge[0] && eo_class == 2) RESTOREPX(0, 0);
    if (diag_edge[1] && eo_class == 3) RESTOREPX(w - 1, 0);
    if (diag_edge[2] && eo_class == 2) RESTOREPX(w - 1, h - 1);
    if (diag_edge[3] && eo_class == 3) RESTOREPX(0, h - 1);
#undef COPYPX
#undef RESTOREPX
}

/* ------------------------------------------------------------------ intra predictors
 * All three take integer neighbour arrays t[-1..2N-1], l[-1..2N-1] and write an N x N block. */
static void predict_planar(int bd, int log2, uint8_t *dst, ptrdiff_t stride, const int *t, const int *l)
{   /* hevcpred_template.c:359-372 */
    int n = 1 << log2, ps = psz(bd);
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            stpx(PX(dst, stride, x, y), bd,
                 ((n - 1 - x) * l[y] + (x + 1) * t[n] + (n - 1 - y) * t[x] + (y + 1) * l[n] + n) >> (log2 + 1));
}

static void predict_dc(int bd, int log2, uint8_t *dst, ptrdiff_t stride, const int *t, const int *l, int c_idx)
{   /* hevcpred_template.c:388-417 */
    int n = 1 << log2, ps = psz(bd), dc = n;
    for (int i = 0; i < n; i++) dc += l[i] + t[i];
    dc >>= log2 + 1;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) stpx(PX(dst, stride, x, y), bd, dc);
    if (c_idx == 0 && n < 32) {
        stpx(PX(dst, stride, 0, 0), bd, (l[0] + 2 * dc + t[0] + 2) >> 2);
        for (int x = 1; x < n; x++) stpx(PX(dst, stride, x, 0), bd, (t[x] + 3 * dc + 2) >> 2);
        for (int y = 1; y < n; y++) stpx(PX(dst, stride, 0, y), bd, (l[y] + 3 * dc + 2) >> 2);
    }
}

static void predict_angular(int bd, int log2, uint8_t *dst, ptrdiff_t stride, const int *t, const int *l,
                            int c_idx, int mode)
{   /* hevcpred_template.c:419-510 */
    static const int8_t angle_tab[33] = {
        32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
        -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    static const int16_t inv_angle_tab[15] = {
        -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };
    int n = 1 << log2, ps = psz(bd), angle = angle_tab[mode - 2], last = (n * angle) >> 5;
    int vertical = mode >= 18;
    const int *main_ = vertical ? t : l, *side = vertical ? l : t;
    int refbuf[3 * 32 + 4], *ref = refbuf + 32;     /* ref[k] == main_[k-1] for k >= 0 */
    /* the reference copies main_[-1..n-1] rounded up to a multiple of four samples when it projects (:444-449);
     * without projection it reads main_ directly, up to index 2n-1 */
    for (int k = 0; k <= 2 * n; k++) ref[k] = main_[k - 1];
    if (angle < 0 && last < -1)
        for (int k = last; k <= -1; k++)
            ref[k] = side[-1 + ((k * inv_angle_tab[mode - 11] + 128) >> 8)];
    for (int a = 0; a < n; a++) {                   /* a runs along the prediction direction's minor axis */
        int idx = ((a + 1) * angle) >> 5, fact = ((a + 1) * angle) & 31;
        for (int b = 0; b < n; b++) {
            int v = fact ? ((32 - fact) * ref[b + idx + 1] + fact * ref[b + idx + 2] + 16) >> 5 : ref[b + idx + 1];
            if (vertical) stpx(PX(dst, stride, b, a), bd, v);
            else          stpx(PX(dst, stride, a, b), bd, v);
        }
    }
    if (c_idx == 0 && n < 32) {                     /* edge smoothing of the pure vertical / horizontal modes */
        if (mode == 26)
            for (int y = 0; y < n; y++) stpx(PX(dst, stride, 0, y), bd, clip_px(t[0] + ((l[y] - l[-1]) >> 1), bd));
        if (mode == 10)
            for (int x = 0; x < n; x++) stpx(PX(dst, stride, x, 0), bd, clip_px(l[0] + ((t[x] - t[-1]) >> 1), bd));
    }
}

static void load_nb(int bd, int n2, const uint8_t *p, int *out)   /* p -> element 0; copies [-1 .. n2-1] */
{
    int ps = psz(bd);
    for (int i = -1; i < n2; i++) out[i] = ldpx(p + (ptrdiff_t)i * ps, bd);
}

void ohor_pred_planar(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride)
{
    int tb[66], lb[66];
    load_nb(bd, 2 << log2, top, tb + 1); load_nb(bd, 2 << log2, left, lb + 1);
    predict_planar(bd, log2, src, stride, tb + 1, lb + 1);
}
void ohor_pred_dc(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int c_idx)
{
    int tb[66], lb[66];
    load_nb(bd, 2 << log2, top, tb + 1); load_nb(bd, 2 << log2, left, lb + 1);
    predict_dc(bd, log2, src, stride, tb + 1, lb + 1, c_idx);
}
void ohor_pred_angular(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left,
                       ptrdiff_t stride, int c_idx, int mode)
{
    int tb[66], lb[66];
    load_nb(bd, 2 << log2, top, tb + 1); load_nb(bd, 2 << log2, left, lb + 1);
    predict_angular(bd, log2, src, stride, tb + 1, lb + 1, c_idx, mode);
}

/* ------------------------------------------------------------------ intra_pred(): neighbour preparation
 * hevcpred_template.c:30-357.  z-scan order inside a CTB, -1 outside (hevc_ps.c:2551-2567). */
static int zscan_addr(int x, int y, int bits)
{
    if (x < 0 || y < 0) return -1;
    int v = 0;
    for (int i = 0; i < bits; i++)
        v |= ((x >> i) & 1) << (2 * i) | ((y >> i) & 1) << (2 * i + 1);
    return v;
}

/* What the constrained-intra substitution walk needs to know about the block's surroundings: per-sample "is the neighbour
 * intra-coded" bits (bit k+1 of top: IS_INTRA(k, -1), of left: IS_INTRA(-1, k), k = -1..63), the two scan limits and whether
 * the block touches the left / top picture border. */
typedef struct cip_view {
    int on;
    uint8_t top[9], left[9];
    int smx, smy, x0_nonzero, y0_nonzero;
} cip_view;

/* The part of intra_pred() (hevcpred_template.c:164-357) that no longer looks at the decoder context: neighbour arrays from the
 * FINAL availability flags, constrained-intra substitution, padding of unavailable runs, smoothing, prediction.
 * luma_edge = (c_idx == 0): DC / mode 10 / mode 26 boundary smoothing; no_smoothing = intra_smoothing_disabled_flag || chroma
 * outside 4:4:4; strong = sps_strong_intra_smoothing_enable_flag && luma. */
static void intra_core(int bd, uint8_t *blk, ptrdiff_t stride, int log2, int mode, int cand_bottom_left, int cand_left, int cand_up_left,
                       int cand_up, int cand_up_right, int bl_size, int tr_size, int no_smoothing, int strong, int luma_edge,
                       const cip_view *cv)
{
    const int ps = psz(bd), n = 1 << log2, c_idx = luma_edge ? 0 : 1;
    int lbuf[2 * 32 + 8], tbuf[2 * 32 + 8], flbuf[2 * 32 + 8], ftbuf[2 * 32 + 8];
    int *l = lbuf + 4, *t = tbuf + 4, *fl = flbuf + 4, *ft = ftbuf + 4;
#define REC(x, y) ldpx(PX(blk, stride, x, y), bd)
#define ISTOP(k)  ((cv->top[((k) + 1) >> 3] >> (((k) + 1) & 7)) & 1)
#define ISLEFT(k) ((cv->left[((k) + 1) >> 3] >> (((k) + 1) & 7)) & 1)
    if (cv->on) {
        /* memset(left/top, 128, 2*MAX_TB_SIZE*sizeof(pixel)); top[-1] = 128  (:160-162): BYTES of value 128 */
        const int fill = bd > 8 ? 0x8080 : 128;
        for (int i = 0; i < 64; i++) l[i] = t[i] = fill;
        t[-1] = 128; l[-1] = 128;                               /* left[-1] is uninitialised in the reference here */
    }
    if (cand_up_left) { l[-1] = REC(-1, -1); t[-1] = l[-1]; }
    if (cand_up) for (int i = 0; i < n; i++) t[i] = REC(i, -1);
    if (cand_up_right) {
        for (int i = n; i < n + tr_size; i++) t[i] = REC(i, -1);
        for (int i = n + tr_size; i < 2 * n; i++) t[i] = REC(n + tr_size - 1, -1);
    }
    if (cand_left) for (int i = 0; i < n; i++) l[i] = REC(-1, i);
    if (cand_bottom_left) {
        for (int i = n; i < n + bl_size; i++) l[i] = REC(-1, i);
        for (int i = n + bl_size; i < 2 * n; i++) l[i] = REC(-1, n + bl_size - 1);
    }
    /* ---- constrained intra prediction, part 2 (:185-249): samples of inter-coded neighbours are overwritten from the
     * nearest intra-coded ones, in groups of four, in this exact order */
    if (cv->on && (cand_bottom_left || cand_left || cand_up_left || cand_up || cand_up_right)) {
        const int smx = cv->smx, smy = cv->smy;
        int j = n + (cand_bottom_left ? bl_size : 0) - 1;
        if (cand_bottom_left || cand_left || cand_up_left) {
            while (j > -1 && !ISLEFT(j)) j--;
            if (!ISLEFT(j)) {
                j = 0;
                while (j < smx && !ISTOP(j)) j++;
                for (int i = j; i > j - (j + 1); i--) if (!ISTOP(i - 1)) t[i - 1] = t[i];      /* EXTEND_LEFT_CIP */
                l[-1] = t[-1];
            }
        } else {
            j = 0;
            while (j < smx && !ISTOP(j)) j++;
            if (j > 0) {
                if (cv->x0_nonzero) {
                    for (int i = j; i > j - (j + 1); i--) if (!ISTOP(i - 1)) t[i - 1] = t[i];
                } else {
                    for (int i = j; i > j - j; i--) if (!ISTOP(i - 1)) t[i - 1] = t[i];
                    t[-1] = t[0];
                }
            }
            l[-1] = t[-1];
        }
        l[-1] = t[-1];
        if (cand_bottom_left || cand_left) {                     /* EXTEND_DOWN_CIP(left, 0, size_max_y) */
            int a = l[-1];
            for (int i = 0; i < smy; i += 4) {
                if (!ISLEFT(i)) { l[i] = l[i + 1] = l[i + 2] = l[i + 3] = a; } else a = l[i + 3];
            }
        }
        if (!cand_left) for (int i = 0; i < n; i++) l[i] = l[-1];
        if (!cand_bottom_left) for (int i = n; i < 2 * n; i++) l[i] = l[n - 1];
        if (cv->x0_nonzero && cv->y0_nonzero) {
            int a = l[smy - 1];
            for (int i = smy - 1; i > smy - 1 - smy; i -= 4) {   /* EXTEND_UP_CIP(left, size_max_y - 1, size_max_y) */
                if (!ISLEFT(i - 3)) { l[i - 3] = l[i - 2] = l[i - 1] = l[i] = a; } else a = l[i - 3];
            }
            if (!ISLEFT(-1)) l[-1] = l[0];
        } else if (!cv->x0_nonzero) {
            for (int i = 0; i < smy; i++) l[i] = 0;              /* EXTEND(left, 0, size_max_y) */
        } else {
            int a = l[smy - 1];
            for (int i = smy - 1; i > smy - 1 - smy; i -= 4) {
                if (!ISLEFT(i - 3)) { l[i - 3] = l[i - 2] = l[i - 1] = l[i] = a; } else a = l[i - 3];
            }
        }
        t[-1] = l[-1];
        if (cv->y0_nonzero) {                                    /* EXTEND_RIGHT_CIP(top, 0, size_max_x) */
            int a = l[-1];
            for (int i = 0; i < smx; i += 4) {
                if (!ISTOP(i)) { t[i] = t[i + 1] = t[i + 2] = t[i + 3] = a; } else a = t[i + 3];
            }
        }
    }
#undef ISTOP
#undef ISLEFT
    /* substitution of unavailable samples (:251-286) */
    if (!cand_bottom_left) {
        if (cand_left) {
            for (int i = n; i < 2 * n; i++) l[i] = l[n - 1];
        } else if (cand_up_left) {
            for (int i = 0; i < 2 * n; i++) l[i] = l[-1];
            cand_left = 1;
        } else if (cand_up) {
            l[-1] = t[0];
            for (int i = 0; i < 2 * n; i++) l[i] = l[-1];
            cand_up_left = cand_left = 1;
        } else if (cand_up_right) {
            for (int i = 0; i < n; i++) t[i] = t[n];
            l[-1] = t[n];
            for (int i = 0; i < 2 * n; i++) l[i] = l[-1];
            cand_up = cand_up_left = cand_left = 1;
        } else {
            l[-1] = 1 << (bd - 1);
            for (int i = 0; i < 2 * n; i++) t[i] = l[i] = l[-1];
        }
    }
    if (!cand_left) for (int i = 0; i < n; i++) l[i] = l[n];
    if (!cand_up_left) l[-1] = l[0];
    if (!cand_up) for (int i = 0; i < n; i++) t[i] = l[-1];
    if (!cand_up_right) for (int i = n; i < 2 * n; i++) t[i] = t[n - 1];
    t[-1] = l[-1];

    /* reference-sample smoothing (:289-327) */
    if (!no_smoothing && mode != 1 && n != 4) {
        static const int thresh[3] = { 7, 1, 0 };
        int dv = abs(mode - 26), dh = abs(mode - 10), dist = dv < dh ? dv : dh;
        if (dist > thresh[log2 - 3]) {
            int lim = 1 << (bd - 5);
            if (strong && log2 == 5 &&
                abs(t[-1] + t[63] - 2 * t[31]) < lim && abs(l[-1] + l[63] - 2 * l[31]) < lim) {
                ft[-1] = t[-1]; ft[63] = t[63]; fl[-1] = l[-1]; fl[63] = l[63];
                for (int i = 0; i < 63; i++) {
                    ft[i] = ((63 - i) * t[-1] + (i + 1) * t[63] + 32) >> 6;
                    fl[i] = ((63 - i) * l[-1] + (i + 1) * l[63] + 32) >> 6;
                }
            } else {
                fl[2 * n - 1] = l[2 * n - 1]; ft[2 * n - 1] = t[2 * n - 1];
                for (int i = 2 * n - 2; i >= 0; i--) {
                    fl[i] = (l[i + 1] + 2 * l[i] + l[i - 1] + 2) >> 2;
                    ft[i] = (t[i + 1] + 2 * t[i] + t[i - 1] + 2) >> 2;
                }
                ft[-1] = fl[-1] = (l[0] + 2 * l[-1] + t[0] + 2) >> 2;
            }
            l = fl; t = ft;
        }
    }
    if (mode == 0)      predict_planar(bd, log2, blk, stride, t, l);
    else if (mode == 1) predict_dc(bd, log2, blk, stride, t, l, c_idx);
    else                predict_angular(bd, log2, blk, stride, t, l, c_idx, mode);
#undef REC
}

void ohor_intra_pred(int bd, const oh_intra_pic *pic, int x0, int y0, int log2, int c_idx, int mode,
                     int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right)
{
    int ps = psz(bd), cfi = pic->chroma_format_idc;
    int hs = c_idx ? (cfi == 1 || cfi == 2) : 0, vs = c_idx ? (cfi == 1) : 0;
    int n = 1 << log2;
    int nlh = n << hs, nlv = n << vs;                         /* block size in luma samples */
    int tb_bits = pic->log2_ctb_size - pic->log2_min_tb_size, tb_mask = (1 << tb_bits) - 1;
    int x_tb = (x0 >> pic->log2_min_tb_size) & tb_mask, y_tb = (y0 >> pic->log2_min_tb_size) & tb_mask;
    int cur = zscan_addr(x_tb, y_tb, tb_bits);
    ptrdiff_t stride = pic->linesize[c_idx];
    uint8_t *blk = pic->data[c_idx] + (ptrdiff_t)(y0 >> vs) * stride + (ptrdiff_t)(x0 >> hs) * ps;
    cip_view cv;
    memset(&cv, 0, sizeof(cv));

    /* z-scan qualification of the two "ahead" candidates (:105-109) */
    cand_bottom_left = cand_bottom_left &&
        cur > zscan_addr(x_tb - 1, (y_tb + (nlv >> pic->log2_min_tb_size)) & tb_mask, tb_bits);
    cand_up_right = cand_up_right &&
        cur > zscan_addr((x_tb + (nlh >> pic->log2_min_tb_size)) & tb_mask, y_tb - 1, tb_bits);

    int y_end = y0 + 2 * nlv < pic->height ? y0 + 2 * nlv : pic->height;
    int x_end = x0 + 2 * nlh < pic->width  ? x0 + 2 * nlh : pic->width;
    int bl_size = (y_end - (y0 + nlv)) >> vs, tr_size = (x_end - (x0 + nlh)) >> hs;   /* :111-114 */

    /* ---- constrained intra prediction, part 1 (:116-163): neighbours coded as inter count as unavailable.
     * IS(px, py): is the minimum PU at LUMA position (px, py) intra?  (MVF_PU / IS_INTRA, :33-40).  Positions outside
     * the picture read undefined memory in the reference; here they count as "not intra". */
    const int cip = pic->constrained_intra_pred;
    const int lpu = pic->log2_min_pu_size;
    const int pu_w = (pic->width + (1 << lpu) - 1) >> lpu, pu_h = (pic->height + (1 << lpu) - 1) >> lpu;
#define ISPU(xp, yp) ((xp) >= 0 && (yp) >= 0 && (xp) < pu_w && (yp) < pu_h && pic->is_intra[(xp) + (yp) * pu_w])
#define ISI(x, y) ISPU((x0 + (int)((unsigned)(x) << hs)) >> lpu, (y0 + (int)((unsigned)(y) << vs)) >> lpu)
    if (cip) {
        int spu_v = nlv >> lpu, spu_h = nlh >> lpu;
        const int edge_x = !(x0 & ((1 << lpu) - 1)), edge_y = !(y0 & ((1 << lpu) - 1));
        if (!spu_h) spu_h++;                                  /* (only the horizontal count is bumped, :122-123) */
        if (cand_bottom_left == 1 && edge_x) {
            const int xl = (x0 - 1) >> lpu, yb = (y0 + nlv) >> lpu;
            const int max = spu_v < pu_h - yb ? spu_v : pu_h - yb;
            cand_bottom_left = 0;
            for (int i = 0; i < max; i += 2) cand_bottom_left |= ISPU(xl, yb + i);
        }
        if (cand_left == 1 && edge_x) {
            const int xl = (x0 - 1) >> lpu, yl = y0 >> lpu;
            const int max = spu_v < pu_h - yl ? spu_v : pu_h - yl;
            cand_left = 0;
            for (int i = 0; i < max; i += 2) cand_left |= ISPU(xl, yl + i);
        }
        if (cand_up_left == 1) cand_up_left = ISPU((x0 - 1) >> lpu, (y0 - 1) >> lpu);
        if (cand_up == 1 && edge_y) {
            const int xt = x0 >> lpu, yt = (y0 - 1) >> lpu;
            const int max = spu_h < pu_w - xt ? spu_h : pu_w - xt;
            cand_up = 0;
            for (int i = 0; i < max; i += 2) cand_up |= ISPU(xt + i, yt);
        }
        if (cand_up_right == 1 && edge_y) {
            const int yt = (y0 - 1) >> lpu, xr = (x0 + nlh) >> lpu;
            const int max = spu_h < pu_w - xr ? spu_h : pu_w - xr;
            cand_up_right = 0;
            for (int i = 0; i < max; i += 2) cand_up_right |= ISPU(xr + i, yt);
        }
        /* what part 2 (in intra_core) reads of the surroundings: the per-sample intra bits and the scan limits, :187-198 */
        cv.on = 1;
        for (int k = -1; k < 64; k++) {
            if (ISI(k, -1)) cv.top[(k + 1) >> 3] |= (uint8_t)(1u << ((k + 1) & 7));
            if (ISI(-1, k)) cv.left[(k + 1) >> 3] |= (uint8_t)(1u << ((k + 1) & 7));
        }
        cv.smx = x0 + ((2 * n) << hs) < pic->width ? 2 * n : (pic->width - x0) >> hs;
        cv.smy = y0 + ((2 * n) << vs) < pic->height ? 2 * n : (pic->height - y0) >> vs;
        if (!cand_up_right) cv.smx = x0 + (n << hs) < pic->width ? n : (pic->width - x0) >> hs;
        if (!cand_bottom_left) cv.smy = y0 + (n << vs) < pic->height ? n : (pic->height - y0) >> vs;
        cv.x0_nonzero = x0 != 0; cv.y0_nonzero = y0 != 0;
    }
#undef ISI
#undef ISPU
    intra_core(bd, blk, stride, log2, mode, cand_bottom_left, cand_left, cand_up_left, cand_up, cand_up_right, bl_size, tr_size,
               pic->intra_smoothing_disabled || !(c_idx == 0 || cfi == 3), pic->strong_intra_smoothing && c_idx == 0, c_idx == 0, &cv);
}

/* The same prediction from a product job record (ohevc_intra_job / ohevc_intra_cip, include/ohevc_hip.h): availability, run
 * lengths, switches and the constrained-intra side record as the host helper resolved them.  Used by the software executor of
 * the CPU-only host-logic tests (oracle/sw_exec.c); shares intra_core with ohor_intra_pred, which is what the reference pins.
 * flags: 1 bottom-left, 2 left, 4 up-left, 8 up, 16 up-right, 32 no smoothing, 64 strong, 128 luma edge. */
void ohor_intra_job(int bd, uint8_t *blk, ptrdiff_t stride, int log2, int mode, int flags, int bl_size, int tr_size,
                    const uint8_t *cip_top_bits, const uint8_t *cip_left_bits, int size_max_x, int size_max_y, int x0_nonzero, int y0_nonzero)
{
    cip_view cv;
    memset(&cv, 0, sizeof(cv));
    if (cip_top_bits && cip_left_bits) {
        cv.on = 1;
        memcpy(cv.top, cip_top_bits, 9); memcpy(cv.left, cip_left_bits, 9);
        cv.smx = size_max_x; cv.smy = size_max_y; cv.x0_nonzero = x0_nonzero; cv.y0_nonzero = y0_nonzero;
    }
    intra_core(bd, blk, stride, log2, mode, !!(flags & 1), !!(flags & 2), !!(flags & 4), !!(flags & 8), !!(flags & 16), bl_size, tr_size,
               !!(flags & 32), !!(flags & 64), !!(flags & 128), &cv);
}

/* ------------------------------------------------------------------ SHVC inter-layer up-sampling
 * upsample_base_layer_frame (hevcdsp_template.c:2165-2438): separable resampling of the base-layer picture into the
 * enhancement layer's inter-layer reference picture, 16 phases, 8 taps luma / 4 taps chroma (H.265 tables H.1 / H.2 as
 * hevcdsp.c:948-986 spells them).  The shipped build reaches the same samples block by block through the twelve
 * upsample_filter_block_* slots and the emulated_edge_up_* helpers (hevc_filter.c:1175-1310, ACTIVE_PU_UPSAMPLING hevc.h:117);
 * tests/test_oracle_vs_reference.py checks this restatement against BOTH call sequences of the reference.
 * Reference particulars kept: the horizontal pass stores int16 (wraps above 8 bit); rounding is fixed at 12 bits whatever
 * the bit depth (N_SHIFT, hevcdsp.h:40-41); the vertical pass walks the intermediate columns with a pointer that only
 * advances inside [left, right - 2], i.e. column min(i, right - 1) - left rather than i (:2270,2284,2292). */
static const int8_t kUpLuma[16][8] = {
    {  0, 0,   0, 64,  0,   0, 0,  0 }, {  0, 1,  -3, 63,  4,  -2, 1,  0 }, { -1, 2,  -5, 62,  8,  -3, 1,  0 }, { -1, 3,  -8, 60, 13,  -4, 1,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 }, { -1, 4, -11, 52, 26,  -8, 3, -1 }, { -1, 3,  -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { -1, 4, -10, 34, 45, -11, 4, -1 }, { -1, 4, -10, 31, 47,  -9, 3, -1 }, { -1, 3,  -8, 26, 52, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 }, {  0, 1,  -4, 13, 60,  -8, 3, -1 }, {  0, 1,  -3,  8, 62,  -5, 2, -1 }, {  0, 1,  -2,  4, 63,  -3, 1,  0 } };
static const int8_t kUpChroma[16][4] = {
    {  0, 64,  0,  0 }, { -2, 62,  4,  0 }, { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
    { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, {  0,  4, 62, -2 } };

/* Where an enhancement-layer column / row reads the base layer: integer position of the filter's centre tap and the phase
 * (row of the 16-phase table).  variant 0 = the general formula of upsample_base_layer_frame (:2217-2226,2255-2262,
 * 2317-2325,2364-2372) and of the *_all block slots (:1835-1953); variants 1 / 2 = what the x2 / x1.5 block slots compute
 * instead (:1956-2163): fixed phase patterns taken from the sample's parity / residue, which ignore the phase offsets in
 * add* (so a stream with phase alignment decodes differently through them -- reproduced, not corrected). */
static void shvc_axis(int variant, int chroma, int vertical, int v, int start, int scale, int add, int *pos, int *phase)
{
    int d = v - start;
    if (variant == 0 || (chroma && vertical)) {                   /* chroma rows keep the scaled position in every variant */
        int r16 = ((d * scale + add) >> 12) + (chroma && vertical ? -4 : 0);
        *pos = r16 >> 4; *phase = r16 & 15;
        if (variant == 1) { static const int ph[2] = { 14, 6 };  *phase = ph[v & 1]; }         /* up_sample_filter_chroma_x2_v[y & 1], :2046 */
        if (variant == 2) { static const int ph[3] = { 15, 9, 4 }; *phase = ph[v % 3]; }       /* up_sample_filter_x1_5chroma[y % 3], :2149 */
    } else if (variant == 1) {                                    /* x2: phases 0 / 8 */
        if (!chroma)         { *phase = ((vertical ? d : v) & 1) * 8; *pos = d >> 1; }           /* :1968-1970 (x & 1), :2018-2019 ((y - top) & 1) */
        else                 { *phase = (v & 1) * 8;                   *pos = v >> 1; }           /* :1993-1995: x >> 1, not (x - left) >> 1 */
    } else {                                                      /* x1.5: phases 0 / 11 / 5 */
        static const int ph[3] = { 0, 11, 5 };
        *phase = ph[d % 3]; *pos = (d << 1) / 3;                                                  /* :2072-2074, :2097-2099, :2124-2125 */
    }
}

/* one plane.  taps 8: luma, 4: chroma.  x window [left, right_end], y window [top, bottom_end - 1] in samples of this plane;
 * right_clip = the upper bound of the horizontal position clamp (right_end; right_end - 1 in the chroma pass of the frame
 * function, :2318). */
static void shvc_plane(int bd, int variant, int taps, uint8_t *el, ptrdiff_t el_stride, int el_w, int el_h, const uint8_t *bl, ptrdiff_t bl_stride,
                       int bl_w, int bl_h, int left, int right_end, int right_clip, int top, int bottom_end,
                       int scale_x, int add_x, int scale_y, int add_y)
{
    const int half = taps / 2 - 1, chroma = taps == 4;
    int16_t *tmp = malloc((size_t)el_w * bl_h * sizeof(int16_t));
    for (int i = 0; i < el_w; i++) {                                  /* horizontal pass */
        int pos, phase;
        shvc_axis(variant, chroma, 0, clip3(i, left, right_clip), left, scale_x, add_x, &pos, &phase);
        const int8_t *c = taps == 8 ? kUpLuma[phase] : kUpChroma[phase];
        for (int j = 0; j < bl_h; j++) {
            int acc = 0;
            for (int k = 0; k < taps; k++)
                acc += c[k] * ldpx(bl + (ptrdiff_t)j * bl_stride + (ptrdiff_t)clip3(pos - half + k, 0, bl_w - 1) * psz(bd), bd);
            tmp[(size_t)j * el_w + i] = (int16_t)acc;
        }
    }
    for (int j = 0; j < el_h; j++) {                                  /* vertical pass */
        int pos, phase;
        shvc_axis(variant, chroma, 1, clip3(j, top, bottom_end - 1), top, scale_y, add_y, &pos, &phase);
        const int8_t *c = taps == 8 ? kUpLuma[phase] : kUpChroma[phase];
        for (int i = 0; i < el_w; i++) {
            int col = (i < right_end - 1 ? i : right_end - 1) - left, acc = 0;
            if (col < 0) col = 0;
            for (int k = 0; k < taps; k++)
                acc += c[k] * tmp[(size_t)clip3(pos - half + k, 0, bl_h - 1) * el_w + col];
            stpx(el + (ptrdiff_t)j * el_stride + (ptrdiff_t)i * psz(bd), bd, clip_px((acc + (1 << 11)) >> 12, bd));
        }
    }
    free(tmp);
}

/* el/bl: 3 planes each (4:2:0).  win = scaled_ref_layer_window {left, right, top, bottom} in luma samples,
 * up = UpsamplInf {addXLum, addYLum, scaleXLum, scaleYLum, addXCr, addYCr, scaleXCr, scaleYCr, idx} (hevc.h:347-357).
 * block_slots = 0: upsample_base_layer_frame; 1: the result of the per-block slots upsample_filter_block_*[up.idx], which for
 * idx 1 (x2) and 2 (x1.5) use their own position / phase rules (shvc_axis). */
void ohor_shvc_upsample_frame(int bd, int block_slots, uint8_t *const el[3], const int32_t el_stride[3], int el_w, int el_h,
                              uint8_t *const bl[3], const int32_t bl_stride[3], int bl_w, int bl_h, const int32_t *win, const int32_t *up)
{
    const int variant = block_slots && (up[8] == 1 || up[8] == 2) ? up[8] : 0;
    /* luma: heightBL = min(BL height, EL height) (:2214) */
    shvc_plane(bd, variant, 8, el[0], el_stride[0], el_w, el_h, bl[0], bl_stride[0], bl_w, bl_h <= el_h ? bl_h : el_h,
               win[0], el_w - win[1], el_w - win[1], win[2], el_h - win[3], up[2], up[0], up[3], up[1]);
    /* chroma: sizes halved; heightBL = max(BL height, EL height / 2) / 2 (:2306-2312) */
    {
        int cw = el_w >> 1, ch = el_h >> 1, bw = bl_w >> 1, bh = (bl_h > ch ? bl_h : ch) >> 1;
        int left = win[0] >> 1, right_end = cw - (win[1] >> 1), top = win[2] >> 1, bottom_end = ch - (win[3] >> 1);
        for (int c = 1; c < 3; c++)
            shvc_plane(bd, variant, 4, el[c], el_stride[c], cw, ch, bl[c], bl_stride[c], bw, bh, left, right_end,
                       block_slots ? right_end : right_end - 1, top, bottom_end, up[6], up[4], up[7], up[5]);
    }
}

/* ------------------------------------------------------------------ boundary strengths
 * boundary_strength(), hevc_filter.c:584-700 (C branch): 0 when the two blocks predict from the same pictures with motion closer than one
 * integer sample in both components, 1 otherwise. */
static int oh_mv_apart(const int16_t *a, const int16_t *b)
{
    return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4;
}
static int oh_bs_motion(const oh_bs_field *cur, const oh_bs_field *nb)
{
    if (cur->pred_flag == 3 && nb->pred_flag == 3) {
        if (cur->poc[0] == nb->poc[0] && cur->poc[0] == cur->poc[1] && nb->poc[0] == nb->poc[1])        /* :602-628: all four references are one picture */
            return (oh_mv_apart(nb->mv[0], cur->mv[0]) || oh_mv_apart(nb->mv[1], cur->mv[1])) &&
                   (oh_mv_apart(nb->mv[1], cur->mv[0]) || oh_mv_apart(nb->mv[0], cur->mv[1]));
        if (nb->poc[0] == cur->poc[0] && nb->poc[1] == cur->poc[1])                                      /* :629-645 */
            return oh_mv_apart(nb->mv[0], cur->mv[0]) || oh_mv_apart(nb->mv[1], cur->mv[1]);
        if (nb->poc[1] == cur->poc[0] && nb->poc[0] == cur->poc[1])                                      /* :646-663 */
            return oh_mv_apart(nb->mv[1], cur->mv[0]) || oh_mv_apart(nb->mv[0], cur->mv[1]);
        return 1;
    }
    if (cur->pred_flag != 3 && nb->pred_flag != 3) {                                                     /* :667-697: one vector each */
        const int la = (cur->pred_flag & 1) ? 0 : 1, lb = (nb->pred_flag & 1) ? 0 : 1;
        if (cur->poc[la] != nb->poc[lb]) return 1;
        return oh_mv_apart(cur->mv[la], nb->mv[lb]);
    }
    return 1;
}
/* ff_hevc_deblocking_boundary_strengths(), hevc_filter.c:805-941, for every recorded call */
void ohor_boundary_strengths(const oh_bs_geom *g, const oh_bs_field *mvf, const uint8_t *cbf_luma, const oh_bs_call *calls, int ncalls,
                             uint8_t *vertical_bs, uint8_t *horizontal_bs)
{
    const int lp = g->log2_min_pu_size, lt = g->log2_min_tb_size, ctb_mask = (1 << g->log2_ctb_size) - 1;
#define OH_MVF(x, y) (&mvf[((y) >> lp) * g->min_pu_width + ((x) >> lp)])
#define OH_CBF(x, y) (cbf_luma[((y) >> lt) * g->min_tb_width + ((x) >> lt)])
    for (int c = 0; c < ncalls; c++) {
        const int x0 = calls[c].x0, y0 = calls[c].y0, n = 1 << calls[c].log2_size, fl = calls[c].flags;
        const int across_slices = (fl >> 4) & 1;
        const int is_intra = OH_MVF(x0, y0)->pred_flag == 0;
        if (y0 > 0 && (y0 & 7) == 0) {                                   /* :818-858 the edge above the block */
            const int inside_ctb = y0 & ctb_mask;
            const int ok_slice = across_slices || !(fl & 1), ok_tiles = g->loop_filter_across_tiles || !(fl & 2);
            if ((ok_slice && ok_tiles) || inside_ctb)
                for (int i = 0; i < n; i += 4) {
                    const oh_bs_field *top = OH_MVF(x0 + i, y0 - 1), *cur = OH_MVF(x0 + i, y0);
                    int bs;
                    if (cur->pred_flag == 0 || top->pred_flag == 0) bs = 2;
                    else if (OH_CBF(x0 + i, y0) || OH_CBF(x0 + i, y0 - 1)) bs = 1;
                    else bs = oh_bs_motion(cur, top);
                    horizontal_bs[((x0 + i) + y0 * g->bs_width) >> 2] = (uint8_t)bs;
                }
        }
        if (x0 > 0 && (x0 & 7) == 0) {                                   /* :861-898 the edge left of it */
            const int inside_ctb = x0 & ctb_mask;
            const int ok_slice = across_slices || !(fl & 4), ok_tiles = g->loop_filter_across_tiles || !(fl & 8);
            if ((ok_slice && ok_tiles) || inside_ctb)
                for (int i = 0; i < n; i += 4) {
                    const oh_bs_field *left = OH_MVF(x0 - 1, y0 + i), *cur = OH_MVF(x0, y0 + i);
                    int bs;
                    if (cur->pred_flag == 0 || left->pred_flag == 0) bs = 2;
                    else if (OH_CBF(x0, y0 + i) || OH_CBF(x0 - 1, y0 + i)) bs = 1;
                    else bs = oh_bs_motion(cur, left);
                    vertical_bs[(x0 + (y0 + i) * g->bs_width) >> 2] = (uint8_t)bs;
                }
        }
        if (calls[c].log2_size > lp && !is_intra) {                      /* :900-940 prediction-block edges inside the transform block */
            for (int i = 0; i < n; i += 4) {
                const oh_bs_field *top = OH_MVF(x0 + i, y0 + 7);
                for (int j = 8; j < n; j += 8) {
                    const oh_bs_field *cur = OH_MVF(x0 + i, y0 + j);
                    horizontal_bs[((x0 + i) + (y0 + j) * g->bs_width) >> 2] = (uint8_t)oh_bs_motion(cur, top);
                    top = cur;                                           /* :916 (not the entry at y0 + j + 7) */
                }
            }
            for (int j = 0; j < n; j += 4) {
                const oh_bs_field *left = OH_MVF(x0 + 7, y0 + j);
                for (int i = 8; i < n; i += 8) {
                    const oh_bs_field *cur = OH_MVF(x0 + i, y0 + j);
                    vertical_bs[((x0 + i) + (y0 + j) * g->bs_width) >> 2] = (uint8_t)oh_bs_motion(cur, left);
                    left = cur;
                }
            }
        }
    }
#undef OH_MVF
#undef OH_CBF
}
