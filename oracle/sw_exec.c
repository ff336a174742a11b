/*
 * oracle/sw_exec.c -- TEST INFRASTRUCTURE ONLY: a software executor for RECORDED job streams.
 *
 * With OHHIP_SW_EXEC=1 the HIP-backed reference decoder (_ref/libopenhevc_hip.so, hip_hooks.c) runs its contexts in
 * record-only mode (no device) and installs this file as the frame sink of include/ohevc_debug.h: at every frame end the jobs
 * the recording tables produced are executed HERE, on the decoder's own host frames, by the CPU oracle (hevc_oracle.c, the
 * restatement the reference pins) -- in the phase order of the device executor (ctx.hip): motion compensation, residuals of
 * inter blocks, intra dependency levels (prediction, then residuals), vertical edges, horizontal edges, SAO from a deblocked
 * copy.  The pictures that come out are compared with the untouched decoder (tests/test_stream_cpu.py): that checks, without a
 * GPU, everything on the host side of the C ABI -- recording slots, pointer registry, job builders, dependency levels, filter
 * lag flags, bypass map, slice-thread merging.  What it cannot check are the HIP kernels; the -m gpu tests do that.
 * Nothing in the product links or calls this.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ohevc_tables.h"
#include "ohevc_debug.h"

#define OHX(name) ohor_##name
#include "oracle_api.h"
void ohor_intra_job(int bd, uint8_t *blk, ptrdiff_t stride, int log2, int mode, int flags, int bl_size, int tr_size,
                    const uint8_t *cip_top_bits, const uint8_t *cip_left_bits, int size_max_x, int size_max_y, int x0_nonzero, int y0_nonzero);

static volatile int g_sw_error;
int ohsw_error(void) { return g_sw_error; }

typedef struct host_pic { uint8_t *data[3]; int linesize[3]; int w[3], h[3]; } host_pic;

static int get_host_pic(struct ohevc_ctx *ctx, int slot, host_pic *hp)
{
    int w, h, cfi, bd;
    if (ohevc_tables_host_planes(ctx, slot, hp->data, hp->linesize) != OHEVC_OK || ohevc_pic_info(ctx, slot, &w, &h, &cfi, &bd) != OHEVC_OK)
        return -1;
    for (int c = 0; c < 3; c++) {
        hp->w[c] = c ? w >> (cfi == 1 || cfi == 2) : w;
        hp->h[c] = c ? h >> (cfi == 1) : h;
    }
    return 0;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* (w + 7) x (h + 7) window around (sx, sy) of a reference plane, coordinates clamped to the picture: what the reference gets from
 * its frame edges / emulated_edge_mc (videodsp_template.c:26-100); returns the pointer to sample (sx, sy) inside `buf` */
static uint8_t *mc_window(uint8_t *buf, ptrdiff_t *stride, const host_pic *ref, int pl, int ps, int sx, int sy, int w, int h, int before, int after)
{
    const int ww = w + before + after, wh = h + before + after;
    *stride = (ptrdiff_t)ww * ps;
    for (int y = 0; y < wh; y++) {
        const uint8_t *row = ref->data[pl] + (ptrdiff_t)clampi(sy - before + y, 0, ref->h[pl] - 1) * ref->linesize[pl];
        for (int x = 0; x < ww; x++)
            memcpy(buf + (size_t)y * *stride + (size_t)x * ps, row + (size_t)clampi(sx - before + x, 0, ref->w[pl] - 1) * ps, ps);
    }
    return buf + (size_t)before * *stride + (size_t)before * ps;
}

static void exec_mc(struct ohevc_ctx *ctx, const host_pic *cur, int bd, const ohevc_mc_job *jobs, int n)
{
    const int ps = bd > 8 ? 2 : 1;
    static __thread uint8_t win0[(64 + 8) * (64 + 8) * 2], win1[(64 + 8) * (64 + 8) * 2];
    static __thread int16_t tmp[64 * 64];
    for (int i = 0; i < n; i++) {
        const ohevc_mc_job *j = &jobs[i];
        const int luma = j->plane == 0, before = luma ? 3 : 1, after = luma ? 4 : 2, bi = j->flags & OHEVC_MC_BI, wt = j->flags & OHEVC_MC_WEIGHTED;
        host_pic r0, r1;
        ptrdiff_t s0, s1;
        uint8_t *dst = cur->data[j->plane] + (ptrdiff_t)j->y * cur->linesize[j->plane] + (ptrdiff_t)j->x * ps;
        if (ohevc_debug_wait_picture(ctx, j->ref0) != OHEVC_OK || get_host_pic(ctx, j->ref0, &r0)) { g_sw_error = 1; return; }
        uint8_t *p0 = mc_window(win0, &s0, &r0, j->plane, ps, j->sx0, j->sy0, j->w, j->h, before, after);
        if (!bi) {
            ohor_mc(bd, luma, wt ? OH_MC_UNI_W : OH_MC_UNI, dst, cur->linesize[j->plane], p0, s0, NULL, 0, j->h, j->mx0, j->my0, j->w,
                    j->denom, j->wx0, 0, j->ox0, 0);
            continue;
        }
        if (ohevc_debug_wait_picture(ctx, j->ref1) != OHEVC_OK || get_host_pic(ctx, j->ref1, &r1)) { g_sw_error = 1; return; }
        uint8_t *p1 = mc_window(win1, &s1, &r1, j->plane, ps, j->sx1, j->sy1, j->w, j->h, before, after);
        /* hevc.c:1761-1773: reference 0 into the int16 intermediate (stride MAX_PB_SIZE), then the bi slot with reference 1 */
        ohor_mc(bd, luma, OH_MC_PUT, (uint8_t *)tmp, 64, p0, s0, NULL, 0, j->h, j->mx0, j->my0, j->w, 0, 0, 0, 0, 0);
        ohor_mc(bd, luma, wt ? OH_MC_BI_W : OH_MC_BI, dst, cur->linesize[j->plane], p1, s1, tmp, 64, j->h, j->mx1, j->my1, j->w,
                j->denom, j->wx0, j->wx1, j->ox0, j->ox1);
    }
}

static void exec_tu(const host_pic *cur, int bd, int log2, int kind, const ohevc_tu_job *jobs, int n, const int16_t *coeffs)
{
    const int N = 1 << log2, ps = bd > 8 ? 2 : 1;
    int16_t blk[32 * 32], blk2[32 * 32];
    for (int i = 0; i < n; i++) {
        const ohevc_tu_job *j = &jobs[i];
        uint8_t *plane = cur->data[j->plane];
        const ptrdiff_t stride = cur->linesize[j->plane];
        uint8_t *dst = plane + (ptrdiff_t)j->y * stride + (ptrdiff_t)j->x * ps;
        const int32_t xy[2] = { j->x, j->y };
        if (kind == OHEVC_TU_PCM) {                            /* put_pcm: the samples replace the block */
            const int16_t *s = coeffs + j->coeff_off;
            for (int y = 0; y < N; y++)
                for (int x = 0; x < N; x++) {
                    if (ps == 2) ((uint16_t *)(dst + y * stride))[x] = (uint16_t)s[y * N + x];
                    else dst[y * stride + x] = (uint8_t)s[y * N + x];
                }
        } else if (kind == OHEVC_TU_CROSS) {                   /* hevc_cabac.c:1942-1949, hevc.c:1315-1330 */
            const int kc = j->reserved0 & 15, ky = j->reserved0 >> 4, scale = j->dc;
            memcpy(blk, coeffs + j->reserved1, sizeof(int16_t) * N * N);
            ohor_tu_residual(bd, ky, log2, blk, N);
            if (kc != 15) {
                memcpy(blk2, coeffs + j->coeff_off, sizeof(int16_t) * N * N);
                ohor_tu_residual(bd, kc, log2, blk2, N);
            } else {
                memset(blk2, 0, sizeof(int16_t) * N * N);
            }
            for (int k = 0; k < N * N; k++) blk2[k] = (int16_t)(blk2[k] + ((scale * blk[k]) >> 3));
            ohor_transform_add(bd, log2, dst, stride, blk2);
        } else if (kind == OHEVC_TU_DC) {
            memset(blk, 0, sizeof(int16_t) * N * N);
            blk[0] = j->dc;
            ohor_tu_batch(bd, kind, log2, 1, blk, plane, stride, xy, N);
        } else {
            ohor_tu_batch(bd, kind, log2, 1, coeffs + j->coeff_off, plane, stride, xy, N);
        }
    }
}

static void exec_intra(const host_pic *cur, int bd, const ohevc_intra_job *jobs, int n, const ohevc_intra_cip *cips)
{
    const int ps = bd > 8 ? 2 : 1;
    for (int i = 0; i < n; i++) {
        const ohevc_intra_job *j = &jobs[i];
        uint8_t *blk = cur->data[j->plane] + (ptrdiff_t)j->y * cur->linesize[j->plane] + (ptrdiff_t)j->x * ps;
        const ohevc_intra_cip *c = (j->flags2 & OHEVC_INTRA2_CIP) ? &cips[j->cip_index] : NULL;
        ohor_intra_job(bd, blk, cur->linesize[j->plane], j->log2_size, j->mode, j->flags, j->bottom_left_size, j->top_right_size,
                       c ? c->top_bits : NULL, c ? c->left_bits : NULL, c ? c->size_max_x : 0, c ? c->size_max_y : 0, c ? c->x0_nonzero : 0,
                       c ? c->y0_nonzero : 0);
    }
}

static void exec_dbk(const host_pic *cur, int bd, const ohevc_dbk_job *jobs, int n)
{
    const int ps = bd > 8 ? 2 : 1;
    for (int i = 0; i < n; i++) {
        const ohevc_dbk_job *j = &jobs[i];
        uint8_t *pix = cur->data[j->plane] + (ptrdiff_t)j->y * cur->linesize[j->plane] + (ptrdiff_t)j->x * ps;
        int tc[2] = { j->tc[0], j->tc[1] };
        uint8_t no_p[2] = { !!(j->flags & OHEVC_DBK_NO_P0), !!(j->flags & OHEVC_DBK_NO_P1) };
        uint8_t no_q[2] = { !!(j->flags & OHEVC_DBK_NO_Q0), !!(j->flags & OHEVC_DBK_NO_Q1) };
        const int vertical = !!(j->flags & OHEVC_DBK_VERTICAL_EDGE);
        if (j->plane == 0) ohor_deblock_luma(bd, vertical, pix, cur->linesize[0], j->beta, tc, no_p, no_q);
        else               ohor_deblock_chroma(bd, vertical, pix, cur->linesize[j->plane], tc, no_p, no_q);
    }
}

/* a copy of the picture with a 2-sample replicated border (SAO reads one sample around every block) */
typedef struct copy_pic { uint8_t *base[3], *data[3]; int linesize[3]; } copy_pic;
static void copy_alloc(copy_pic *cp, const host_pic *p, int ps)
{
    for (int c = 0; c < 3; c++) {
        cp->linesize[c] = (p->w[c] + 4) * ps;
        cp->base[c] = malloc((size_t)cp->linesize[c] * (p->h[c] + 4));
        cp->data[c] = cp->base[c] + 2 * cp->linesize[c] + 2 * ps;
    }
}
static void copy_fill(copy_pic *cp, const host_pic *p, int ps, int first_plane)
{
    for (int c = first_plane; c < 3; c++)
        for (int y = -2; y < p->h[c] + 2; y++) {
            const uint8_t *row = p->data[c] + (ptrdiff_t)clampi(y, 0, p->h[c] - 1) * p->linesize[c];
            for (int x = -2; x < p->w[c] + 2; x++)
                memcpy(cp->data[c] + (ptrdiff_t)y * cp->linesize[c] + (ptrdiff_t)x * ps, row + (size_t)clampi(x, 0, p->w[c] - 1) * ps, ps);
        }
}
static void copy_free(copy_pic *cp) { for (int c = 0; c < 3; c++) free(cp->base[c]); }

static void exec_filters(struct ohevc_ctx *ctx, const host_pic *cur, int bd)
{
    const ohevc_dbk_job *dv, *dh;
    const ohevc_sao_job *sao;
    int nv, nh, ns, lagged = 0;
    ohevc_sao_bypass bp;
    const int ps = bd > 8 ? 2 : 1;
    copy_pic twin, lag;
    if (ohevc_debug_filters(ctx, &dv, &nv, &dh, &nh, &sao, &ns, &bp) != OHEVC_OK) { g_sw_error = 1; return; }
    for (int i = 0; i < ns; i++) lagged |= sao[i].quirks != 0;
    exec_dbk(cur, bd, dv, nv);                                 /* all vertical edges, then all horizontal edges (hevc_filter.c:385-580) */
    if (lagged) { copy_alloc(&lag, cur, ps); copy_fill(&lag, cur, ps, 1); }    /* the state the reference's early copy saw (ohevc_hip.h) */
    exec_dbk(cur, bd, dh, nh);
    if (!ns) { if (lagged) copy_free(&lag); return; }
    copy_alloc(&twin, cur, ps);                                /* the reference's sao_frame */
    copy_fill(&twin, cur, ps, 0);
    for (int i = 0; i < ns; i++) {
        const ohevc_sao_job *j = &sao[i];
        const int pl = j->plane, w = j->w, h = j->h, eo = j->klass;
        uint8_t *dst = cur->data[pl] + (ptrdiff_t)j->y * cur->linesize[pl] + (ptrdiff_t)j->x * ps;
        uint8_t *src = twin.data[pl] + (ptrdiff_t)j->y * twin.linesize[pl] + (ptrdiff_t)j->x * ps;
        /* OHEVC_SAO_LAG_*: neighbour samples in the column right of the block, in the listed rows, as they were between the two
         * deblocking passes -- patched into the copy for this job only */
        uint8_t saved[8][2];
        int rows[8], nrows = 0;
        if (j->quirks && j->type == OHEVC_SAO_EDGE && eo != 1 && j->x + w < cur->w[pl]) {
            const int cand[6] = { h - 1, h, -1, 0, 7, 8 };
            const int on[6] = { j->quirks & OHEVC_SAO_LAG_BELOW, j->quirks & OHEVC_SAO_LAG_BELOW, j->quirks & OHEVC_SAO_LAG_ABOVE,
                                j->quirks & OHEVC_SAO_LAG_ABOVE, j->quirks & OHEVC_SAO_LAG_MID, j->quirks & OHEVC_SAO_LAG_MID };
            for (int k = 0; k < 6; k++) {
                const int ny = cand[k], ay = j->y + ny;
                int dup = 0;
                for (int q = 0; q < nrows; q++) dup |= rows[q] == ny;
                if (!on[k] || dup || ay < 0 || ay >= cur->h[pl]) continue;
                uint8_t *s = src + (ptrdiff_t)ny * twin.linesize[pl] + (ptrdiff_t)w * ps;
                memcpy(saved[nrows], s, ps);
                memcpy(s, lag.data[pl] + (ptrdiff_t)ay * lag.linesize[pl] + (ptrdiff_t)(j->x + w) * ps, ps);
                rows[nrows++] = ny;
            }
        }
        if (j->type == OHEVC_SAO_BAND) {
            ohor_sao_band(bd, dst, src, cur->linesize[pl], twin.linesize[pl], j->offset_val, j->klass, w, h);
        } else {
            int borders[4] = { j->borders & 1, (j->borders >> 1) & 1, (j->borders >> 2) & 1, (j->borders >> 3) & 1 };
            uint8_t ve[2] = { j->edges & 1, (j->edges >> 1) & 1 }, he[2] = { (j->edges >> 2) & 1, (j->edges >> 3) & 1 };
            uint8_t de[4] = { (j->edges >> 4) & 1, (j->edges >> 5) & 1, (j->edges >> 6) & 1, (j->edges >> 7) & 1 };
            ohor_sao_edge(bd, j->restore, dst, src, cur->linesize[pl], twin.linesize[pl], j->offset_val, eo, borders, w, h, ve, he, de);
        }
        for (int q = 0; q < nrows; q++) memcpy(src + (ptrdiff_t)rows[q] * twin.linesize[pl] + (ptrdiff_t)w * ps, saved[q], ps);
        /* restore_tqb_pixels (hevc_filter.c:163-193), with the reference's partial walk when exact_reference */
        if (bp.map) {
            const int hs = pl ? bp.chroma_hshift : 0, vs = pl ? bp.chroma_vshift : 0, l2 = bp.log2_min_pu_size;
            const int xlim = bp.exact_reference ? ((j->x << hs) + w) >> l2 : 0x7fffffff, ylim = bp.exact_reference ? ((j->y << vs) + h) >> l2 : 0x7fffffff;
            const int len = (bp.exact_reference && ps == 2) ? ((1 << l2) >> hs) >> 1 : 0x7fffffff;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const int xpu = ((j->x + x) << hs) >> l2, ypu = ((j->y + y) << vs) >> l2;
                    if (xpu < xlim && ypu < ylim && (j->x + x) - ((xpu << l2) >> hs) < len && bp.map[(size_t)ypu * bp.stride + xpu])
                        memcpy(dst + (ptrdiff_t)y * cur->linesize[pl] + (ptrdiff_t)x * ps, src + (ptrdiff_t)y * twin.linesize[pl] + (ptrdiff_t)x * ps, ps);
                }
        }
    }
    copy_free(&twin);
    if (lagged) copy_free(&lag);
}

/* the frame sink (ohevc_debug_set_frame_sink) */
void ohsw_sink(void *user, struct ohevc_ctx *ctx, int stage)
{
    int slot, w, h, cfi, bd;
    host_pic cur;
    (void)user;
    if (ohevc_debug_target(ctx, &slot, &w, &h, &cfi, &bd) != OHEVC_OK || get_host_pic(ctx, slot, &cur)) { g_sw_error = 1; return; }
    if (stage == 0) {
        const ohevc_mc_job *mc;
        const int16_t *coeffs;
        const ohevc_intra_cip *cips;
        int n;
        ohevc_debug_arena(ctx, &coeffs, &cips);
        for (int small = 0; small < 2; small++)               /* inter prediction reads other pictures only: first */
            if (ohevc_debug_mc(ctx, small, &mc, &n) == OHEVC_OK) exec_mc(ctx, &cur, bd, mc, n);
        for (int level = 0; level < ohevc_debug_level_count(ctx); level++) {
            const ohevc_intra_job *ij;
            if (ohevc_debug_level_intra(ctx, level, &ij, &n) == OHEVC_OK) exec_intra(&cur, bd, ij, n, cips);
            for (int log2 = 2; log2 <= 5; log2++)
                for (int kind = 0; kind < OHEVC_TU_NKINDS; kind++) {
                    const ohevc_tu_job *tj;
                    if (ohevc_debug_level_tu(ctx, level, log2, kind, &tj, &n) == OHEVC_OK && n) exec_tu(&cur, bd, log2, kind, tj, n, coeffs);
                }
        }
        {   /* the CTB executor's share (ohevc_dev_ctbs): tasks in raster order, every task's operations in decoding order */
            const ohevc_ctb_task *tasks;
            const uint32_t *ops;
            const ohevc_intra_job *ij;
            const ohevc_tu_job *tj;
            int ntasks = 0, l2 = 0;
            if (ohevc_debug_ctbs(ctx, &tasks, &ntasks, &ops, &ij, &tj, &l2) != OHEVC_OK) { g_sw_error = 1; return; }
            for (int t = 0; t < ntasks; t++) {
                for (int d = 0; d < 4; d++)
                    if (tasks[t].dep[d] >= t) g_sw_error = 1;              /* a task may only wait for earlier ones */
                for (uint32_t k = 0; k < tasks[t].nops; k++) {
                    const uint32_t op = ops[tasks[t].first_op + k], idx = op & 0x1ffffffu;
                    if (!(op >> 31)) exec_intra(&cur, bd, ij + idx, 1, cips);
                    else exec_tu(&cur, bd, 2 + (int)((op >> 29) & 3), (int)((op >> 25) & 15), tj + idx, 1, coeffs);
                }
            }
        }
    } else {
        exec_filters(ctx, &cur, bd);
    }
}

/* see ohor_sao_band_above_range (hevc_oracle.c): band SAO on samples above the bit depth's range since the last reset */
long ohor_sao_band_above_range(int reset);
long ohsw_sao_band_above_range(int reset) { return ohor_sao_band_above_range(reset); }
