/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY.  Compiled against the REFERENCE's own headers and linked
 * with the reference's own hevcdsp.o / hevcpred.o (built unmodified from /root/reference by
 * oracle/Makefile) into oracle/_ref/libhevcref.so.  Every function below is a call-through into the
 * reference's function-pointer tables as filled by ff_hevc_dsp_init() (libavcodec/hevcdsp.c:1071-1328)
 * and ff_hevc_pred_init() (libavcodec/hevcpred.c:47-86); no algorithm lives in this file.
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "libavcodec/get_bits.h"
#include "libavcodec/hevc.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/hevcpred.h"

#define OHX(name) ohref_##name
#include "oracle_api.h"

/* the reference's table index for a PU width, ff_hevc_pel_weight[] (libavcodec/hevc.c:42);
 * hevc.c itself is not part of this build, so the mapping is restated here */
static int pel_weight_idx(int w)
{
    switch (w) {
    case 2: return 0; case 4: return 1; case 6: return 2; case 8: return 3; case 12: return 4;
    case 16: return 5; case 24: return 6; case 32: return 7; case 48: return 8; case 64: return 9;
    }
    return -1;
}

static HEVCDSPContext  g_dsp[15];
static HEVCPredContext g_pred[15];
static int g_inited[15];
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void ensure(int bd)
{
    pthread_mutex_lock(&g_lock);
    if (!g_inited[bd]) {
        ff_hevc_dsp_init(&g_dsp[bd], bd);
        ff_hevc_pred_init(&g_pred[bd], bd);
        g_inited[bd] = 1;
    }
    pthread_mutex_unlock(&g_lock);
}

int ohref_available(void) { return 1; }

void ohref_tu_residual(int bd, int kind, int log2, int16_t *coeffs, int col_limit)
{
    ensure(bd);
    HEVCDSPContext *c = &g_dsp[bd];
    switch (kind) {
    case OH_TU_IDCT:   c->idct[log2 - 2](coeffs, col_limit); break;
    case OH_TU_DC:     c->idct_dc[log2 - 2](coeffs); break;
    case OH_TU_DST4:   c->idct_4x4_luma(coeffs); break;
    case OH_TU_SKIP:   c->transform_skip(coeffs, log2); break;
    case OH_TU_SKIP_RDPCM_H: c->transform_skip(coeffs, log2); c->transform_rdpcm(coeffs, log2, 0); break;
    case OH_TU_SKIP_RDPCM_V: c->transform_skip(coeffs, log2); c->transform_rdpcm(coeffs, log2, 1); break;
    case OH_TU_BYPASS: break;
    case OH_TU_BYPASS_RDPCM_H: c->transform_rdpcm(coeffs, log2, 0); break;
    case OH_TU_BYPASS_RDPCM_V: c->transform_rdpcm(coeffs, log2, 1); break;
    }
}

void ohref_transform_add(int bd, int log2, uint8_t *dst, ptrdiff_t stride, int16_t *res)
{
    ensure(bd);
    g_dsp[bd].transform_add[log2 - 2](dst, res, stride);
}

void ohref_tu_batch(int bd, int kind, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                    ptrdiff_t stride, const int32_t *xy, int col_limit)
{
    int nn = 1 << (2 * log2), ps = bd > 8 ? 2 : 1;
    DECLARE_ALIGNED(32, int16_t, tmp)[32 * 32];
    ensure(bd);
    for (int i = 0; i < n; i++) {
        memcpy(tmp, coeffs + (size_t)i * nn, nn * sizeof(int16_t));
        ohref_tu_residual(bd, kind, log2, tmp, col_limit);
        g_dsp[bd].transform_add[log2 - 2](plane + (ptrdiff_t)xy[2 * i + 1] * stride + xy[2 * i] * ps, tmp, stride);
    }
}

struct tu_mt_arg { int bd, kind, log2, n, col_limit; const int16_t *coeffs; uint8_t *plane; ptrdiff_t stride; const int32_t *xy; };
static void *tu_mt_worker(void *p)
{
    struct tu_mt_arg *a = p;
    ohref_tu_batch(a->bd, a->kind, a->log2, a->n, a->coeffs, a->plane, a->stride, a->xy, a->col_limit);
    return NULL;
}
void ohref_tu_batch_mt(int bd, int kind, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                       ptrdiff_t stride, const int32_t *xy, int col_limit, int threads)
{
    pthread_t th[256];
    struct tu_mt_arg a[256];
    int nn = 1 << (2 * log2);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    ensure(bd);
    for (int t = 0; t < threads; t++) {
        int lo = (int)((long long)n * t / threads), hi = (int)((long long)n * (t + 1) / threads);
        a[t] = (struct tu_mt_arg){ bd, kind, log2, hi - lo, col_limit, coeffs + (size_t)lo * nn, plane, stride, xy + 2 * lo };
        pthread_create(&th[t], NULL, tu_mt_worker, &a[t]);
    }
    for (int t = 0; t < threads; t++)
        pthread_join(th[t], NULL);
}

void ohref_mc(int bd, int luma, int variant, uint8_t *dst, ptrdiff_t dststride,
              uint8_t *src, ptrdiff_t srcstride, int16_t *src2, ptrdiff_t src2stride,
              int height, int mx, int my, int width,
              int denom, int wx0, int wx1, int ox0, int ox1)
{
    ensure(bd);
    HEVCDSPContext *c = &g_dsp[bd];
    int idx = pel_weight_idx(width), a = !!my, b = !!mx;
    if (idx < 0) return;
    if (luma) {
        switch (variant) {
        case OH_MC_PUT:   c->put_hevc_qpel[idx][a][b]((int16_t *)dst, dststride, src, srcstride, height, mx, my, width); break;
        case OH_MC_UNI:   c->put_hevc_qpel_uni[idx][a][b](dst, dststride, src, srcstride, height, mx, my, width); break;
        case OH_MC_UNI_W: c->put_hevc_qpel_uni_w[idx][a][b](dst, dststride, src, srcstride, height, denom, wx0, ox0, mx, my, width); break;
        case OH_MC_BI:    c->put_hevc_qpel_bi[idx][a][b](dst, dststride, src, srcstride, src2, src2stride, height, mx, my, width); break;
        case OH_MC_BI_W:  c->put_hevc_qpel_bi_w[idx][a][b](dst, dststride, src, srcstride, src2, src2stride, height, denom, wx0, wx1, ox0, ox1, mx, my, width); break;
        }
    } else {
        switch (variant) {
        case OH_MC_PUT:   c->put_hevc_epel[idx][a][b]((int16_t *)dst, dststride, src, srcstride, height, mx, my, width); break;
        case OH_MC_UNI:   c->put_hevc_epel_uni[idx][a][b](dst, dststride, src, srcstride, height, mx, my, width); break;
        case OH_MC_UNI_W: c->put_hevc_epel_uni_w[idx][a][b](dst, dststride, src, srcstride, height, denom, wx0, ox0, mx, my, width); break;
        case OH_MC_BI:    c->put_hevc_epel_bi[idx][a][b](dst, dststride, src, srcstride, src2, src2stride, height, mx, my, width); break;
        /* call order of hevc.c:1940-1948 (the header's parameter NAMES are permuted, hevcdsp.h:92-95) */
        case OH_MC_BI_W:  c->put_hevc_epel_bi_w[idx][a][b](dst, dststride, src, srcstride, src2, src2stride, height, denom, wx0, wx1, ox0, ox1, mx, my, width); break;
        }
    }
}

void ohref_deblock_luma(int bd, int vertical_edge, uint8_t *pix, ptrdiff_t stride, int beta,
                        int *tc, uint8_t *no_p, uint8_t *no_q)
{
    ensure(bd);
    if (vertical_edge) g_dsp[bd].hevc_v_loop_filter_luma_c(pix, stride, beta, tc, no_p, no_q);
    else               g_dsp[bd].hevc_h_loop_filter_luma_c(pix, stride, beta, tc, no_p, no_q);
}

void ohref_deblock_chroma(int bd, int vertical_edge, uint8_t *pix, ptrdiff_t stride,
                          int *tc, uint8_t *no_p, uint8_t *no_q)
{
    ensure(bd);
    if (vertical_edge) g_dsp[bd].hevc_v_loop_filter_chroma_c(pix, stride, tc, no_p, no_q);
    else               g_dsp[bd].hevc_h_loop_filter_chroma_c(pix, stride, tc, no_p, no_q);
}

void ohref_sao_band(int bd, uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src,
                    const int16_t *offset_val, int band_position, int width, int height)
{
    SAOParams sao;
    int borders[4] = { 0, 0, 0, 0 };
    ensure(bd);
    memset(&sao, 0, sizeof(sao));
    memcpy(sao.offset_val[0], offset_val, 5 * sizeof(int16_t));
    sao.band_position[0] = band_position;
    g_dsp[bd].sao_band_filter(dst, src, stride_dst, stride_src, &sao, borders, width, height, 0);
}

void ohref_sao_edge(int bd, int restore, uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst,
                    ptrdiff_t stride_src, const int16_t *offset_val, int eo_class, int *borders,
                    int width, int height, uint8_t *vert_edge, uint8_t *horiz_edge, uint8_t *diag_edge)
{
    SAOParams sao;
    ensure(bd);
    memset(&sao, 0, sizeof(sao));
    memcpy(sao.offset_val[0], offset_val, 5 * sizeof(int16_t));
    sao.eo_class[0] = eo_class;
    g_dsp[bd].sao_edge_filter[!!restore](dst, src, stride_dst, stride_src, &sao, borders, width, height, 0,
                                         vert_edge, horiz_edge, diag_edge);
}

void ohref_pred_planar(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride)
{
    ensure(bd);
    /* note: pred_* take the stride in ELEMENTS (hevcpred_template.c:67,329-343) */
    g_pred[bd].pred_planar[log2 - 2](src, top, left, stride / (bd > 8 ? 2 : 1));
}
void ohref_pred_dc(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int c_idx)
{
    ensure(bd);
    g_pred[bd].pred_dc(src, top, left, stride / (bd > 8 ? 2 : 1), log2, c_idx);
}
void ohref_pred_angular(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left,
                        ptrdiff_t stride, int c_idx, int mode)
{
    ensure(bd);
    g_pred[bd].pred_angular[log2 - 2](src, top, left, stride / (bd > 8 ? 2 : 1), c_idx, mode);
}

/* ---- intra_pred(): needs a HEVCContext; build the minimum the template reads
 *      (hevcpred_template.c:73-116,289-343 -- list in SURVEY.md 8c) ---- */
void ohref_intra_pred(int bd, const oh_intra_pic *pic, int x0, int y0, int log2, int c_idx, int mode,
                      int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right)
{
    ensure(bd);
    HEVCContext      *s   = calloc(1, sizeof(*s));
    HEVCLocalContext *lc  = calloc(1, sizeof(*lc));
    HEVCSPS          *sps = calloc(1, sizeof(*sps));
    HEVCPPS          *pps = calloc(1, sizeof(*pps));
    HEVCFrame        *ref = calloc(1, sizeof(*ref));
    AVFrame          *frm = calloc(1, sizeof(*frm));
    int cfi = pic->chroma_format_idc;

    sps->width  = pic->width;
    sps->height = pic->height;
    sps->chroma_array_type = cfi;
    sps->hshift[0] = sps->vshift[0] = 0;
    sps->hshift[1] = sps->hshift[2] = (cfi == 1 || cfi == 2);
    sps->vshift[1] = sps->vshift[2] = (cfi == 1);
    sps->log2_ctb_size    = pic->log2_ctb_size;
    sps->log2_min_tb_size = pic->log2_min_tb_size;
    sps->log2_min_pu_size = pic->log2_min_pu_size;
    sps->tb_mask       = (1 << (pic->log2_ctb_size - pic->log2_min_tb_size)) - 1;   /* hevc_ps.c:2011 */
    sps->min_pu_width  = (pic->width  + (1 << pic->log2_min_pu_size) - 1) >> pic->log2_min_pu_size;
    sps->min_pu_height = (pic->height + (1 << pic->log2_min_pu_size) - 1) >> pic->log2_min_pu_size;
    sps->pixel_shift   = bd > 8;
    sps->sps_strong_intra_smoothing_enable_flag = pic->strong_intra_smoothing;
    sps->spsRext.intra_smoothing_disabled_flag   = pic->intra_smoothing_disabled;
    pps->constrained_intra_pred_flag = pic->constrained_intra_pred;

    /* CTB-local z-scan table with a -1 border, as built by hevc_ps.c:2551-2567 (no tiles) */
    int n = sps->tb_mask + 2, ld = pic->log2_ctb_size - pic->log2_min_tb_size;
    int *tab = malloc(sizeof(int) * n * n);
    for (int i = 0; i < n * n; i++) tab[i] = 0;
    for (int y = 0; y < n; y++) { tab[y * n] = -1; tab[y] = -1; }
    for (int y = 0; y < n - 1; y++)
        for (int x = 0; x < n - 1; x++) {
            int v = 0;
            for (int i = 0; i < ld; i++) {
                int m = 1 << i;
                v += (m & x ? m * m : 0) + (m & y ? 2 * m * m : 0);
            }
            tab[(y + 1) * n + (x + 1)] = v;
        }
    pps->min_tb_addr_zs_tab = tab;
    pps->min_tb_addr_zs     = &tab[n + 1];

    MvField *mvf = NULL;
    if (pic->constrained_intra_pred && pic->is_intra) {
        int cnt = sps->min_pu_width * sps->min_pu_height;
        mvf = calloc(cnt, sizeof(*mvf));
        for (int i = 0; i < cnt; i++)
            mvf[i].pred_flag = pic->is_intra[i] ? PF_INTRA : PF_L0;
    }
    ref->tab_mvf = mvf;
    for (int p = 0; p < 3; p++) { frm->data[p] = pic->data[p]; frm->linesize[p] = pic->linesize[p]; }

    lc->na.cand_bottom_left = cand_bottom_left;
    lc->na.cand_left        = cand_left;
    lc->na.cand_up_left     = cand_up_left;
    lc->na.cand_up          = cand_up;
    lc->na.cand_up_right    = cand_up_right;
    lc->tu.intra_pred_mode   = mode;
    lc->tu.intra_pred_mode_c = mode;

    s->HEVClc = lc;
    s->sps = sps; s->pps = pps;
    s->frame = frm; s->ref = ref;
    s->hpc = g_pred[bd];
    g_pred[bd].intra_pred[log2 - 2](s, x0, y0, c_idx);

    free(mvf); free(tab); free(frm); free(ref); free(pps); free(sps); free(lc); free(s);
}
