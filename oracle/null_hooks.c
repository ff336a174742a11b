/*
 * oracle/null_hooks.c -- TEST INFRASTRUCTURE ONLY (measurement aid).
 *
 * Same link-time interposition as hip_hooks.c, but every HEVCDSPContext / HEVCPredContext / VideoDSPContext slot is
 * replaced by an empty function: _ref/libopenhevc_null.so decodes the syntax and produces no pixels.  Its run time is
 * the part of the reference decoder NO table back-end can remove (entropy decoding, motion-vector and boundary-strength
 * derivation, DPB management): the Amdahl floor quoted in DESIGN.md next to the GPU-backed numbers.
 */
#include <string.h>
#include <stdlib.h>
#include "libavcodec/hevc.h"
#include "libavcodec/thread.h"
#include "libavutil/pixdesc.h"

static void nop(void) {}

static void fill(void *table, size_t bytes)
{
    void (**slot)(void) = table;
    size_t i;
    for (i = 0; i < bytes / sizeof(*slot); i++)
        slot[i] = nop;
}

void ohhip_hevc_dsp_init(HEVCDSPContext *c, int bit_depth)  { (void)bit_depth; fill(c, sizeof(*c)); }
void ohhip_videodsp_init(VideoDSPContext *c, int bpc)        { (void)bpc; fill(c, sizeof(*c)); }
void ohhip_hevc_pred_init(HEVCPredContext *c, int bit_depth) { (void)bit_depth; fill(c, sizeof(*c)); }
int  ohhip_set_new_ref(HEVCContext *s, AVFrame **frame, int poc) { return ff_hevc_set_new_ref(s, frame, poc); }
int  ohhip_frame_rps(HEVCContext *s) { return ff_hevc_frame_rps(s); }
const AVPixFmtDescriptor *ohhip_pix_fmt_desc_get(enum AVPixelFormat f) { return av_pix_fmt_desc_get(f); }
void ohhip_report_progress(ThreadFrame *f, int progress, int field) { ff_thread_report_progress(f, progress, field); }
/* OHNULL_AWAIT=1 / ohnull_set_await(1): keep the reference's frame-thread waits (hevc_await_progress, hevc.c:1951-1958: a picture's thread waits until the thread of
 * a reference picture has reported the rows its motion vectors reach - here reported as that picture is PARSED).  The time of this build is
 * then what a back end with the reference's own dependency structure could reach at best; without it the pictures do not wait for each
 * other at all (a bound no decoder reaches). */
static int g_keep_await = -1;
void ohnull_set_await(int on) { g_keep_await = on != 0; }            /* between decoders (bench.py) */
void ohhip_await_progress(ThreadFrame *f, int progress, int field)
{
    if (g_keep_await < 0) g_keep_await = getenv("OHNULL_AWAIT") && atoi(getenv("OHNULL_AWAIT")) != 0;
    if (g_keep_await) ff_thread_await_progress(f, progress, field);
}
void ohhip_cabac_init(HEVCContext *s, int ctb_addr_ts) { ff_hevc_cabac_init(s, ctb_addr_ts); }
int  ohhip_log2_res_scale_abs(HEVCContext *s, int idx) { return ff_hevc_log2_res_scale_abs(s, idx); }
int  ohhip_res_scale_sign_flag(HEVCContext *s, int idx) { return ff_hevc_res_scale_sign_flag(s, idx); }
void ohhip_hls_filter(HEVCContext *s, int x, int y, int ctb_size) { ff_hevc_hls_filter(s, x, y, ctb_size); }
void ohhip_hls_filters(HEVCContext *s, int x, int y, int ctb_size) { ff_hevc_hls_filters(s, x, y, ctb_size); }
void ohhip_upsample_block(HEVCContext *s, HEVCFrame *ref0, int x0, int y0, int nPbW, int nPbH) { ff_upsample_block(s, ref0, x0, y0, nPbW, nPbH); }

#ifndef OHNULL_NO_BS
/* The tap that pins oracle/hevc_oracle.c's ohor_boundary_strengths (tests/test_oracle_vs_reference.py): with it on, every call of
 * ff_hevc_deblocking_boundary_strengths is logged with what ohevc_bs_call carries, and the motion-field and cbf_luma entries the call reads
 * (its block, the row above, the column to the left) are copied into shadows at that moment.  After the access unit (ONE decoding thread)
 * ohnull_bs_fetch hands out the log, the shadows and the arrays the reference filled - s->horizontal_bs / vertical_bs stay as they are
 * until the next hevc_frame_start (hevc.c:3207-3208). */
typedef struct ohnull_bs_call { uint16_t x0, y0; uint8_t log2_size, flags; uint16_t reserved; } ohnull_bs_call;
typedef struct ohnull_bs_field { int16_t mv[2][2]; int32_t poc[2]; uint32_t pred_flag; } ohnull_bs_field;
typedef struct ohnull_bs_frame {
    const ohnull_bs_call *calls; int32_t ncalls;
    const ohnull_bs_field *mvf; const uint8_t *cbf_luma;
    const uint8_t *vertical_bs, *horizontal_bs;                /* the reference's */
    int32_t n_vertical, n_horizontal;
    int32_t min_pu_width, min_pu_height, log2_min_pu_size, min_tb_width, min_tb_height, log2_min_tb_size, log2_ctb_size, bs_width, width, height;
    int32_t loop_filter_across_tiles;
} ohnull_bs_frame;
static int g_tap;
static ohnull_bs_call *g_calls;
static int g_ncalls, g_cap;
static ohnull_bs_field *g_mvf;
static uint8_t *g_cbf;
static size_t g_mvf_n, g_cbf_n;
static const HEVCContext *g_tap_s;

void ohnull_bs_tap(int on)            /* on: start a fresh log (call before every access unit); off: stop logging */
{
    g_tap = on; g_ncalls = 0; g_tap_s = NULL;
    if (g_mvf) memset(g_mvf, 0, g_mvf_n * sizeof(*g_mvf));
    if (g_cbf) memset(g_cbf, 0, g_cbf_n);
}
int ohnull_bs_fetch(ohnull_bs_frame *out)
{
    const HEVCContext *s = g_tap_s;
    if (!s || !out) return -1;
    memset(out, 0, sizeof(*out));
    out->calls = g_calls; out->ncalls = g_ncalls; out->mvf = g_mvf; out->cbf_luma = g_cbf;
    out->vertical_bs = s->vertical_bs; out->horizontal_bs = s->horizontal_bs;
    out->n_vertical = out->n_horizontal = s->bs_width * s->bs_height;      /* what hevc_frame_start clears (the allocations are larger, hevc.c:170-171) */
    out->min_pu_width = s->sps->min_pu_width; out->min_pu_height = s->sps->min_pu_height; out->log2_min_pu_size = s->sps->log2_min_pu_size;
    out->min_tb_width = s->sps->min_tb_width; out->min_tb_height = s->sps->min_tb_height; out->log2_min_tb_size = s->sps->log2_min_tb_size;
    out->log2_ctb_size = s->sps->log2_ctb_size; out->bs_width = s->bs_width; out->width = s->sps->width; out->height = s->sps->height;
    out->loop_filter_across_tiles = s->pps->loop_filter_across_tiles_enabled_flag;
    return 0;
}
static void tap_call(HEVCContext *s, int x0, int y0, int log2_size)
{
    const HEVCLocalContext *lc = s->HEVClc;
    const int lp = s->sps->log2_min_pu_size, lt = s->sps->log2_min_tb_size, n = 1 << log2_size;
    const size_t mvf_n = (size_t)s->sps->min_pu_width * s->sps->min_pu_height, cbf_n = (size_t)s->sps->min_tb_width * s->sps->min_tb_height;
    int x, y;
    if (mvf_n != g_mvf_n || cbf_n != g_cbf_n) {
        free(g_mvf); free(g_cbf);
        g_mvf = calloc(mvf_n, sizeof(*g_mvf)); g_cbf = calloc(cbf_n, 1); g_mvf_n = mvf_n; g_cbf_n = cbf_n;
    }
    if (g_ncalls == g_cap) { g_cap = g_cap ? 2 * g_cap : 4096; g_calls = realloc(g_calls, (size_t)g_cap * sizeof(*g_calls)); }
    g_calls[g_ncalls].x0 = (uint16_t)x0; g_calls[g_ncalls].y0 = (uint16_t)y0; g_calls[g_ncalls].log2_size = (uint8_t)log2_size;
    g_calls[g_ncalls].flags = (uint8_t)((lc->slice_or_tiles_up_boundary & 3) | ((lc->slice_or_tiles_left_boundary & 3) << 2) |
                                        (s->sh.slice_loop_filter_across_slices_enabled_flag ? 16 : 0));
    g_calls[g_ncalls].reserved = 0;
    g_ncalls++;
    g_tap_s = s;
    for (y = FFMAX(y0 - 1, 0); y < FFMIN(y0 + n, s->sps->height); y++)
        for (x = FFMAX(x0 - 1, 0); x < FFMIN(x0 + n, s->sps->width); x++) {
            const MvField *f = &s->ref->tab_mvf[(y >> lp) * s->sps->min_pu_width + (x >> lp)];
            ohnull_bs_field *o = &g_mvf[(y >> lp) * s->sps->min_pu_width + (x >> lp)];
            o->mv[0][0] = f->mv[0].x; o->mv[0][1] = f->mv[0].y; o->mv[1][0] = f->mv[1].x; o->mv[1][1] = f->mv[1].y;
            o->poc[0] = f->poc[0]; o->poc[1] = f->poc[1]; o->pred_flag = f->pred_flag;
            g_cbf[(y >> lt) * s->sps->min_tb_width + (x >> lt)] = s->cbf_luma[(y >> lt) * s->sps->min_tb_width + (x >> lt)];
            if (x > x0 && y > y0 && !((x | y) & 3)) x += 3;       /* whole 4x4 units inside the block: one visit is enough */
        }
}
void ohhip_deblocking_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size)
{
    if (g_tap) tap_call(s, x0, y0, log2_trafo_size);
    ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_trafo_size);
}
#endif
#ifdef OHNULL_NO_BS
/* oracle/Makefile target `nobs`: the boundary strengths are not derived either (measurement of their share of the front end) */
void ohnull_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size) { (void)s; (void)x0; (void)y0; (void)log2_trafo_size; }
#endif
