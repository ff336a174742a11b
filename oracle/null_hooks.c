/*
 * oracle/null_hooks.c -- TEST INFRASTRUCTURE ONLY (measurement aid).
 *
 * Same link-time interposition as hip_hooks.c, but every HEVCDSPContext / HEVCPredContext / VideoDSPContext slot is
 * replaced by an empty function: _ref/libopenhevc_null.so decodes the syntax and produces no pixels.  Its run time is
 * the part of the reference decoder NO table back-end can remove (entropy decoding, motion-vector and boundary-strength
 * derivation, DPB management): the Amdahl floor quoted in DESIGN.md next to the GPU-backed numbers.
 */
#include <string.h>
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
void ohhip_await_progress(ThreadFrame *f, int progress, int field) { (void)f; (void)progress; (void)field; }
void ohhip_cabac_init(HEVCContext *s, int ctb_addr_ts) { ff_hevc_cabac_init(s, ctb_addr_ts); }
int  ohhip_log2_res_scale_abs(HEVCContext *s, int idx) { return ff_hevc_log2_res_scale_abs(s, idx); }
int  ohhip_res_scale_sign_flag(HEVCContext *s, int idx) { return ff_hevc_res_scale_sign_flag(s, idx); }
void ohhip_hls_filter(HEVCContext *s, int x, int y, int ctb_size) { ff_hevc_hls_filter(s, x, y, ctb_size); }
void ohhip_hls_filters(HEVCContext *s, int x, int y, int ctb_size) { ff_hevc_hls_filters(s, x, y, ctb_size); }

#ifndef OHNULL_NO_BS
void ohhip_deblocking_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size) { ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_trafo_size); }
#endif
#ifdef OHNULL_NO_BS
/* oracle/Makefile target `nobs`: the boundary strengths are not derived either (measurement of their share of the front end) */
void ohnull_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size) { (void)s; (void)x0; (void)y0; (void)log2_trafo_size; }
#endif
