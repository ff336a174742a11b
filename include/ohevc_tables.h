/*
 * ohevc_tables.h -- the drop-in: fillers for the reference's own function-pointer tables.
 *
 * The reference selects its kernels by filling HEVCDSPContext / HEVCPredContext / VideoDSPContext once per SPS
 * (set_sps, hevc.c:421-423) and lets arch back-ends override slots at the end of the init functions
 * (`if (ARCH_X86) ff_hevcdsp_init_x86(hevcdsp, bit_depth);` hevcdsp.c:1326-1327, hevcpred.c:84).  The functions below
 * are a third back-end for exactly that hook: after them, every table call RECORDS a job into the calling thread's
 * ohevc_ctx (include/ohevc_ctx.h) instead of computing on host memory.  INTEGRATION.md shows the three-line patch.
 *
 * The structs here mirror the reference's layouts slot for slot (same order, same signatures); they are ABI
 * declarations, not code.  Every slot is overridden, the thirteen SHVC up-sampling slots and vdsp.emulated_edge_up_{h,v} included
 * (ohevc_tables_upsample_frame below); put_pcm keeps the reference's bit reader on the host and ships the samples as blocks.
 *
 * Pointer arguments are HOST addresses inside the reference's frame buffers; they are translated to (picture slot,
 * plane, x, y) through the registry fed by ohevc_tables_begin_frame / ohevc_tables_register_picture.
 */
#ifndef OHEVC_TABLES_H
#define OHEVC_TABLES_H

#include "ohevc_ctx.h"

#ifdef __cplusplus
extern "C" {
#endif

struct GetBitContext;
struct AVFrame;
struct UpsamplInf;
struct HEVCWindow;

/* HEVCWindow (hevc.h:384-389) and UpsamplInf (hevc.h:347-357) */
typedef struct ohevc_HEVCWindow { int left_offset, right_offset, top_offset, bottom_offset; } ohevc_HEVCWindow;
typedef struct ohevc_UpsamplInf { int addXLum, addYLum, scaleXLum, scaleYLum, addXCr, addYCr, scaleXCr, scaleYCr, idx; } ohevc_UpsamplInf;

/* SAOParams, libavcodec/hevc.h:514-523 */
typedef struct ohevc_SAOParams {
    uint8_t offset_abs[3][4];
    uint8_t offset_sign[3][4];
    uint8_t band_position[3];
    int16_t offset_val[3][5];
    uint8_t eo_class[3];
    uint8_t type_idx[3];
} ohevc_SAOParams;

/* HEVCDSPContext, libavcodec/hevcdsp.h:44-124 (COM16_C806_EMT == 0, hevc.h:41) */
typedef struct ohevc_HEVCDSPContext {
    void (*put_pcm)(uint8_t *dst, ptrdiff_t stride, int width, int height, struct GetBitContext *gb, int pcm_bit_depth);
    void (*transform_add[4])(uint8_t *dst, int16_t *coeffs, ptrdiff_t stride);
    void (*transform_skip)(int16_t *coeffs, int16_t log2_size);
    void (*transform_rdpcm)(int16_t *coeffs, int16_t log2_size, int mode);
    void (*idct_4x4_luma)(int16_t *coeffs);
    void (*idct[4])(int16_t *coeffs, int col_limit);
    void (*idct_dc[4])(int16_t *coeffs);
    void (*sao_band_filter)(uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, ohevc_SAOParams *sao,
                            int *borders, int width, int height, int c_idx);
    void (*sao_edge_filter[2])(uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src, ohevc_SAOParams *sao,
                               int *borders, int width, int height, int c_idx, uint8_t *vert_edge, uint8_t *horiz_edge,
                               uint8_t *diag_edge);
    void (*put_hevc_qpel[10][2][2])(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                    int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_uni[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                        int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_uni_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                          int height, int denom, int wx, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_bi[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                       int16_t *src2, ptrdiff_t src2stride, int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_bi_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                         int16_t *src2, ptrdiff_t src2stride, int height, int denom, int wx0, int wx1,
                                         int ox0, int ox1, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel[10][2][2])(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                    int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_uni[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                        int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_uni_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                          int height, int denom, int wx, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_bi[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                       int16_t *src2, ptrdiff_t src2stride, int height, intptr_t mx, intptr_t my, int width);
    /* call order (denom, wx0, wx1, ox0, ox1) as at hevc.c:1940-1948; the reference header permutes the NAMES only */
    void (*put_hevc_epel_bi_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                         int16_t *src2, ptrdiff_t src2stride, int height, int denom, int wx0, int wx1,
                                         int ox0, int ox1, intptr_t mx, intptr_t my, int width);
    void (*hevc_h_loop_filter_luma)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_luma)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_h_loop_filter_chroma)(uint8_t *pix, ptrdiff_t stride, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_chroma)(uint8_t *pix, ptrdiff_t stride, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_h_loop_filter_luma_c)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_luma_c)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_h_loop_filter_chroma_c)(uint8_t *pix, ptrdiff_t stride, int *tc, uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_chroma_c)(uint8_t *pix, ptrdiff_t stride, int *tc, uint8_t *no_p, uint8_t *no_q);
    /* SHVC inter-layer slots (hevcdsp.h:106-123).  upsample_base_layer_frame takes AVFrames: left as filled, the reference-side
     * stub of INTEGRATION.md section 2b calls ohevc_tables_upsample_frame instead; the twelve block slots are overridden. */
    void *upsample_base_layer_frame;
    void (*upsample_filter_block_luma_h[3])(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride, int x_EL, int x_BL,
                                            int block_w, int block_h, int widthEL, const struct ohevc_HEVCWindow *Enhscal,
                                            struct ohevc_UpsamplInf *up_info);
    void (*upsample_filter_block_luma_v[3])(uint8_t *dst, ptrdiff_t dststride, int16_t *src, ptrdiff_t srcstride, int y_BL, int x_EL, int y_EL,
                                            int block_w, int block_h, int widthEL, int heightEL, const struct ohevc_HEVCWindow *Enhscal,
                                            struct ohevc_UpsamplInf *up_info);
    void (*upsample_filter_block_cr_h[3])(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride, int x_EL, int x_BL,
                                          int block_w, int block_h, int widthEL, const struct ohevc_HEVCWindow *Enhscal,
                                          struct ohevc_UpsamplInf *up_info);
    void (*upsample_filter_block_cr_v[3])(uint8_t *dst, ptrdiff_t dststride, int16_t *src, ptrdiff_t srcstride, int y_BL, int x_EL, int y_EL,
                                          int block_w, int block_h, int widthEL, int heightEL, const struct ohevc_HEVCWindow *Enhscal,
                                          struct ohevc_UpsamplInf *up_info);
} ohevc_HEVCDSPContext;

/* VideoDSPContext, libavcodec/videodsp.h:44-91 */
typedef struct ohevc_VideoDSPContext {
    void (*emulated_edge_mc)(uint8_t *dst, const uint8_t *src, ptrdiff_t dst_linesize, ptrdiff_t src_linesize,
                             int block_w, int block_h, int src_x, int src_y, int w, int h);
    int (*emulated_edge_up_h)(uint8_t *src, ptrdiff_t linesize, const struct ohevc_HEVCWindow *Enhscal, int block_w, int block_h,
                              int bl_edge_left, int bl_edge_right, int shift);
    int (*emulated_edge_up_v)(int16_t *src, ptrdiff_t linesize, const struct ohevc_HEVCWindow *Enhscal, int block_w, int block_h,
                              int src_x, int bl_edge_up, int bl_edge_bottom, int wEL, int shift);
    void (*prefetch)(uint8_t *buf, ptrdiff_t stride, int h);
} ohevc_VideoDSPContext;

/* ---- the hooks: call right after the reference's own init (same place as ff_hevcdsp_init_x86 / ff_videodsp_init_x86) */
void ohevc_hevcdsp_init_hip(ohevc_HEVCDSPContext *c, int bit_depth);
void ohevc_videodsp_init_hip(ohevc_VideoDSPContext *c, int bit_depth);

/* ---- per-thread binding and the pointer registry */
int  ohevc_tables_bind(ohevc_ctx *ctx);                    /* this thread's table calls record into ctx (NULL unbinds) */
/* Slice threads (the reference's WPP-row / tile workers, hls_decode_entry_wpp / hls_decode_entry_tiles, hevc.c:2744-2920,
 * run through avctx->execute2): all workers of a picture record into the SAME context.  Turn this on for the context and call
 * ohevc_tables_bind(ctx) at the top of each worker entry function.  Every worker then records into arrays of its own
 * (ohevc_ctx_set_concurrent: merged when the frame is executed, no lock on the recording path); per-thread call-sequence
 * state (pending transform, first half of a bi-prediction, edge-emulation windows) is thread-local anyway; only the filter-lag
 * bookkeeping of 16x16-CTB streams takes a spin lock.  What makes it safe: WPP / tile decoding never reads a neighbour
 * another worker has not yet parsed (2-CTB lag, hevc.c:2779), so every intra block still sees its neighbours' dependency
 * levels. */
int  ohevc_tables_set_concurrent(ohevc_ctx *ctx, int on);
/* host planes of a picture living in picture-store slot `slot` (a DPB entry): used to resolve MC source pointers */
int  ohevc_tables_register_picture(ohevc_ctx *ctx, int slot, uint8_t *const data[3], const int linesize[3]);
int  ohevc_tables_unregister_picture(ohevc_ctx *ctx, int slot);
/* hevc_frame_start: picture `slot` (already registered) becomes the reconstruction target */
int  ohevc_tables_begin_frame(ohevc_ctx *ctx, int slot);
/* frame end: run everything; with download != 0 the final planes are copied back into the registered host buffers
 * (needed wherever the CPU still reads pixels: output, MD5 check hevc.c:4146-4181) */
int  ohevc_tables_end_frame(ohevc_ctx *ctx, int download);
/* the same; *issued_at (may be NULL) = CLOCK_MONOTONIC seconds at which the issue of the frame end was over - the rest of the call is the
 * wait for the device and the copy-back (per-picture timelines: integration/hip_backend.h, trace_path) */
int  ohevc_tables_end_frame2(ohevc_ctx *ctx, int download, double *issued_at);
/* The same, with the host-side work of the frame end (staging, upload, launches) on the store's issuer thread: returns at once
 * (ohevc_frame_end_async, ohevc_ctx.h).  For decoders with frame threads.  With download != 0 the copy-back into the registered host
 * planes is queued behind the picture's device work; ohevc_tables_fetch_picture(ctx, slot) - where the application takes the picture
 * out - returns when it has landed. */
int  ohevc_tables_end_frame_async(ohevc_ctx *ctx, int download);
int  ohevc_tables_fetch_picture(ohevc_ctx *ctx, int slot);
/* restore_tqb_pixels (hevc_filter.c:163-193) works on host pixels and is lost behind recording tables: before
 * ohevc_tables_end_frame hand over s->is_pcm (s->sps->min_pu_width x min_pu_height bytes, s->sps->log2_min_pu_size) whenever
 * pps->transquant_bypass_enable_flag || (sps->pcm_enabled_flag && sps->pcm.loop_filter_disable_flag).  With
 * ohevc_tables_emulate_filter_lag on, the reference's partial restore is reproduced bit for bit (ohevc_sao_bypass.exact_reference) */
int  ohevc_tables_set_bypass_map(ohevc_ctx *ctx, const uint8_t *is_pcm, int min_pu_width, int min_pu_height, int log2_min_pu_size);
/* SHVC inter-layer up-sampling.  The twelve upsample_filter_block_* slots and emulated_edge_up_{h,v} are overridden: the first
 * vertical-pass call that names an enhancement-layer picture since its registration / the last ohevc_tables_begin_frame
 * resamples the WHOLE base-layer picture into it on the device (ohevc_pic_upsample, block-slot rules); later calls for the same
 * picture find it done -- the reference's is_upsampled map (hevc.c:3221-3223) per picture instead of per CTB.  Both pictures must
 * be registered (the inter-layer reference picture is a DPB frame of its own: ff_hevc_set_new_iter_layer_ref, hevc.c:3236).
 * For builds that resample whole frames (upsample_base_layer_frame, hevc.c:3240-3242) the reference-side stub passes the two
 * luma plane pointers here. */
int  ohevc_tables_upsample_frame(const uint8_t *el_data0, const uint8_t *bl_data0, const ohevc_HEVCWindow *Enhscal, const ohevc_UpsamplInf *up_info);
/* Cross-component prediction (RExt 4:4:4).  hls_cross_component_pred (hevc.c:1186-1200) decodes lc->tu.res_scale_val right
 * before each chroma component of a transform unit; the chroma residual then gets (res_scale_val * luma residual) >> 3
 * added on the host (hevc_cabac.c:1942-1948, hevc.c:1315-1330) -- from a luma residual that does not exist behind recording
 * tables.  Call this with lc->tu.res_scale_val at the end of hls_cross_component_pred: the next chroma transform_add of the
 * calling thread becomes an OHEVC_TU_CROSS job (both coefficient blocks travel, the kernel forms both residuals). */
int  ohevc_tables_cross_component(int res_scale_val);
/* The in-loop filter drivers in bulk (host half).  Instead of letting deblocking_filter_CTB / sao_filter_CTB
 * (hevc_filter.c:197-581) make one table call per 8-sample edge and per CTB plane while the picture is parsed, skip
 * ff_hevc_hls_filters / ff_hevc_hls_filter altogether (keep their ff_thread_report_progress calls) and hand the arrays they read
 * over once, before ohevc_tables_end_frame: the same deblocking and SAO jobs are recorded in one pass.  All pointers are the
 * reference's own arrays, only read during the call:
 *   horizontal_bs / vertical_bs / bs_width   s->horizontal_bs, s->vertical_bs, s->bs_width   (boundary strengths: still derived
 *                                            on the CPU by ff_hevc_deblocking_boundary_strengths, which stays)
 *   qp_y_tab / min_cb_width                  s->qp_y_tab, sps->min_cb_width
 *   deblock / deblock_stride                 (const int8_t *)s->deblock, sizeof(DBParams): [0] beta_offset, [1] tc_offset per CTB
 *   sao                                      s->sao (SAOParams per CTB, raster order)
 *   filter_slice_edges, tab_slice_address    s->filter_slice_edges, s->tab_slice_address
 *   ctb_addr_rs_to_ts, tile_id               pps arrays
 *   is_pcm / min_pu_width / min_pu_height    s->is_pcm etc., read when pcm_or_bypass (the reference's `pcmf`, hevc_filter.c:363-365)
 *   emulate_filter_lag, ctb_addr_ts_to_rs    16x16-CTB streams with SAO: the reference's output depends on the ORDER of its driver calls
 *                                            (filter lag, OHEVC_SAO_LAG_*); with emulate_filter_lag != 0 the drivers' control flow is
 *                                            replayed CTB by CTB in decoding order (pps->ctb_addr_ts_to_rs) and the flags derived from
 *                                            it, exactly as the recording slots do; 0 = every SAO block reads the fully deblocked
 *                                            picture (H.265 8.7.3) */
typedef struct ohevc_filter_maps {
    int32_t width, height, log2_ctb_size, log2_min_cb_size, log2_min_pu_size, chroma_format_idc;
    int32_t cb_qp_offset, cr_qp_offset;                    /* pps->cb_qp_offset, pps->cr_qp_offset */
    int32_t sao_enabled, tiles_enabled, loop_filter_across_tiles, pcm_or_bypass;
    const uint8_t *horizontal_bs, *vertical_bs;
    int32_t bs_width;
    const int8_t *qp_y_tab;
    int32_t min_cb_width;
    const int8_t *deblock;
    int32_t deblock_stride;
    const ohevc_SAOParams *sao;
    const uint8_t *filter_slice_edges;
    const int32_t *tab_slice_address;
    const int *ctb_addr_rs_to_ts, *tile_id;
    const uint8_t *is_pcm;
    int32_t min_pu_width, min_pu_height;
    int32_t emulate_filter_lag;
    const int *ctb_addr_ts_to_rs;
    /* device-side boundary strengths (ohevc_tables_bs_wanted() != 0 for this picture and every call of ff_hevc_deblocking_boundary_strengths
     * replaced by ohevc_tables_bs_call): s->cbf_luma and - ohevc_tables_bs_wanted() == 1 - the motion field s->ref->tab_mvf (entry layout as
     * ohevc_bs_maps); == 2: tab_mvf NULL, the picture kept the motion of its MC jobs (ohevc_tables_keep_motion).  cbf_luma NULL:
     * horizontal_bs / vertical_bs above are the reference's own */
    const void *tab_mvf;
    int32_t mvf_stride, mvf_off_mv, mvf_off_poc, mvf_off_pred_flag, mvf_pred_flag_bytes;
    const uint8_t *cbf_luma;
    int32_t min_tb_width, min_tb_height, log2_min_tb_size;
} ohevc_filter_maps;
/* Will ohevc_tables_derive_filters derive the boundary strengths of a picture with this geometry on the device?  (Then the front end records
 * its ff_hevc_deblocking_boundary_strengths calls with ohevc_tables_bs_call instead of making them.)  No for record-only contexts, for the
 * per-edge host derivation (ohevc_debug_set_filters_on_device(0)), for 16x16 CTBs with SAO in 4:2:0 / 4:2:2 (the filter-lag replay reads the
 * host arrays) and when switched off (ohevc_debug_set_bs_on_device(0); environment OHEVC_DEVICE_BS=0). */
int  ohevc_tables_bs_wanted(ohevc_ctx *ctx, int log2_ctb_size, int sao_enabled, int chroma_format_idc, int emulate_filter_lag);
/* Returns 0 (no), 1 (from the motion field ohevc_filter_maps.tab_mvf hands over) or 2 (from the picture's own MC jobs: call
 * ohevc_tables_keep_motion(ctx, sps->log2_min_pu_size) right after ohevc_tables_begin_frame; OHEVC_DEVICE_BS=0|1|2 chooses, default 2). */
int  ohevc_tables_keep_motion(ohevc_ctx *ctx, int log2_min_pu_size);
int  ohevc_tables_bs_call(int x0, int y0, int log2_size, int flags);       /* records into the calling thread's bound context */
int  ohevc_tables_bs_calls(const ohevc_bs_call *calls, int n);                /* the same for calls the front end collected itself (copied) */
int  ohevc_tables_derive_filters(ohevc_ctx *ctx, const ohevc_filter_maps *maps);
/* the host planes registered for picture-store slot `slot` (tests: oracle/sw_exec.c executes recorded jobs on them) */
int  ohevc_tables_host_planes(ohevc_ctx *ctx, int slot, uint8_t *data[3], int linesize[3]);
/* sticky status of the recording since begin_frame: table slots return void, so failures surface here */
int  ohevc_tables_status(ohevc_ctx *ctx);
/* Reproduce the reference front-end's filter lag (ff_hevc_hls_filter / ff_hevc_hls_filters, hevc_filter.c:1027-1063):
 * with 16x16 CTBs in 4:2:0 its SAO copies a few chroma samples of the next CTB column before the horizontal edge through
 * them has been deblocked (OHEVC_SAO_LAG_* in ohevc_hip.h).  When enabled, the recording slots keep the order of the
 * SAO and horizontal-edge calls and flag exactly the SAO jobs whose neighbour edge was filtered later, which keeps the
 * output bit-identical with the reference decoder; disabled (default) every SAO job reads the fully deblocked picture
 * (H.265 8.7.3).  Call once after ohevc_tables_bind. */
int  ohevc_tables_emulate_filter_lag(ohevc_ctx *ctx, int enable);
/* drop everything registered for ctx (done automatically by ohevc_ctx_destroy) */
void ohevc_tables_forget(ohevc_ctx *ctx);

/* intra_pred[log2-2] cannot be mirrored as a bare table slot (it takes HEVCContext*, hevcpred.h:32): the reference-side
 * stub in INTEGRATION.md extracts the fields below from HEVCContext and calls this. */
int  ohevc_tables_intra_pred(const ohevc_intra_geom *geom, int x0, int y0, int log2_size, int c_idx, int mode,
                             int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right);
/* same for constrained_intra_pred streams: pass &s->ref->tab_mvf[0].pred_flag, sizeof(MvField), PF_INTRA and
 * s->sps->log2_min_pu_size (see ohevc_intra_make_job_cip) */
int  ohevc_tables_intra_pred_cip(const ohevc_intra_geom *geom, int log2_min_pu_size, const uint8_t *pred_flag,
                                 ptrdiff_t pred_flag_stride, int intra_value, int x0, int y0, int log2_size, int c_idx, int mode,
                                 int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right);

#ifdef __cplusplus
}
#endif
#endif /* OHEVC_TABLES_H */
