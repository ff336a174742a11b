/*
 * shvc_driver.c -- TEST INFRASTRUCTURE ONLY.  Drives the 13 SHVC inter-layer up-sampling slots of HEVCDSPContext
 * (hevcdsp.h:106-123) and the two edge helpers of VideoDSPContext (videodsp.h, emulated_edge_up_h / _v) the way the
 * reference's own call sites do, with the REAL struct types from the reference headers:
 *   - ohref_shvc_frame : hevc_frame_start, hevc.c:3240-3242 -- upsample_base_layer_frame over a whole picture;
 *   - ohref_shvc_blocks: upsample_block_luma / upsample_block_mc, hevc_filter.c:1175-1310, for every CTB of the
 *                        enhancement-layer picture (what ff_upsample_block, :1370-1395, triggers CTB by CTB).
 * Like table_driver.c the sequences run either on the tables as the reference fills them or on tables overridden through
 * the product's init hooks (function pointers: this file never links the product).
 */
#include <stdlib.h>
#include <string.h>

#include "libavcodec/get_bits.h"
#include "libavcodec/hevc.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/videodsp.h"
#include "libavutil/frame.h"

typedef struct drv_pic { uint8_t *data[3]; int32_t linesize[3]; } drv_pic;
typedef void (*dsp_hook)(void *dsp, int bd);

static void fill_tables(HEVCDSPContext *dsp, VideoDSPContext *vdsp, int bd, dsp_hook hevcdsp_hook, dsp_hook videodsp_hook)
{
    memset(dsp, 0, sizeof(*dsp)); memset(vdsp, 0, sizeof(*vdsp));
    ff_hevc_dsp_init(dsp, bd);                 /* set_sps, hevc.c:421-423 */
    ff_videodsp_init(vdsp, bd);
    if (hevcdsp_hook) hevcdsp_hook(dsp, bd);   /* <- the patch of INTEGRATION.md */
    if (videodsp_hook) videodsp_hook(vdsp, bd);
}

static void fill_frame(AVFrame *f, const drv_pic *p, int w, int h)
{
    memset(f, 0, sizeof(*f));
    for (int c = 0; c < 3; c++) { f->data[c] = p->data[c]; f->linesize[c] = p->linesize[c]; }
    f->width = f->coded_width = w; f->height = f->coded_height = h;
}

/* the reference-side stub for the one SHVC slot that takes AVFrames */
typedef int (*frame_helper_fn)(const uint8_t *el_data0, const uint8_t *bl_data0, const HEVCWindow *w, const UpsamplInf *u);
static __thread frame_helper_fn t_frame_helper;
static void upsample_base_layer_frame_stub(struct AVFrame *FrameEL, struct AVFrame *FrameBL, short *Buffer[3], const struct HEVCWindow *Enhscal,
                                           struct UpsamplInf *up_info, int channel)
{
    (void)Buffer; (void)channel;
    t_frame_helper(FrameEL->data[0], FrameBL->data[0], Enhscal, up_info);
}

/* win = scaled_ref_layer_window {left, right, top, bottom}; up = UpsamplInf {addXLum, addYLum, scaleXLum, scaleYLum,
 * addXCr, addYCr, scaleXCr, scaleYCr, idx} (hevc.h:347-357) */
int ohref_shvc_frame(int bd, drv_pic *el, int el_w, int el_h, drv_pic *bl, int bl_w, int bl_h, const int32_t *win, const int32_t *up,
                     dsp_hook hevcdsp_hook, dsp_hook videodsp_hook, frame_helper_fn frame_helper)
{
    HEVCDSPContext dsp;
    VideoDSPContext vdsp;
    AVFrame fel, fbl;
    HEVCWindow w = { win[0], win[1], win[2], win[3] };
    UpsamplInf u = { up[0], up[1], up[2], up[3], up[4], up[5], up[6], up[7], up[8] };
    short *buf[3];
    fill_tables(&dsp, &vdsp, bd, hevcdsp_hook, videodsp_hook);
    fill_frame(&fel, el, el_w, el_h);
    fill_frame(&fbl, bl, bl_w, bl_h);
    if (frame_helper) { t_frame_helper = frame_helper; dsp.upsample_base_layer_frame = upsample_base_layer_frame_stub; }
    /* s->buffer_frame[]: pic_arrays_init allocates width * height shorts per plane (hevc.c:197-204) */
    for (int c = 0; c < 3; c++) buf[c] = calloc((size_t)el_w * (el_h > bl_h ? el_h : bl_h) + 64, sizeof(short));
    dsp.upsample_base_layer_frame(&fel, &fbl, buf, &w, &u, 1);
    for (int c = 0; c < 3; c++) free(buf[c]);
    return 0;
}

#define MAX_EDGE_BUFFER_STRIDE_ ((MAX_PB_SIZE + 20) * 2)      /* hevc.h:98 */

/* conf = the enhancement layer's pic_conf_win {left, top} (the block path derives the base-layer position from it) */
int ohref_shvc_blocks(int bd, int log2_ctb, drv_pic *el, int el_width, int el_height, drv_pic *bl, int bl_width_, int bl_height_,
                      const int32_t *win, const int32_t *conf, const int32_t *up, dsp_hook hevcdsp_hook, dsp_hook videodsp_hook)
{
    HEVCDSPContext dsp;
    VideoDSPContext vdsp;
    HEVCWindow w = { win[0], win[1], win[2], win[3] };
    UpsamplInf u = { up[0], up[1], up[2], up[3], up[4], up[5], up[6], up[7], up[8] };
    const int ps = bd > 8 ? 2 : 1;
    int16_t *edge_emu_buffer_up_v = calloc(MAX_EDGE_BUFFER_SIZE + 64, sizeof(int16_t));     /* hevc.h:1164 */
    /* OHREF_SHVC_POISON=v: the scratch buffer starts out filled with v instead of zeros.  In the decoder it is a member of the thread's
     * HEVCLocalContext that every CTB reuses; a result that changes with v was computed from rows no slot call of that CTB wrote. */
    if (getenv("OHREF_SHVC_POISON"))
        for (int i = 0; i < MAX_EDGE_BUFFER_SIZE + 64; i++) edge_emu_buffer_up_v[i] = (int16_t)atoi(getenv("OHREF_SHVC_POISON"));
    fill_tables(&dsp, &vdsp, bd, hevcdsp_hook, videodsp_hook);
    if (u.idx == SNR) { free(edge_emu_buffer_up_v); return -1; }    /* x1 (SNR) scalability is a plain copy_block, hevc_filter.c:1187-1190 */

    for (int y0 = 0; y0 < el_height; y0 += 1 << log2_ctb)
        for (int x0 = 0; x0 < el_width; x0 += 1 << log2_ctb) {
            /* OHREF_SHVC_POISON_EACH=1: ... and is refilled in front of every CTB.  The decoder up-samples CTBs on demand, in the order
             * motion compensation happens to reference them (ff_upsample_block, hevc_filter.c:1370-1395): what an earlier CTB left
             * in the buffer is no more a function of the stream than its initial contents. */
            if (getenv("OHREF_SHVC_POISON") && getenv("OHREF_SHVC_POISON_EACH"))
                for (int i = 0; i < MAX_EDGE_BUFFER_SIZE + 64; i++) edge_emu_buffer_up_v[i] = (int16_t)atoi(getenv("OHREF_SHVC_POISON"));
            /* ---- upsample_block_luma, hevc_filter.c:1175-1238 */
            {
                uint8_t *src, *dst = el->data[0];
                int ctb_size = 1 << log2_ctb;
                int bl_width = bl_width_, bl_height = bl_height_, bl_stride = bl->linesize[0];
                int ePbW = x0 + ctb_size > el_width ? el_width - x0 : ctb_size;
                int ePbH = y0 + ctb_size > el_height ? el_height - y0 : ctb_size;
                int bl_edge_bottom, bl_edge_right, ret;
                int bPbW = (((ePbW + 1) * u.scaleXLum + u.addXLum) >> 12) >> 4;
                int bPbH = (((ePbH + 2) * u.scaleYLum + u.addYLum) >> 12) >> 4;
                int bl_x = (((x0 - conf[0]) * u.scaleXLum + u.addXLum) >> 12) >> 4;
                int bl_y = (((y0 - conf[1]) * u.scaleYLum + u.addYLum) >> 12) >> 4;
                int bl_edge_left = (MAX_EDGE - 1 - bl_x) > 0 ? 0 : MAX_EDGE - 1;
                int bl_edge_top  = (MAX_EDGE - 1 - bl_y) > 0 ? 0 : MAX_EDGE - 1;
                int16_t *tmp0;
                if (bl_x + bPbW > bl_width)  bPbW = bl_width - bl_x;
                if (bl_y + bPbH > bl_height) bPbH = bl_height - bl_y;
                bl_edge_right  = (MAX_EDGE > (bl_width  - bl_x - bPbW)) ? bl_width  - bl_x - bPbW : MAX_EDGE;
                bl_edge_bottom = (MAX_EDGE > (bl_height - bl_y - bPbH)) ? bl_height - bl_y - bPbH : MAX_EDGE;
                src = bl->data[0] + (bl_y - bl_edge_top) * bl_stride + (bl_x - bl_edge_left) * ps;
                ret = vdsp.emulated_edge_up_h(src, bl_stride, &w, bPbW + bl_edge_left + bl_edge_right, bPbH + bl_edge_top + bl_edge_bottom,
                                              bl_edge_left, bl_edge_right, MAX_EDGE - 1);
                if (ret) src += (MAX_EDGE - 1) * ps;
                tmp0 = edge_emu_buffer_up_v + ((MAX_EDGE - 1) * MAX_EDGE_BUFFER_STRIDE_);
                dsp.upsample_filter_block_luma_h[u.idx](tmp0, MAX_EDGE_BUFFER_STRIDE_, src, bl_stride, x0, bl_x, ePbW,
                                                        bPbH + bl_edge_top + bl_edge_bottom, el_width, &w, &u);
                ret = vdsp.emulated_edge_up_v(tmp0, MAX_EDGE_BUFFER_STRIDE_, &w, ePbW, bPbH + bl_edge_top + bl_edge_bottom, x0, bl_edge_top,
                                              bl_edge_bottom, el_width, MAX_EDGE - 1);
                if (ret) tmp0 += ((MAX_EDGE - 1) * MAX_EDGE_BUFFER_STRIDE_);
                dsp.upsample_filter_block_luma_v[u.idx](dst, el->linesize[0], tmp0, MAX_EDGE_BUFFER_STRIDE_, bl_y, x0, y0, ePbW, ePbH,
                                                        el_width, el_height, &w, &u);
            }
            /* ---- upsample_block_mc, hevc_filter.c:1240-1310 (called with the CTB origin >> 1, :1381-1394) */
            {
                int x0c = x0 >> 1, y0c = y0 >> 1;
                uint8_t *src;
                int16_t *tmp0;
                int el_w = el_width >> 1, el_h = el_height >> 1;
                int bl_width  = bl_width_ >> 1;
                int bl_height = bl_height_ > el_h ? bl_height_ >> 1 : el_h >> 1;
                int ret, cr, bl_edge_top0;
                int ctb_size = 1 << (log2_ctb - 1);
                int ePbW = x0c + ctb_size > el_w ? el_w - x0c : ctb_size;
                int ePbH = y0c + ctb_size > el_h ? el_h - y0c : ctb_size;
                int bl_stride = bl->linesize[1], el_stride = el->linesize[1];
                int bl_edge_right, bl_edge_bottom;
                int bPbW = (((ePbW + 1) * u.scaleXLum + u.addXLum) >> 12) >> 4;
                int bPbH = (((ePbH + 2) * u.scaleYLum + u.addYLum) >> 12) >> 4;
                int bl_x = (((x0c - (conf[0] >> 1)) * u.scaleXLum + u.addXLum) >> 12) >> 4;
                int bl_y = ((((y0c - (conf[1] >> 1)) * u.scaleYLum + u.addYLum) >> 12) - 4) >> 4;
                int bl_edge_left = (MAX_EDGE_CR - 1 - bl_x) > 0 ? 0 : MAX_EDGE_CR - 1;
                int bl_edge_top  = (MAX_EDGE_CR - 1 - bl_y) > 0 ? 0 : MAX_EDGE_CR - 1;
                bPbW = bl_x + bPbW > bl_width  ? bl_width  - bl_x : bPbW;
                bPbH = bl_y + bPbH > bl_height ? bl_height - bl_y : bPbH;
                bl_edge_top0 = bl_y < 0 ? bl_y : 0;
                bl_edge_right  = MAX_EDGE_CR < (bl_width  - bl_x - bPbW) ? MAX_EDGE_CR : bl_width  - bl_x - bPbW;
                bl_edge_bottom = MAX_EDGE_CR < (bl_height - bl_y - bPbH) ? MAX_EDGE_CR : bl_height - bl_y - bPbH;
                for (cr = 1; cr <= 2; cr++) {
                    /* OHREF_SHVC_POISON_EACH=2: ... and in front of each chroma plane as well.  The decoder resamples a CTB's chroma BEFORE
                     * its luma (ff_upsample_block, hevc_filter.c:1381-1395), this driver after it: a chroma row read from scratch memory
                     * comes from the luma pass of ANOTHER CTB there, from this CTB's here - and from the poison with this setting */
                    if (getenv("OHREF_SHVC_POISON") && getenv("OHREF_SHVC_POISON_EACH") && atoi(getenv("OHREF_SHVC_POISON_EACH")) >= 2)
                        for (int i = 0; i < MAX_EDGE_BUFFER_SIZE + 64; i++) edge_emu_buffer_up_v[i] = (int16_t)atoi(getenv("OHREF_SHVC_POISON"));
                    src = bl->data[cr] + (bl_y - bl_edge_top) * bl_stride + (bl_x - bl_edge_left) * ps;
                    ret = vdsp.emulated_edge_up_h(src, bl_stride, &w, bPbW + bl_edge_left + bl_edge_right, bPbH + bl_edge_top + bl_edge_bottom,
                                                  bl_edge_left, bl_edge_right, MAX_EDGE_CR - 1);
                    if (ret) src += (MAX_EDGE_CR - 1) * ps;
                    tmp0 = edge_emu_buffer_up_v + ((MAX_EDGE_CR - 1) * MAX_EDGE_BUFFER_STRIDE_);
                    dsp.upsample_filter_block_cr_h[u.idx](tmp0, MAX_EDGE_BUFFER_STRIDE_, src, bl_stride, x0c, bl_x, ePbW,
                                                          bPbH + bl_edge_top + bl_edge_bottom, el_w, &w, &u);
                    ret = vdsp.emulated_edge_up_v(tmp0, MAX_EDGE_BUFFER_STRIDE_, &w, ePbW, bPbH + bl_edge_top + bl_edge_bottom, x0c,
                                                  bl_edge_top + bl_edge_top0, bl_edge_bottom, el_w, MAX_EDGE_CR - 1);
                    if (ret) tmp0 += ((MAX_EDGE_CR - 1) * MAX_EDGE_BUFFER_STRIDE_);
                    dsp.upsample_filter_block_cr_v[u.idx](el->data[cr], el_stride, tmp0, MAX_EDGE_BUFFER_STRIDE_, bl_y, x0c, y0c, ePbW, ePbH,
                                                          el_w, el_h, &w, &u);
                }
            }
        }
    free(edge_emu_buffer_up_v);
    return 0;
}
