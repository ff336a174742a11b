/*
 * table_driver.c -- TEST INFRASTRUCTURE ONLY.  A miniature front-end that drives HEVCDSPContext / VideoDSPContext /
 * intra_pred the way the reference's own call sites do (luma_mc_uni/bi + chroma_mc_* hevc.c:1641-1949,
 * ff_hevc_hls_residual_coding hevc_cabac.c:1868-1949, hls_pcm_sample hevc.c:1587-1621, deblocking_filter_CTB /
 * sao_filter_CTB hevc_filter.c:197-581), using the REAL struct types from the reference headers.
 * The same op list is run (a) on the tables as the reference fills them and (b) on tables overridden through the
 * product's init hooks (ohevc_hevcdsp_init_hip / ohevc_videodsp_init_hip, passed in as function pointers so this file
 * never links the product); tests/test_tables_gpu.py then compares the two host frames.
 */
#include <stdlib.h>
#include <string.h>

#include "libavcodec/get_bits.h"
#include "libavcodec/hevc.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/hevcpred.h"
#include "libavcodec/videodsp.h"

#define OHX(name) ohref_##name
#include "oracle_api.h"

typedef struct drv_pic { uint8_t *data[3]; int32_t linesize[3]; } drv_pic;
typedef void (*dsp_hook)(void *dsp, int bd);
typedef int  (*intra_hook)(const void *geom, int x0, int y0, int log2, int c_idx, int mode, int bl, int l, int ul, int u, int ur);

enum { OP_MC = 0, OP_TU = 1, OP_INTRA = 2, OP_DBK = 3, OP_SAO = 4, OP_PCM = 5, OP_WORDS = 24 };

static int pel_idx(int w)
{
    switch (w) { case 2: return 0; case 4: return 1; case 6: return 2; case 8: return 3; case 12: return 4;
                 case 16: return 5; case 24: return 6; case 32: return 7; case 48: return 8; case 64: return 9; }
    return -1;
}

/* returns 0, or a negative number naming the op that could not be driven */
int ohref_drive_tables(int bd, int W, int H, drv_pic *cur, drv_pic *refs, int nrefs,
                       const int32_t *ops, int nops, const int16_t *coeffs, const uint8_t *pcm_bits,
                       dsp_hook hevcdsp_hook, dsp_hook videodsp_hook, intra_hook intra, const void *intra_geom)
{
    HEVCDSPContext dsp;
    VideoDSPContext vdsp;
    HEVCPredContext hpc;
    const int ps = bd > 8 ? 2 : 1, pixel_shift = bd > 8;
    DECLARE_ALIGNED(32, int16_t, tucoeffs)[32 * 32];
    DECLARE_ALIGNED(16, int16_t, tmp)[MAX_PB_SIZE * MAX_PB_SIZE];
    /* adjacent, exactly as the two members of HEVCLocalContext (hevc.h:1162-1163) */
    uint8_t *emu1 = aligned_alloc(32, 2 * (MAX_PB_SIZE + 7) * EDGE_EMU_BUFFER_STRIDE * 2 + 64);
    uint8_t *emu2 = emu1 + (MAX_PB_SIZE + 7) * EDGE_EMU_BUFFER_STRIDE * 2;
    drv_pic twin;                              /* the reference's sao_frame: a second full-size frame (hevc.c:369-385) */
    int sao_copied = 0, rc = 0;

    memset(&dsp, 0, sizeof(dsp)); memset(&vdsp, 0, sizeof(vdsp));
    ff_hevc_dsp_init(&dsp, bd);                /* set_sps, hevc.c:421-423 */
    ff_hevc_pred_init(&hpc, bd);
    ff_videodsp_init(&vdsp, bd);
    if (hevcdsp_hook) hevcdsp_hook(&dsp, bd);  /* <- the patch of INTEGRATION.md: same place as ff_hevcdsp_init_x86 */
    if (videodsp_hook) videodsp_hook(&vdsp, bd);
    for (int c = 0; c < 3; c++) {
        int h = c ? H / 2 : H;
        twin.linesize[c] = cur->linesize[c];
        twin.data[c] = (uint8_t *)malloc((size_t)twin.linesize[c] * (h + 2) + 64) + twin.linesize[c] + 32;
    }

    for (int k = 0; k < nops && !rc; k++) {
        const int32_t *o = ops + (size_t)k * OP_WORDS;
        const int c = o[1];
        uint8_t *plane = cur->data[c];
        const ptrdiff_t stride = cur->linesize[c];
        switch (o[0]) {
        case OP_MC: {      /* [0,plane,x,y,w,h,flags(1 bi,2 weighted),ref0,ref1,xoff0,yoff0,mx0,my0,xoff1,yoff1,mx1,my1,denom,wx0,wx1,ox0,ox1] */
            const int x = o[2], y = o[3], bw = o[4], bh = o[5], bi = o[6] & 1, wt = o[6] & 2, luma = c == 0;
            const int before = luma ? QPEL_EXTRA_BEFORE : EPEL_EXTRA_BEFORE, after = luma ? QPEL_EXTRA_AFTER : EPEL_EXTRA_AFTER;
            const int extra = luma ? QPEL_EXTRA : EPEL_EXTRA;
            const int pic_w = c ? W / 2 : W, pic_h = c ? H / 2 : H, idx = pel_idx(bw);
            uint8_t *dst = plane + (ptrdiff_t)y * stride + x * ps;
            uint8_t *src[2]; ptrdiff_t sstride[2];
            if (idx < 0) { rc = -(k + 1); break; }
            for (int r = 0; r < (bi ? 2 : 1); r++) {
                drv_pic *rp = &refs[o[7 + r]];
                const int x_off = o[9 + 4 * r], y_off = o[10 + 4 * r];
                uint8_t *emu = r ? emu2 : emu1;
                sstride[r] = rp->linesize[c];
                src[r] = rp->data[c] + (ptrdiff_t)y_off * sstride[r] + (ptrdiff_t)x_off * ps;
                /* the reference's own (asymmetric) emulation test, hevc.c:1660-1663 / 1812-1815 */
                if (x_off < before || y_off < after || x_off >= pic_w - bw - after || y_off >= pic_h - bh - after) {
                    const int emu_stride = EDGE_EMU_BUFFER_STRIDE << pixel_shift;
                    const int offset = before * sstride[r] + (before << pixel_shift);
                    const int buf_offset = before * emu_stride + (before << pixel_shift);
                    vdsp.emulated_edge_mc(emu, src[r] - offset, emu_stride, sstride[r], bw + extra, bh + extra,
                                          x_off - before, y_off - before, pic_w, pic_h);
                    src[r] = emu + buf_offset;
                    sstride[r] = emu_stride;
                }
            }
            {
                const int mx0 = o[11], my0 = o[12], mx1 = o[15], my1 = o[16];
                const int denom = o[17], wx0 = o[18], wx1 = o[19], ox0 = o[20], ox1 = o[21];
#define SLOT(tab, mx, my) (luma ? dsp.put_hevc_qpel##tab[idx][!!(my)][!!(mx)] : dsp.put_hevc_epel##tab[idx][!!(my)][!!(mx)])
                if (!bi && !wt)      SLOT(_uni, mx0, my0)(dst, stride, src[0], sstride[0], bh, mx0, my0, bw);
                else if (!bi)        SLOT(_uni_w, mx0, my0)(dst, stride, src[0], sstride[0], bh, denom, wx0, ox0, mx0, my0, bw);
                else {
                    SLOT(, mx0, my0)(tmp, MAX_PB_SIZE, src[0], sstride[0], bh, mx0, my0, bw);
                    if (!wt) SLOT(_bi, mx1, my1)(dst, stride, src[1], sstride[1], tmp, MAX_PB_SIZE, bh, mx1, my1, bw);
                    else     SLOT(_bi_w, mx1, my1)(dst, stride, src[1], sstride[1], tmp, MAX_PB_SIZE, bh, denom, wx0, wx1, ox0, ox1, mx1, my1, bw);
                }
#undef SLOT
            }
            break;
        }
        case OP_TU: {      /* [1,plane,x,y,log2,kind,coeff_off] */
            const int x = o[2], y = o[3], log2 = o[4], kind = o[5], n = 1 << log2;
            uint8_t *dst = plane + (ptrdiff_t)y * stride + x * ps;
            memcpy(tucoeffs, coeffs + o[6], n * n * sizeof(int16_t));
            switch (kind) {
            case OH_TU_IDCT: dsp.idct[log2 - 2](tucoeffs, n); break;
            case OH_TU_DC:   dsp.idct_dc[log2 - 2](tucoeffs); break;
            case OH_TU_DST4: dsp.idct_4x4_luma(tucoeffs); break;
            case OH_TU_SKIP: dsp.transform_skip(tucoeffs, log2); break;
            case OH_TU_SKIP_RDPCM_H: dsp.transform_skip(tucoeffs, log2); dsp.transform_rdpcm(tucoeffs, log2, 0); break;
            case OH_TU_SKIP_RDPCM_V: dsp.transform_skip(tucoeffs, log2); dsp.transform_rdpcm(tucoeffs, log2, 1); break;
            case OH_TU_BYPASS: break;
            case OH_TU_BYPASS_RDPCM_H: dsp.transform_rdpcm(tucoeffs, log2, 0); break;
            case OH_TU_BYPASS_RDPCM_V: dsp.transform_rdpcm(tucoeffs, log2, 1); break;
            default: rc = -(k + 1);
            }
            dsp.transform_add[log2 - 2](dst, tucoeffs, stride);
            break;
        }
        case OP_PCM: {     /* [5,plane,x,y,log2,pcm_bd,byte_off,nbytes] : hls_pcm_sample, hevc.c:1603-1621 */
            GetBitContext gb;
            const int n = 1 << o[4];
            if (init_get_bits(&gb, pcm_bits + o[6], o[7] * 8) < 0) { rc = -(k + 1); break; }
            dsp.put_pcm(plane + (ptrdiff_t)o[3] * stride + o[2] * ps, stride, n, n, &gb, o[5]);
            break;
        }
        case OP_INTRA: {   /* [2,c_idx,x0,y0,log2,mode,bl,l,ul,u,ur] (luma coordinates) */
            if (intra) {
                if (intra(intra_geom, o[2], o[3], o[4], c, o[5], o[6], o[7], o[8], o[9], o[10]) != 0) rc = -(k + 1);
            } else {
                oh_intra_pic pic;
                memset(&pic, 0, sizeof(pic));
                for (int i = 0; i < 3; i++) { pic.data[i] = cur->data[i]; pic.linesize[i] = cur->linesize[i]; }
                pic.width = W; pic.height = H; pic.chroma_format_idc = 1; pic.log2_ctb_size = 6; pic.log2_min_tb_size = 2;
                pic.log2_min_pu_size = 2; pic.strong_intra_smoothing = 1;
                ohref_intra_pred(bd, &pic, o[2], o[3], o[4], c, o[5], o[6], o[7], o[8], o[9], o[10]);
            }
            break;
        }
        case OP_DBK: {     /* [3,plane,x,y,vertical,beta,tc0,tc1,no_p0,no_p1,no_q0,no_q1] */
            int tc[2] = { o[6], o[7] };
            uint8_t no_p[2] = { o[8], o[9] }, no_q[2] = { o[10], o[11] };
            uint8_t *pix = plane + (ptrdiff_t)o[3] * stride + o[2] * ps;
            if (c == 0) (o[4] ? dsp.hevc_v_loop_filter_luma : dsp.hevc_h_loop_filter_luma)(pix, stride, o[5], tc, no_p, no_q);
            else        (o[4] ? dsp.hevc_v_loop_filter_chroma : dsp.hevc_h_loop_filter_chroma)(pix, stride, tc, no_p, no_q);
            break;
        }
        case OP_SAO: {     /* [4,plane,x,y,w,h,band,klass,b0,b1,b2,b3,ov0..ov4] : sao_filter_CTB, hevc_filter.c:269-315 */
            SAOParams sao;
            int borders[4] = { o[8], o[9], o[10], o[11] };
            uint8_t ve[2] = { 0, 0 }, he[2] = { 0, 0 }, de[4] = { 0, 0, 0, 0 };
            if (!sao_copied) {                         /* all CTBs are deblocked by now: one copy of the whole picture */
                for (int i = 0; i < 3; i++) {
                    const int h = i ? H / 2 : H, wb = (i ? W / 2 : W) * ps;
                    for (int yy = 0; yy < h; yy++) memcpy(twin.data[i] + (ptrdiff_t)yy * twin.linesize[i], cur->data[i] + (ptrdiff_t)yy * cur->linesize[i], wb);
                }
                sao_copied = 1;
            }
            memset(&sao, 0, sizeof(sao));
            for (int i = 0; i < 5; i++) sao.offset_val[c][i] = o[12 + i];
            sao.band_position[c] = o[7]; sao.eo_class[c] = o[7];
            {
                uint8_t *fr = plane + (ptrdiff_t)o[3] * stride + o[2] * ps;
                uint8_t *cp = twin.data[c] + (ptrdiff_t)o[3] * twin.linesize[c] + o[2] * ps;
                if (o[6]) dsp.sao_band_filter(fr, cp, stride, twin.linesize[c], &sao, borders, o[4], o[5], c);
                else      dsp.sao_edge_filter[0](fr, cp, stride, twin.linesize[c], &sao, borders, o[4], o[5], c, ve, he, de);
            }
            break;
        }
        default: rc = -(k + 1);
        }
    }
    for (int c = 0; c < 3; c++) free(twin.data[c] - twin.linesize[c] - 32);
    free(emu1);
    return rc;
}
