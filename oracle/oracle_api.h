/*
 * oracle_api.h -- TEST INFRASTRUCTURE ONLY.
 *
 * One C API, implemented twice:
 *   - ohref_*  (oracle/ref_shim.c)   : thin call-through into the REFERENCE's own function-pointer
 *                                      tables (HEVCDSPContext / HEVCPredContext) compiled unmodified from
 *                                      /root/reference  ->  oracle/_ref/libhevcref.so
 *   - ohor_*   (oracle/hevc_oracle.c): our plain-C restatement of the same algorithms
 *                                      ->  oracle/liboracle.so
 * tests/ pins ohor_* against ohref_* (and against tests/golden/ fixtures generated from ohref_*),
 * then uses ohor_* as the checker for the HIP path.  Nothing in the product library links this.
 *
 * Conventions: `bd` = bit depth (8, 9, 10, 12); pixels are uint8_t for bd == 8, uint16_t otherwise;
 * every stride is in BYTES like the reference's table signatures (libavcodec/hevcdsp.h:45-123).
 */
#ifndef OHEVC_ORACLE_API_H
#define OHEVC_ORACLE_API_H
#include <stddef.h>
#include <stdint.h>

#ifndef OHX
#error "define OHX(name) before including oracle_api.h"
#endif

/* residual kinds == the kernel-choice branches of ff_hevc_hls_residual_coding (hevc_cabac.c:1868-1949) */
enum {
    OH_TU_IDCT   = 0, /* idct[log2-2](coeffs, col_limit)            hevcdsp_template.c:279-321 */
    OH_TU_DC     = 1, /* idct_dc[log2-2](coeffs)                    hevcdsp_template.c:303-326 */
    OH_TU_DST4   = 2, /* idct_4x4_luma(coeffs)                      hevcdsp_template.c:170-203 */
    OH_TU_SKIP   = 3, /* transform_skip(coeffs, log2)               hevcdsp_template.c:139-163 */
    OH_TU_SKIP_RDPCM_H = 4, /* transform_skip then transform_rdpcm(mode 0)  :114-136          */
    OH_TU_SKIP_RDPCM_V = 5, /* transform_skip then transform_rdpcm(mode 1)                    */
    OH_TU_BYPASS = 6, /* cu_transquant_bypass: coefficients are the residual                   */
    OH_TU_BYPASS_RDPCM_H = 7,
    OH_TU_BYPASS_RDPCM_V = 8,
    OH_TU_NKINDS
};

/* MC variants == the five table families put_hevc_{qpel,epel}[_uni|_uni_w|_bi|_bi_w] (hevcdsp.h:68-95) */
enum { OH_MC_PUT = 0, OH_MC_UNI = 1, OH_MC_UNI_W = 2, OH_MC_BI = 3, OH_MC_BI_W = 4 };

int  OHX(available)(void);

/* in-place inverse transform of one dense N x N block (N = 1 << log2) -> residual */
void OHX(tu_residual)(int bd, int kind, int log2, int16_t *coeffs, int col_limit);
/* transform_add[log2-2](dst, res, stride)   hevcdsp_template.c:45-111 */
void OHX(transform_add)(int bd, int log2, uint8_t *dst, ptrdiff_t stride, int16_t *res);
/* n blocks: residual of a COPY of coeffs[i*N*N ..] added at plane + xy[2i+1]*stride + xy[2i]*sizeof(pixel) */
void OHX(tu_batch)(int bd, int kind, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                   ptrdiff_t stride, const int32_t *xy, int col_limit);
/* same split over `threads` host threads (blocks are independent); CPU-baseline timing helper */
void OHX(tu_batch_mt)(int bd, int kind, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                      ptrdiff_t stride, const int32_t *xy, int col_limit, int threads);

/* one MC table call.  luma != 0 -> qpel (mx,my in 1/4), else epel (mx,my in 1/8).
 * OH_MC_PUT writes int16 to (int16_t*)dst with dststride in ELEMENTS (reference: MAX_PB_SIZE);
 * the others write pixels with dststride in bytes.  src2/src2stride(elements) only for BI/BI_W.
 * Weighted argument order is the CALL order used by hevc.c:1767-1773,1940-1948:
 * (denom, wx0, wx1, ox0, ox1) where wx0/ox0 weight src2 (list0) and wx1/ox1 the filtered src. */
void OHX(mc)(int bd, int luma, int variant, uint8_t *dst, ptrdiff_t dststride,
             uint8_t *src, ptrdiff_t srcstride, int16_t *src2, ptrdiff_t src2stride,
             int height, int mx, int my, int width,
             int denom, int wx0, int wx1, int ox0, int ox1);

/* deblocking: vertical_edge != 0 -> hevc_v_loop_filter_* (filters across a vertical edge),
 * else hevc_h_loop_filter_*.  hevcdsp_template.c:1629-1787 */
void OHX(deblock_luma)(int bd, int vertical_edge, uint8_t *pix, ptrdiff_t stride, int beta,
                       int *tc, uint8_t *no_p, uint8_t *no_q);
void OHX(deblock_chroma)(int bd, int vertical_edge, uint8_t *pix, ptrdiff_t stride,
                         int *tc, uint8_t *no_p, uint8_t *no_q);

/* SAO.  hevcdsp_template.c:340-567.  offset_val[5] as SAOParams.offset_val[c_idx] (hevc.h:514-523) */
void OHX(sao_band)(int bd, uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst, ptrdiff_t stride_src,
                   const int16_t *offset_val, int band_position, int width, int height);
void OHX(sao_edge)(int bd, int restore, uint8_t *dst, uint8_t *src, ptrdiff_t stride_dst,
                   ptrdiff_t stride_src, const int16_t *offset_val, int eo_class, int *borders,
                   int width, int height, uint8_t *vert_edge, uint8_t *horiz_edge, uint8_t *diag_edge);

/* pure intra predictors.  hevcpred_template.c:359-537.  top/left point at element 0 with [-1] valid */
void OHX(pred_planar)(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride);
void OHX(pred_dc)(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int c_idx);
void OHX(pred_angular)(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left,
                       ptrdiff_t stride, int c_idx, int mode);

/* Full intra_pred() (hevcpred_template.c:30-357) on a picture described by `oh_intra_pic`.
 * The caller supplies the five HEVClc->na.cand_* flags (hevc_mvs.c:41-58) exactly as the reference's
 * front-end would; z-scan qualification (pps->min_tb_addr_zs) is done inside, as in the reference. */
typedef struct oh_intra_pic {
    uint8_t  *data[3];
    int32_t   linesize[3];          /* bytes */
    int32_t   width, height;        /* luma samples */
    int32_t   chroma_format_idc;    /* 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 */
    int32_t   log2_ctb_size, log2_min_tb_size, log2_min_pu_size;
    int32_t   strong_intra_smoothing, intra_smoothing_disabled, constrained_intra_pred;
    const uint8_t *is_intra;        /* per min-PU map, 1 = PF_INTRA; only read when constrained_intra_pred */
} oh_intra_pic;
void OHX(intra_pred)(int bd, const oh_intra_pic *pic, int x0, int y0, int log2, int c_idx, int mode,
                     int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right);

/* SHVC inter-layer up-sampling of a whole 4:2:0 picture = upsample_base_layer_frame (hevcdsp_template.c:2165-2438), which
 * the per-block slots upsample_filter_block_{luma,cr}_{h,v}[3] + emulated_edge_up_{h,v} reproduce CTB by CTB
 * (hevc_filter.c:1175-1310).  Only implemented by the restatement (ohor_); the reference side is oracle/shvc_driver.c. */
void OHX(shvc_upsample_frame)(int bd, int block_slots, uint8_t *const el[3], const int32_t el_stride[3], int el_w, int el_h,
                              uint8_t *const bl[3], const int32_t bl_stride[3], int bl_w, int bl_h, const int32_t *win, const int32_t *up);

/* Boundary strengths of the deblocking filter: ff_hevc_deblocking_boundary_strengths (hevc_filter.c:805-941) with boundary_strength()
 * (:584-700, the C branch; TEST_MV_POC build: the motion field carries the POCs of its references).  One oh_bs_call per call the reference's
 * front end makes (hevc.c:1578,1607,2400,2484); flags: bit 0/1 = lc->slice_or_tiles_up_boundary, bit 2/3 = ..._left_boundary, bit 4 =
 * sh.slice_loop_filter_across_slices_enabled_flag.  The two arrays are only written where a call writes them (zero them first, hevc.c:3207-3208).
 * Only implemented by the restatement (ohor_); pinned against the reference's own arrays through oracle/null_hooks.c's tap. */
typedef struct oh_bs_field { int16_t mv[2][2]; int32_t poc[2]; uint32_t pred_flag; } oh_bs_field;       /* pred_flag: 0 intra, 1 L0, 2 L1, 3 both */
typedef struct oh_bs_call { uint16_t x0, y0; uint8_t log2_size, flags; uint16_t reserved; } oh_bs_call;
typedef struct oh_bs_geom { int32_t min_pu_width, log2_min_pu_size, min_tb_width, log2_min_tb_size, log2_ctb_size, bs_width, loop_filter_across_tiles; } oh_bs_geom;
void OHX(boundary_strengths)(const oh_bs_geom *g, const oh_bs_field *mvf, const uint8_t *cbf_luma, const oh_bs_call *calls, int ncalls,
                             uint8_t *vertical_bs, uint8_t *horizontal_bs);

#endif
