/*
 * ohevc_hip.h -- C ABI of libohevc_hip.so: openHEVC's per-CTU pixel-reconstruction hot path on MI355X.
 *
 * The reference hides this path behind two function-pointer tables, HEVCDSPContext
 * (libavcodec/hevcdsp.h:44-124) and HEVCPredContext (libavcodec/hevcpred.h:31-41), whose slots are
 * called one block at a time from the CPU decoder (hevc.c / hevc_cabac.c / hevc_filter.c).  A GPU cannot
 * be driven one 4x4 block at a time, so this library exposes the same kernels in BATCHED form: the host
 * records one fixed-size job record per table call it would have made and ships a whole CTU row / frame
 * of records at once.  Every entry point below names the table slot(s) it replaces.
 *
 * Two layers:
 *   ohevc_dev_*   device-resident batched kernels (this file, section 2).  All pointers are DEVICE
 *                 pointers unless said otherwise; `stream` is a hipStream_t passed as void* (NULL = the
 *                 default stream).  Launches are asynchronous; errors are returned as negative codes.
 *   ohevc_ctx_*   host-side job recorder + device frame store (include/ohevc_ctx.h).
 *   The drop-in table fillers ohevc_hevcdsp_init_hip()/ohevc_hevcpred_init_hip() live in
 *   include/ohevc_tables.h (the hook the reference calls at hevcdsp.c:1326-1327 / hevcpred.c:84).
 *
 * Plain C, no C++/torch types in any signature.  Pixels are uint8_t for bit_depth == 8 and uint16_t
 * for 9..12 (bit_depth_template.c:50-88).  Strides are in BYTES, like the reference's.
 */
#ifndef OHEVC_HIP_H
#define OHEVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ 1. common types */
enum {
    OHEVC_OK          =  0,
    OHEVC_ERR_ARG     = -1,   /* bad argument (size, alignment, bit depth, kind)   */
    OHEVC_ERR_HIP     = -2,   /* a HIP runtime call failed; see ohevc_last_error() */
    OHEVC_ERR_NODEV   = -3,   /* no gfx950 device visible                          */
    OHEVC_ERR_STATE   = -4    /* call sequence error (ctx layer)                   */
};

/* One picture plane in HBM.  For ohevc_dev_tu_batch data must be 16-byte aligned and stride a multiple of 16 bytes
 * (what the reference's STRIDE_ALIGN frame buffers give, libavcodec/utils.c:432); the other kernels only need
 * natural pixel alignment. */
typedef struct ohevc_plane {
    void    *data;
    int32_t  stride;          /* bytes */
    int32_t  width, height;   /* samples */
} ohevc_plane;

/* ------------------------------------------------------------------ 2. batched device kernels */

/* ---- 2.1 residual (TU) family: replaces the call sequence
 *   idct[log2-2] | idct_dc[log2-2] | idct_4x4_luma | transform_skip [+ transform_rdpcm]   (in place on coeffs)
 *   followed by transform_add[log2-2](dst, coeffs, stride)
 * of ff_hevc_hls_residual_coding (hevc_cabac.c:1868-1949; slots hevcdsp.h:48-58). */
enum {
    OHEVC_TU_IDCT = 0, OHEVC_TU_DC = 1, OHEVC_TU_DST4 = 2, OHEVC_TU_SKIP = 3,
    OHEVC_TU_SKIP_RDPCM_H = 4, OHEVC_TU_SKIP_RDPCM_V = 5,
    OHEVC_TU_BYPASS = 6, OHEVC_TU_BYPASS_RDPCM_H = 7, OHEVC_TU_BYPASS_RDPCM_V = 8,
    OHEVC_TU_PCM = 9,     /* put_pcm (hevcdsp_template.c:30-43): the arena block holds the N*N samples the host read from
                             the bitstream, already << (bit_depth - pcm_bit_depth); they REPLACE the block */
    OHEVC_TU_CROSS = 10,  /* cross-component prediction (RExt 4:4:4; hevc.c:1291-1365, hevc_cabac.c:1942-1949): a CHROMA block whose
                             residual is (int16)(rC + ((res_scale_val * rY) >> 3)), rY = residual of the transform unit's luma block,
                             rC = the block's own residual.  The job carries both: coeff_off = chroma block, reserved1 = luma block
                             (dense N*N int16 each, also for DC kinds, which read [0]), reserved0 = chroma kind | luma kind << 4
                             (chroma kind 15 = no coded coefficients, rC = 0), dc = res_scale_val (+-1, 2, 4, 8) */
    OHEVC_TU_NKINDS = 11
};

typedef struct ohevc_tu_job {           /* 16 bytes */
    uint16_t x, y;                      /* top-left sample of the block inside its plane; multiples of the block size */
    uint8_t  plane;                     /* index into planes[3] */
    uint8_t  reserved0;                 /* OHEVC_TU_CROSS: residual kinds, see the enum */
    int16_t  dc;                        /* OHEVC_TU_DC: coeffs[0] (no arena storage needed); OHEVC_TU_CROSS: res_scale_val */
    uint32_t coeff_off;                 /* offset of the dense N*N int16 block in the coefficient arena, in int16
                                           units; must be a multiple of 8 (16 bytes).  Ignored for OHEVC_TU_DC */
    uint32_t reserved1;                 /* OHEVC_TU_CROSS: offset of the luma block in the arena (int16 units, multiple of 8) */
} ohevc_tu_job;

/* Runs `njobs` blocks of one size (1 << log2_size, 2..5) and one residual kind.  Blocks of a batch must not
 * overlap (they never do inside a frame).  col_limit of the reference's idct is not needed: a full transform is
 * result-identical on the inputs the decoder can produce (hevc_cabac.c:1923-1934, x86/hevc_idct_sse.c:504). */
int ohevc_dev_tu_batch(const ohevc_plane planes[3], int bit_depth, int log2_size, int kind,
                       const ohevc_tu_job *jobs, int njobs, const int16_t *coeffs, void *stream);

/* Same work as several ohevc_dev_tu_batch calls in ONE launch: `segs` (HOST array, <= 40 entries) cuts the job array
 * into runs of equal (log2_size, kind).  Blocks of different segments must not overlap either. */
typedef struct ohevc_tu_segment {
    int32_t log2_size, kind;
    int32_t first_job, njobs;           /* run inside `jobs` */
} ohevc_tu_segment;
int ohevc_dev_tu_multi(const ohevc_plane planes[3], int bit_depth, const ohevc_tu_segment *segs, int nsegs,
                       const ohevc_tu_job *jobs, const int16_t *coeffs, void *stream);

/* ---- 2.2 motion compensation: replaces, per prediction block, the wrappers luma_mc_uni/bi, chroma_mc_uni/bi
 * (hevc.c:1641-1949) together with the table slots they call:
 *   put_hevc_{qpel,epel}[idx][!!my][!!mx]            (first half of bi-pred, 14-bit intermediate)   hevcdsp.h:68,81
 *   put_hevc_{qpel,epel}_{uni,uni_w,bi,bi_w}[..]      hevcdsp.h:70-95
 * and vdsp.emulated_edge_mc (videodsp_template.c:26-100): reference coordinates are clamped to the picture
 * inside the kernel instead of copying a padded window.  Bi-prediction reads both references in one job, so the
 * int16 tmp[64*64] round trip of hevc.c:1761-1764 never touches memory. */
enum { OHEVC_MC_BI = 1, OHEVC_MC_WEIGHTED = 2 };

typedef struct ohevc_mc_job {           /* 32 bytes */
    uint16_t x, y;                      /* destination block position in its plane (samples) */
    uint8_t  w, h;                      /* block size in samples: w in {2,4,6,8,12,16,24,32,48,64}, h <= 64 */
    uint8_t  plane;                     /* 0: luma, 8-tap quarter-sample; 1,2: chroma, 4-tap eighth-sample */
    uint8_t  flags;                     /* OHEVC_MC_BI | OHEVC_MC_WEIGHTED */
    int16_t  sx0, sy0;                  /* integer sample position of the block in reference 0: x_off + (mv.x >> 2)
                                           (chroma: >> (2 + hshift)); may lie outside the picture */
    int16_t  sx1, sy1;                  /* same for reference 1 (bi only) */
    uint8_t  mx0, my0, mx1, my1;        /* fractional phase: luma 0..3, chroma 0..7 (the reference's _mx/_my) */
    int8_t   ref0, ref1;                /* slot of the reference picture in the `refs` table */
    uint8_t  denom;                     /* luma/chroma_log2_weight_denom */
    uint8_t  reserved;
    int16_t  wx0, wx1;                  /* weights: uni uses (wx0, ox0); bi: wx0/ox0 weight reference 0 (list 0,
                                           the int16 "src2" of the reference), wx1/ox1 reference 1 */
    int16_t  ox0, ox1;
} ohevc_mc_job;

/* Coefficients cross the bus COMPACT and are expanded on the device into the dense arena the TU jobs index (coeff_off): of an inverse-DCT
 * block only the top-left cols x rows rectangle that can hold non-zero coefficients travels - the bound the reference computes from the last
 * significant coefficient and hands its idct slot as col_limit (hevc_cabac.c:1923-1934; its C transforms skip the rest,
 * hevcdsp_template.c:271-277,288-291).  One record per piece of the compact stream, offsets and sizes in int16 elements:
 *   kind 0     : dims elements copied as they are (whole blocks; dims a multiple of 16, at most 1024);
 *   kind 3,4,5 : an (1 << kind)-sample block; its compact form is rows rows of cols coefficients (dims = cols | rows << 8, multiples of 4),
 *                everything else of the block is written as zero;
 *   kind 0x100 | log2 | part << 9 (log2 3..5): the SUB-BLOCK form - only the 4x4 coefficient groups that hold a non-zero coefficient travel (the
 *                coded sub-blocks of residual coding, hevc_cabac.c:1832-1841: at qp22-like density 42 % of a picture's coefficient volume where
 *                the rectangles keep 61 %), 16 elements each (4 rows of 4), in raster order of the groups; dims = one bit per group of the
 *                record's region, bit gy * (N / 4) + gx.  part 0: the whole block (log2 3, 4) / rows 0..15 of a 32x32 block, 1: rows 16..31 of a
 *                32x32 block (dst = the block's offset + 512), 2: a whole 32x32 block whose rows 16..31 are zero (the mask covers rows 0..15).
 *                Groups without a bit are written as zero.
 * src: offset in the compact stream (a multiple of 4), dst: offset in the dense arena (a multiple of 16). */
typedef struct ohevc_expand_rec { uint32_t src, dst, dims, kind; } ohevc_expand_rec;
int ohevc_dev_expand_coeffs(const int16_t *compact, const ohevc_expand_rec *recs, int nrecs, int16_t *dense, void *stream);

/* dst: the 3 planes of the picture being reconstructed.  refs: DEVICE array of n_ref_slots * 3 ohevc_plane
 * (slot-major: refs[3 * slot + plane]); width/height there are the picture size used for clamping. */
int ohevc_dev_mc_batch(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                       const ohevc_mc_job *jobs, int njobs, void *stream);
/* Same contract plus a promise: no job of the batch is wider than max_w or taller than max_h (1..64).  The kernel's unit of work is a
 * 16x16 tile; with the bound it knows how many tiles a job can have and spreads them over wavefronts (ohevc_dev_mc_batch assumes
 * 64x64).  Jobs that violate the bound are only partly written. */
int ohevc_dev_mc_batch_bounded(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                               const ohevc_mc_job *jobs, int njobs, int max_w, int max_h, void *stream);
/* Same contract for batches in which EVERY job has w <= 8 and h <= 8 (most chroma blocks of 4:2:0 content): four jobs share
 * one wavefront.  Jobs that violate the size limit produce undefined pixels inside their own block. */
int ohevc_dev_mc_batch_small(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                             const ohevc_mc_job *jobs, int njobs, void *stream);

/* ---- 2.3 deblocking: replaces hevc_{h,v}_loop_filter_{luma,chroma}[_c] (hevcdsp.h:97-104;
 * hevcdsp_template.c:1629-1787).  One job = one table call = one 8-sample edge (two 4-line segments).
 * The caller orders passes like deblocking_filter_CTB (hevc_filter.c:345-581): all vertical edges of a
 * picture (one launch), then all horizontal edges (another launch). */
enum { OHEVC_DBK_VERTICAL_EDGE = 1, OHEVC_DBK_NO_P0 = 2, OHEVC_DBK_NO_P1 = 4, OHEVC_DBK_NO_Q0 = 8, OHEVC_DBK_NO_Q1 = 16 };

typedef struct ohevc_dbk_job {          /* 16 bytes */
    uint16_t x, y;                      /* q0 sample of the first line of the edge */
    uint8_t  plane;                     /* 0: luma filter; 1,2: chroma filter */
    uint8_t  flags;                     /* OHEVC_DBK_* */
    uint8_t  beta;                      /* luma only (pre bit-depth scaling) */
    uint8_t  reserved0;
    int16_t  tc[2];                     /* per 4-line segment (pre bit-depth scaling) */
    uint32_t reserved1;
} ohevc_dbk_job;

int ohevc_dev_deblock_batch(const ohevc_plane planes[3], int bit_depth, const ohevc_dbk_job *jobs, int njobs,
                            void *stream);

/* Deblocking straight from the decoder's maps: the device derives what deblocking_filter_CTB (hevc_filter.c:345-581)
 * derives per edge - which edges exist (boundary strengths), the QP average (get_qPy :144-150), beta / tc (tables :50-60, TC_CALC
 * :340-343, chroma_tc :62-89), the pcm / bypass flags (get_pcm :325-338) - and filters; no per-edge host work, no job list.
 * Pointers are DEVICE pointers for ohevc_dev_deblock_maps and HOST pointers for ohevc_rec_deblock_maps (ohevc_ctx.h), which copies
 * and uploads them.  Array sizes as the reference allocates them (hevc.c:123-171):
 *   vertical_bs   bs_width * ((height >> 2) + 4 * (1 << vshift))      bs_width = width >> 2
 *   horizontal_bs (bs_width + 4 * (1 << hshift)) * (height >> 2)
 *   qp_y_tab      min_cb_width * (height >> log2_min_cb_size)         int8
 *   deblock       DBParams { int8 beta_offset, tc_offset } per CTB, deblock_stride bytes apart
 *   is_pcm        min_pu_width * min_pu_height, or NULL when neither pcm loop-filter bypass nor transquant bypass is in use */
typedef struct ohevc_dbk_maps {
    const uint8_t *vertical_bs, *horizontal_bs;
    const int8_t  *qp_y_tab;
    const int8_t  *deblock;
    const uint8_t *is_pcm;
    int32_t bs_width, min_cb_width, deblock_stride, min_pu_width, min_pu_height;
    int32_t width, height, log2_ctb_size, log2_min_cb_size, log2_min_pu_size, chroma_format_idc;
    int32_t cb_qp_offset, cr_qp_offset;
} ohevc_dbk_maps;
/* one pass: vertical != 0 all vertical edges (luma and chroma), else all horizontal edges; the caller runs both, in that order */
int ohevc_dev_deblock_maps(const ohevc_plane planes[3], int bit_depth, const ohevc_dbk_maps *maps, int vertical, void *stream);

/* Boundary strengths on the device (second half): what ff_hevc_deblocking_boundary_strengths (hevc_filter.c:805-941, with
 * boundary_strength :584-700) computes per transform unit on the host - 15-18 % of the front end's time on 1080p inter content
 * (profiles/r03cpu_boundary_strength_share.jsonl) - from the decoder's motion field.  The host records one ohevc_bs_call per call of that
 * function (its arguments and the slice / tile flags of its CTB) and hands over HEVCFrame.tab_mvf and s->cbf_luma as they are at the end
 * of the picture (every entry is written once per picture, so evaluating the calls then gives what evaluating them in decoding order gave);
 * the kernel fills vertical_bs / horizontal_bs (zeroed, sized as in ohevc_dbk_maps) for ohevc_dev_deblock_maps. */
enum { OHEVC_BS_SLICE_UP = 1, OHEVC_BS_TILE_UP = 2, OHEVC_BS_SLICE_LEFT = 4, OHEVC_BS_TILE_LEFT = 8,      /* lc->slice_or_tiles_{up,left}_boundary, hevc.c:2636-2637 */
       OHEVC_BS_ACROSS_SLICES = 16 };                                                                     /* s->sh.slice_loop_filter_across_slices_enabled_flag */
typedef struct ohevc_bs_call {          /* 8 bytes */
    uint16_t x0, y0;                    /* luma position of the transform / coding block */
    uint8_t  log2_size;                 /* log2_trafo_size / log2_cb_size of the call */
    uint8_t  flags;                     /* OHEVC_BS_* */
    uint16_t reserved;
} ohevc_bs_call;
typedef struct ohevc_bs_maps {
    const uint8_t *mvf;                 /* the picture's motion field: min_pu_width * min_pu_height entries, mvf_stride bytes apart (sizeof(MvField)) */
    int32_t mvf_stride, off_mv, off_poc, off_pred_flag;    /* byte offsets inside an entry: Mv mv[2] (4 x int16), int32 poc[2], pred_flag */
    int32_t pred_flag_bytes;            /* 4 (TEST_MV_POC builds, hevc.h:1032-1041) or 1; values PF_INTRA 0, PF_L0 1, PF_L1 2, PF_BI 3 */
    const uint8_t *cbf_luma;            /* s->cbf_luma: min_tb_width * min_tb_height bytes */
    int32_t min_pu_width, min_pu_height, log2_min_pu_size;
    int32_t min_tb_width, min_tb_height, log2_min_tb_size;
    int32_t log2_ctb_size, bs_width, width, height;
    int32_t loop_filter_across_tiles;   /* pps->loop_filter_across_tiles_enabled_flag */
} ohevc_bs_maps;
/* maps->mvf / cbf_luma, calls, vertical_bs, horizontal_bs: DEVICE pointers; the two outputs must have been zeroed (hevc_frame_start memsets them) */
int ohevc_dev_boundary_strengths(const ohevc_bs_maps *maps, const ohevc_bs_call *calls, int ncalls, uint8_t *vertical_bs, uint8_t *horizontal_bs, void *stream);

/* The motion field without the upload: every inter prediction block already travels as a luma motion-compensation job whose source
 * position and phase are its motion vector and whose reference slot names its reference picture (two pictures of one DPB never share a POC,
 * so "same slot" is the reference's "same POC", hevc_filter.c:584-700).  Writes one OHEVC_MOTION_GRID_ENTRY-byte entry per unit
 * (1 << log2_unit samples, sps->log2_min_pu_size) covered by a luma job of at most 16x16 samples (the tiles ohevc_rec_mc cuts):
 * int16 mv[2][2], int32 ref[2], uint32 pred_flag (1: one reference - in mv[0] / ref[0] whatever its list was, which boundary_strength()
 * does not look at; 3: two) - i.e. ohevc_bs_maps {mvf_stride 20, off_mv 0, off_poc 8, off_pred_flag 16, pred_flag_bytes 4}.  Units no job
 * covers keep what the grid held: zero it before the first call of a picture (pred_flag 0 = PF_INTRA).  jobs, grid: DEVICE pointers. */
enum { OHEVC_MOTION_GRID_ENTRY = 20 };
int ohevc_dev_motion_grid(const ohevc_mc_job *jobs, int njobs, uint8_t *grid, int grid_width, int grid_height, int log2_unit, void *stream);
/* the same over two job arrays in one launch */
int ohevc_dev_motion_grid2(const ohevc_mc_job *jobs, int njobs, const ohevc_mc_job *more, int nmore, uint8_t *grid, int grid_width, int grid_height,
                           int log2_unit, void *stream);

/* device-to-device copy of `bytes` (a multiple of 16; both pointers 16-byte aligned) as a kernel launch: the deblocked copy SAO reads */
int ohevc_dev_copy(void *dst, const void *src, size_t bytes, void *stream);
int ohevc_dev_zero(void *dst, size_t bytes, void *stream);                       /* the same for zeros (16-byte aligned buffer and size) */

/* ---- 2.4 SAO: replaces sao_band_filter / sao_edge_filter[0|1] (hevcdsp.h:60-62; hevcdsp_template.c:340-567)
 * as called from sao_filter_CTB (hevc_filter.c:197-322): dst = the picture, src = its deblocked copy
 * (the reference's sao_frame), one job per CTB and colour plane. */
enum { OHEVC_SAO_BAND = 1, OHEVC_SAO_EDGE = 2 };

typedef struct ohevc_sao_job {          /* 32 bytes */
    uint16_t x, y, w, h;                /* block inside the plane */
    uint8_t  plane;
    uint8_t  type;                      /* OHEVC_SAO_BAND / OHEVC_SAO_EDGE */
    uint8_t  klass;                     /* band: band_position (0..31); edge: eo_class (0..3) */
    uint8_t  borders;                   /* bit i = borders[i] (left, top, right, bottom picture border) */
    uint8_t  restore;                   /* 0: sao_edge_filter[0]; 1: sao_edge_filter[1] (uses the edge flags below) */
    uint8_t  edges;                     /* bit0-1 vert_edge[0..1], bit2-3 horiz_edge[0..1], bit4-7 diag_edge[0..3] */
    int16_t  offset_val[5];             /* SAOParams.offset_val[c_idx][0..4] (hevc.h:519), already << log2_sao_offset_scale */
    uint8_t  quirks;                    /* OHEVC_SAO_LAG_* or 0 */
    uint8_t  reserved[7];
} ohevc_sao_job;

/* The reference front-end filters with a one-CTB lag (ff_hevc_hls_filter, hevc_filter.c:1027-1051) and postpones the
 * horizontal chroma edges of the last 8*h luma columns of a CTB to the next call (hevc_filter.c:541-547).  With 16x16
 * CTBs in 4:2:0 that is the whole CTB, so some of the neighbour samples sao_filter_CTB copies (:289-325) from the first
 * chroma column of the CTB to the right have only been vertically deblocked at that moment:
 *   LAG_BELOW  the samples right of the CTB's last row and diagonally below-right (p0/q0 of the next column's edge
 *              below) -- every CTB row but the last;
 *   LAG_ABOVE  the samples diagonally above-right and right of the CTB's first row (p0/q0 of the edge above) -- the
 *              last two CTB rows, whose filter calls the reference interleaves (ff_hevc_hls_filters, :1053-1063).
 *   LAG_MID    4:2:2 only (a 16x16 CTB is 8x16 chroma samples, with a horizontal chroma edge 8 rows below its top): the
 *              samples right of block rows 7 and 8 (p0/q0 of that edge in the next CTB column).
 * A job carrying a flag reads those samples from `lagged` (the picture as it was between the vertical and the
 * horizontal deblocking pass) and therefore reproduces the reference decoder bit for bit; without flags SAO reads the
 * fully deblocked picture everywhere, as H.265 8.7.3 says. */
enum { OHEVC_SAO_LAG_BELOW = 1, OHEVC_SAO_LAG_ABOVE = 2, OHEVC_SAO_LAG_MID = 4 };

int ohevc_dev_sao_batch(const ohevc_plane dst[3], const ohevc_plane src[3], int bit_depth,
                        const ohevc_sao_job *jobs, int njobs, void *stream);
int ohevc_dev_sao_batch_lagged(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3], int bit_depth,
                               const ohevc_sao_job *jobs, int njobs, void *stream);

/* Samples SAO must not change: restore_tqb_pixels (hevc_filter.c:163-193) puts the deblocked samples back over every
 * min-PU block flagged in the reference's s->is_pcm map (PCM coding units when pcm_loop_filter_disabled_flag is set and
 * cu_transquant_bypass coding units, set_deblocking_bypass hevc.c:1430-1441) right after each sao_* table call.  `map` is
 * that array in DEVICE memory: one byte per min-PU block, `stride` bytes per row, nonzero = keep the deblocked sample.
 * The reference bounds its PU walk with the block's width/height in samples of the plane being filtered while the
 * origin is in luma samples (sao_filter_CTB passes x, y, width, height, hevc_filter.c:275,316): with subsampled chroma
 * only PUs in the first half of the CTB are restored; and it copies `min_pu_size >> hshift` BYTES per row (:176,:184), i.e.
 * only the first half of each PU row when samples are 16 bits wide.  With exact_reference != 0 the kernel reproduces
 * exactly that (bit-identical with the reference decoder); with 0 every sample of a flagged PU is restored in every plane
 * (H.265 8.7.1: pcm_loop_filter_disabled / cu_transquant_bypass samples are not modified by the in-loop filters). */
typedef struct ohevc_sao_bypass {
    const uint8_t *map;
    int32_t stride;
    int32_t log2_min_pu_size;
    int32_t chroma_hshift, chroma_vshift;   /* sps->hshift[1], sps->vshift[1] */
    int32_t exact_reference;
} ohevc_sao_bypass;
int ohevc_dev_sao_batch_bypass(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3], int bit_depth,
                               const ohevc_sao_job *jobs, int njobs, const ohevc_sao_bypass *bypass /* may be NULL */, void *stream);
/* Two kernels serve SAO: a wide form (16 bytes of a row per lane) for blocks with 16-byte row pieces, a power-of-two number of them,
 * offsets that fit a byte and no filter-lag flag, and a general one.  The entry points above run both over all jobs (each block is taken
 * by exactly one).  A caller that sorts its jobs - ohevc_sao_job_is_wide() != 0 first - saves the second pass over them: */
int ohevc_sao_job_is_wide(const ohevc_sao_job *job, const ohevc_plane dst[3], const ohevc_plane src[3], int bit_depth);
int ohevc_dev_sao_batch_sorted(const ohevc_plane dst[3], const ohevc_plane src[3], const ohevc_plane lagged[3], int bit_depth,
                               const ohevc_sao_job *jobs, int n_wide, int n_other, const ohevc_sao_bypass *bypass, void *stream);

/* ---- 2.5 intra prediction: replaces intra_pred[log2-2] (hevcpred.h:32; hevcpred_template.c:30-357) and the
 * predictors it dispatches to, pred_planar / pred_dc / pred_angular (hevcpred.h:34-40).  Everything the
 * reference derives from HEVCContext is resolved by the host into the job: availability AFTER the z-scan
 * qualification (hevcpred_template.c:105-109), the picture-clipped neighbour run lengths (:111-114) and the
 * smoothing switches (:289-296).  Jobs of one launch must be independent (no job reads samples another writes);
 * the ctx layer orders dependent TUs into successive launches. */
enum {
    OHEVC_INTRA_BOTTOM_LEFT = 1, OHEVC_INTRA_LEFT = 2, OHEVC_INTRA_UP_LEFT = 4, OHEVC_INTRA_UP = 8, OHEVC_INTRA_UP_RIGHT = 16,
    OHEVC_INTRA_NO_SMOOTHING = 32,      /* intra_smoothing_disabled_flag, or chroma outside 4:4:4 */
    OHEVC_INTRA_STRONG = 64,            /* sps_strong_intra_smoothing_enable_flag && luma */
    OHEVC_INTRA_LUMA_EDGE = 128         /* c_idx == 0: DC / mode 10 / mode 26 boundary smoothing applies */
};

typedef struct ohevc_intra_job {        /* 16 bytes */
    uint16_t x, y;                      /* block position in its plane (samples) */
    uint8_t  plane;
    uint8_t  log2_size;                 /* 2..5 */
    uint8_t  mode;                      /* 0 planar, 1 DC, 2..34 angular */
    uint8_t  flags;                     /* OHEVC_INTRA_* */
    uint8_t  bottom_left_size;          /* valid samples below the block in the left column (0..N) */
    uint8_t  top_right_size;            /* valid samples right of the block in the top row (0..N) */
    uint8_t  flags2;                    /* OHEVC_INTRA2_* */
    uint8_t  log2_ctb_size;             /* 4..6 when the maker knew it (ohevc_intra_make_job*: geom->log2_ctb_size), else 0: the ctx layer
                                           groups intra work by CTB (2.6b) only when every job of the picture names the same size */
    uint32_t cip_index;                 /* OHEVC_INTRA2_CIP: index of this job's ohevc_intra_cip record */
} ohevc_intra_job;

/* constrained_intra_pred_flag streams (hevcpred_template.c:116-163,185-249): the five availability flags in the job
 * are the RE-DERIVED ones (neighbours coded as inter do not count), and this side record carries what the substitution
 * walk needs: per-sample "is the neighbour intra-coded" bits and the two scan limits. */
enum { OHEVC_INTRA2_CIP = 1 };
typedef struct ohevc_intra_cip {        /* 32 bytes */
    uint8_t top_bits[9];                /* bit k+1: IS_INTRA(k, -1) for k = -1 .. 63 */
    uint8_t left_bits[9];               /* bit k+1: IS_INTRA(-1, k) for k = -1 .. 63 */
    uint8_t size_max_x, size_max_y;     /* hevcpred_template.c:187-198 */
    uint8_t x0_nonzero, y0_nonzero;
    uint8_t reserved[10];
} ohevc_intra_cip;

int ohevc_dev_intra_batch(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, int njobs,
                          void *stream);
/* HOST helper: put the blocks of one dependency level (mutually independent) into the order ohevc_dev_intra_recon_sorted / ohevc_dev_intra_chain
 * want - by size, otherwise as given (decoding order: neighbours in the picture stay neighbours in a wavefront; sorting by prediction mode as
 * well was measured and lost to its cache misses, host_jobs.hip).  Stable; `residuals` (parallel to jobs, may be NULL) is permuted alike;
 * count_by_size[k] = number of (4 << k)-sample blocks.  The ctx layer calls it per level; callers of the device entry points may. */
int ohevc_intra_sort_level(ohevc_intra_job *jobs, ohevc_tu_job *residuals, int n, int32_t count_by_size[4]);
/* same, with the side records of the OHEVC_INTRA2_CIP jobs (device pointer, 16-byte aligned) */
int ohevc_dev_intra_batch_cip(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, int njobs,
                              const ohevc_intra_cip *cip, void *stream);

/* ---- 2.6 a whole chain of intra dependency levels in ONE launch (the ctx layer's executor for intra content).  Work is a list
 * of phases, each a run of `ohevc_level_phase_workgroups()` virtual workgroups:
 *   type 0: intra prediction of `njobs` jobs starting at intra_jobs[first_job];
 *   type 1: residuals of one (log2_size, kind) bin: `njobs` jobs starting at tu_jobs[first_job].
 * Phases are listed in execution order with their running workgroup offset in first_wg; phases that may run side by
 * side share a `step`, and every workgroup of step s waits inside the kernel until all workgroups of step s-1 are done
 * (steps are consecutive from 0).  `sync` = DEVICE array of (number of steps + 2) zeroed uint32 (home XCD, ticket
 * counter, one completion counter per step; consumed by the launch); `need[s]` = DEVICE array, workgroups in step s.  Jobs of one
 * step must be independent, exactly as for the separate entry points; results are identical to launching the phases
 * one after the other with ohevc_dev_intra_batch_cip / ohevc_dev_tu_batch. */
typedef struct ohevc_level_phase {      /* 32 bytes */
    int32_t first_wg, step, type, first_job, njobs, log2_size, kind, reserved;
} ohevc_level_phase;
int ohevc_level_phase_workgroups(int type, int log2_size, int kind, int njobs);
int ohevc_dev_levels(const ohevc_plane planes[3], int bit_depth, const ohevc_level_phase *phases, int nphases, int total_wgs,
                     uint32_t *sync, const uint32_t *need, const ohevc_intra_job *intra_jobs, const ohevc_intra_cip *cips,
                     const ohevc_tu_job *tu_jobs, const int16_t *coeffs, void *stream);


/* Intra prediction of `njobs` mutually independent blocks, each followed - in the same wavefront - by the block's own residual:
 * residuals[i] is the residual of jobs[i] (same plane, position and size; reserved0 = residual kind + 1, coefficient offsets into
 * `coeffs`) or all zeros when block i has none.  What hls_transform_unit does per transform block (hevc.c:1214-1215, 1260-1290), one launch
 * per dependency level instead of ohevc_dev_intra_batch_cip + ohevc_dev_tu_multi.  OHEVC_TU_CROSS residuals are not taken here. */
int ohevc_dev_intra_recon_batch(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, const ohevc_tu_job *residuals, int njobs,
                                const ohevc_intra_cip *cip, const int16_t *coeffs, void *stream);

/* The same work with the jobs SORTED BY SIZE - count_by_size[0] blocks of 4x4 first, then the 8x8, 16x16 and 32x32 ones; residuals[i]
 * (NULL: prediction only) belongs to jobs[i] as above.  N lanes serve an N x N block, so 16 / 8 / 4 / 2 blocks share a wavefront; the
 * neighbour samples are fetched with the substitution rules of hevcpred_template.c:251-286 folded into their addresses, the residual row is
 * added in registers and every row is stored once (no prediction store, no read-back).  Jobs marked OHEVC_INTRA2_CIP are NOT taken here
 * (their substitution walk is sequential: ohevc_dev_intra_recon_batch); neither are OHEVC_TU_CROSS / OHEVC_TU_PCM residuals. */
int ohevc_dev_intra_recon_sorted(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, const ohevc_tu_job *residuals,
                                 const int32_t count_by_size[4], const int16_t *coeffs, void *stream);

/* A run of consecutive NARROW dependency levels in one launch: ONE workgroup of ohevc_intra_chain_workgroup_waves() wavefronts (16 blocks of
 * 4x4, 8 of 8x8, 4 of 16x16 or 2 of 32x32 per wavefront) takes level after level and hands each to the next inside the kernel (workgroup
 * barrier + L1 invalidate: all of a level's wavefronts run on one CU), so a level costs its memory round trips instead of a kernel boundary.
 * A level of more wavefronts than the workgroup has (first_wave[4] up to ohevc_intra_chain_max_waves()) takes further passes of the workgroup.
 * levels[l] (DEVICE array, 16-byte aligned) describes level l like ohevc_dev_intra_recon_sorted's arguments: its jobs sorted by size,
 * njobs[k] blocks of (4 << k) samples, first_wave[k] = running wavefront count, jobs / residuals found at base + 16 * jobs_off16 /
 * res_off16 (res_off16 = 0xffffffff: prediction only).  Same results as one ohevc_dev_intra_recon_sorted launch per level. */
typedef struct ohevc_intra_chain_level {   /* 48 bytes */
    int32_t  first_wave[5];
    int32_t  njobs[4];
    uint32_t jobs_off16, res_off16;
    int32_t  reserved;
} ohevc_intra_chain_level;
int ohevc_intra_chain_max_waves(void);
int ohevc_intra_chain_workgroup_waves(void);
int ohevc_intra_chain_max_levels(void);        /* levels per launch (nlevels <= this) */
int ohevc_dev_intra_chain(const ohevc_plane planes[3], int bit_depth, const void *base, const ohevc_intra_chain_level *levels, int nlevels,
                          const int16_t *coeffs, void *stream);

/* ---- 2.6b every intra-coded block of a picture in ONE launch, coding-tree blocks as tasks (the ctx layer's executor).  Inside a CTB the
 * reference reconstructs block after block in decoding order -- intra_pred[..] (hevcpred_template.c:30-357), then the residual of
 * that block (hevc_cabac.c:1868-1949), whose samples the next block's prediction reads; between CTBs only the wavefront order of
 * hls_decode_entry_wpp remains (a CTB reads its left, above-left, above and above-right neighbours; hevc.c:2779).  A task is one CTB:
 *   tasks[t]   CTB position in CTB units, its run of `ops`, and up to four tasks it must wait for (indices < t, -1 = none): tasks
 *              must be listed so that every dependency comes earlier (raster order of the CTBs does);
 *   ops[]      the CTB's operations in decoding order: bit 31 = 0: intra prediction of intra_jobs[bits 24..0];
 *              bit 31 = 1: residual of tu_jobs[bits 24..0], size log2 = 2 + bits 30..29, kind = bits 28..25 (OHEVC_TU_*);
 *   sync       DEVICE array of 2 * ntasks + 2 zeroed uint32 (consumed by the launch: home XCC, ticket, done flags, progress words).
 * Job records are the ones of 2.1 / 2.5, positions in plane samples of the whole picture.  Results are identical to running the
 * operations one after the other through ohevc_dev_intra_batch_cip / ohevc_dev_tu_batch in a valid order. */
typedef struct ohevc_ctb_task {         /* 32 bytes */
    uint16_t cx, cy;                    /* CTB position, in CTBs */
    uint32_t first_op, nops;
    int32_t  dep[4];                    /* tasks of the left, above-left, above, above-right CTB, or -1 */
    uint32_t reserved;
} ohevc_ctb_task;
int ohevc_dev_ctbs(const ohevc_plane planes[3], int bit_depth, int chroma_format_idc, int log2_ctb_size, const ohevc_ctb_task *tasks, int ntasks,
                   const uint32_t *ops, const ohevc_intra_job *intra_jobs, const ohevc_intra_cip *cips, const ohevc_tu_job *tu_jobs,
                   const int16_t *coeffs, uint32_t *sync, void *stream);

/* Host helper (no GPU work): turn one intra_pred[log2-2](s, x0, y0, c_idx) call of the reference into a job.
 * Inputs are exactly what the reference's front-end holds at the call site (hevc.c:1214-1215): the block position
 * in LUMA samples, HEVClc->na.cand_* (ff_hevc_set_neighbour_available, hevc_mvs.c:41-58), the prediction mode
 * (lc->tu.intra_pred_mode[_c]) and the SPS/PPS geometry.  Performs the z-scan qualification of
 * hevcpred_template.c:105-109 (CTB-local MinTbAddrZs, hevc_ps.c:2551-2567) and the picture clipping of :111-114.
 * For constrained_intra_pred streams use ohevc_intra_make_job_cip (this function returns OHEVC_ERR_ARG for them). */
typedef struct ohevc_intra_geom {
    int32_t width, height;              /* luma samples */
    int32_t chroma_format_idc;          /* 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 */
    int32_t log2_ctb_size, log2_min_tb_size;
    int32_t strong_intra_smoothing;     /* sps_strong_intra_smoothing_enable_flag */
    int32_t intra_smoothing_disabled;   /* spsRext.intra_smoothing_disabled_flag */
    int32_t constrained_intra_pred;     /* pps->constrained_intra_pred_flag */
} ohevc_intra_geom;

int ohevc_intra_make_job(const ohevc_intra_geom *geom, int x0, int y0, int log2_size, int c_idx, int mode,
                         int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                         ohevc_intra_job *out);
/* As above for any stream.  With geom->constrained_intra_pred set, `pred_flag` points at the prediction-mode byte of the
 * first minimum-PU entry of the picture's motion-field map (the reference's s->ref->tab_mvf[0].pred_flag, hevc.h:1032-1041),
 * `pred_flag_stride` is the distance in bytes between entries (sizeof(MvField)) and an entry is intra when the byte equals
 * `intra_value` (PF_INTRA = 0).  Fills *cip and marks the job OHEVC_INTRA2_CIP; the caller stores cip and sets cip_index. */
int ohevc_intra_make_job_cip(const ohevc_intra_geom *geom, int log2_min_pu_size, const uint8_t *pred_flag,
                             ptrdiff_t pred_flag_stride, int intra_value, int x0, int y0, int log2_size, int c_idx, int mode,
                             int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                             ohevc_intra_job *out, ohevc_intra_cip *cip);

/* ------------------------------------------------------------------ 3. library management */
const char *ohevc_last_error(void);                 /* text of the last HIP failure on this thread */
int  ohevc_device_count(void);
int  ohevc_set_device(int device);
/* name of the dominant kernel ohevc_dev_tu_batch launches for (bit_depth, log2_size, kind); for profilers */
const char *ohevc_tu_kernel_name(int bit_depth, int log2_size, int kind);
const char *ohevc_version(void);

/* ---- 2.7 SHVC inter-layer up-sampling: replaces upsample_base_layer_frame and upsample_filter_block_{luma,cr}_{h,v}[3]
 * (hevcdsp.h:106-123; hevcdsp_template.c:1835-2438) together with vdsp.emulated_edge_up_{h,v} (videodsp_template.c:103-166):
 * the base-layer picture is resampled into the enhancement layer's inter-layer reference picture (4:2:0 planes).
 * The reference's position / phase rules live in host-built maps (one entry per output column and row); the kernel is a
 * separable 8-tap (luma) / 4-tap (chroma) gather filter with the reference's int16 intermediate and fixed 12-bit rounding. */
typedef struct ohevc_upsample_params {
    int32_t el_width, el_height, bl_width, bl_height;          /* luma samples (FrameEL / FrameBL coded size) */
    int32_t win_left, win_right, win_top, win_bottom;          /* sps->scaled_ref_layer_window[ref_layer] (HEVCWindow, hevc.h:384-389) */
    int32_t add_x_luma, add_y_luma, scale_x_luma, scale_y_luma;            /* UpsamplInf, hevc.h:347-357 (set_sps, hevc.c:445-499) */
    int32_t add_x_chroma, add_y_chroma, scale_x_chroma, scale_y_chroma;
    int32_t idx;                                               /* UpsamplInf.idx: 0 general, 1 x2, 2 x1.5 (3 = SNR: a copy, not here) */
    int32_t block_slots;                                       /* 0: upsample_base_layer_frame; 1: the per-block slots [idx], whose x2 /
                                                                  x1.5 variants use fixed phase patterns (shipped build: hevc.h:117) */
} ohevc_upsample_params;
typedef struct ohevc_upsample_tap { int16_t pos; uint8_t phase; uint8_t reserved; } ohevc_upsample_tap;   /* centre tap, phase 0..15 */
/* HOST helper: maps of one plane (0 luma, 1/2 chroma): cols[w], col_of[w], rows[h] for the plane's EL size; src_cols / src_rows =
 * the base-layer extent the reference clamps to */
int ohevc_upsample_make_maps(const ohevc_upsample_params *p, int plane, ohevc_upsample_tap *cols, int16_t *col_of,
                             ohevc_upsample_tap *rows, int *src_cols, int *src_rows);
/* one plane; cols / col_of / rows are DEVICE arrays as made above */
int ohevc_dev_upsample_plane(const ohevc_plane *dst, const ohevc_plane *src, int bit_depth, int chroma, const ohevc_upsample_tap *cols,
                             const int16_t *col_of, const ohevc_upsample_tap *rows, int src_cols, int src_rows, void *stream);
/* the three planes of an inter-layer picture in one launch (same maps and extents as three ohevc_dev_upsample_plane calls: luma with the
 * 8-tap filters, the two chroma planes with the 4-tap ones) */
int ohevc_dev_upsample_picture(const ohevc_plane dst[3], const ohevc_plane src[3], int bit_depth, const ohevc_upsample_tap *const cols[3],
                               const int16_t *const col_of[3], const ohevc_upsample_tap *const rows[3], const int src_cols[3], const int src_rows[3],
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OHEVC_HIP_H */
