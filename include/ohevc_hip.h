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

/* One picture plane in HBM.  data must be 16-byte aligned, stride a multiple of 16 bytes. */
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
    OHEVC_TU_NKINDS = 9
};

typedef struct ohevc_tu_job {           /* 16 bytes */
    uint16_t x, y;                      /* top-left sample of the block inside its plane; multiples of the block size */
    uint8_t  plane;                     /* index into planes[3] */
    uint8_t  reserved0;
    int16_t  dc;                        /* OHEVC_TU_DC: coeffs[0] (no arena storage needed) */
    uint32_t coeff_off;                 /* offset of the dense N*N int16 block in the coefficient arena, in int16
                                           units; must be a multiple of 8 (16 bytes).  Ignored for OHEVC_TU_DC */
    uint32_t reserved1;
} ohevc_tu_job;

/* Runs `njobs` blocks of one size (1 << log2_size, 2..5) and one residual kind.  Blocks of a batch must not
 * overlap (they never do inside a frame).  col_limit of the reference's idct is not needed: a full transform is
 * result-identical on the inputs the decoder can produce (hevc_cabac.c:1923-1934, x86/hevc_idct_sse.c:504). */
int ohevc_dev_tu_batch(const ohevc_plane planes[3], int bit_depth, int log2_size, int kind,
                       const ohevc_tu_job *jobs, int njobs, const int16_t *coeffs, void *stream);

/* ------------------------------------------------------------------ 3. library management */
const char *ohevc_last_error(void);                 /* text of the last HIP failure on this thread */
int  ohevc_device_count(void);
int  ohevc_set_device(int device);
/* name of the dominant kernel ohevc_dev_tu_batch launches for (bit_depth, log2_size, kind); for profilers */
const char *ohevc_tu_kernel_name(int bit_depth, int log2_size, int kind);
const char *ohevc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OHEVC_HIP_H */
