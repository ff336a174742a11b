/*
 * ohevc_ctx.h -- host-side layer of libohevc_hip.so: device-resident picture store (the DPB's pixel planes), a job
 * recorder and the phase-ordered executor that turns one frame's recorded jobs into kernel launches.
 *
 * This is the "extra API" a drop-in needs that the reference does not have: the reference's table
 * slots compute immediately on host memory; here the front-end records, and flushes at the points where the
 * reference's own schedule allows it:
 *     hevc_frame_start (hevc.c:3197)                      -> ohevc_frame_begin
 *     table calls in hls_coding_unit / hls_transform_unit -> ohevc_rec_* (coefficients are COPIED at call time: the
 *                                                            reference reuses lc->tu.coeffs for the next TU, hevc.h:1063)
 *     end of a CTU row / before ff_hevc_hls_filters       -> ohevc_frame_reconstruct (optional, for overlap)
 *     frame end, before output / MD5 (hevc.c:4146-4181)   -> ohevc_frame_end, ohevc_pic_download
 *
 * Ordering rules preserved: inter prediction first (reads other pictures only), then the inter
 * residuals, then intra blocks in dependency levels (each level: prediction, then its residuals) so that every intra
 * block sees reconstructed, un-deblocked neighbours; then all vertical edges, all horizontal edges, then SAO from a
 * deblocked copy (the reference's sao_frame).
 *
 * One ohevc_ctx serves one decoding thread (like one HEVCContext); it is not internally synchronised.
 */
#ifndef OHEVC_CTX_H
#define OHEVC_CTX_H

#include "ohevc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ohevc_ctx ohevc_ctx;

int  ohevc_ctx_create(ohevc_ctx **out, int device);
/* One context per decoding thread.  Contexts created with share_with != NULL use share_with's picture store (the slots
 * of ohevc_pic_alloc are common to all of them) and its device, each on its own stream: a frame thread can predict from
 * pictures another thread reconstructs.  Ordering between the streams is kept by the library: a frame waits for the
 * frame_end of every picture it references (blocking the host until that frame_end has been issued by its thread) and
 * for earlier users of the memory it overwrites.  Destroy the sharing contexts before the one they were created from,
 * or all of them after the decoding threads have stopped; the pictures are freed with the last context. */
int  ohevc_ctx_create_shared(ohevc_ctx **out, int device, ohevc_ctx *share_with);
const void *ohevc_ctx_store_id(ohevc_ctx *ctx);            /* identity of the picture store (equal for sharing contexts) */
void ohevc_ctx_destroy(ohevc_ctx *ctx);
/* Several threads record into this context at once (the reference's slice threads: WPP rows / tiles of ONE picture, all joined
 * before the frame ends).  The thread that called ohevc_frame_begin keeps recording into the context itself, every other thread
 * gets a private recorder that ohevc_frame_reconstruct / ohevc_frame_end fold in (jobs keep their dependency levels): no lock on
 * the recording path.  The caller guarantees what the reference's WPP / tile decoding guarantees: a block is recorded after the
 * blocks it predicts from.  Off (default): a context is not internally synchronised. */
int  ohevc_ctx_set_concurrent(ohevc_ctx *ctx, int on);
/* the context's HIP stream (hipStream_t as void*): the same handle for the context's whole life.  Everything the context has issued when
 * ohevc_frame_end / ohevc_pic_* return is ordered before what the caller enqueues on this handle afterwards (a picture with a long dependency
 * chain is issued on a second, internal stream and joins this one at its frame end). */
void *ohevc_ctx_stream(ohevc_ctx *ctx);
int  ohevc_ctx_sync(ohevc_ctx *ctx);

/* A picture store holds at most OHEVC_MAX_PICTURES pictures (ohevc_pic_alloc / ohevc_pic_adopt fail with OHEVC_ERR_ARG beyond: a reference slot is one
 * byte of a motion-compensation job, and the tables of the layers above are sized by it); picture sides are at most 65535 samples (16-bit job
 * coordinates).  HEVC needs 17 per layer (a DPB of 16 + the current picture). */
enum { OHEVC_MAX_PICTURES = 127 };

/* ---- picture store: pixel planes of DPB entries, resident in HBM for the life of the HEVCFrame (hevc_refs.c:75-147) */
int  ohevc_pic_alloc(ohevc_ctx *ctx, int width, int height, int chroma_format_idc, int bit_depth);   /* slot >= 0 or error */
/* register planes allocated by someone else (e.g. tensors an RCCL broadcast writes into) as a picture; never freed by ctx */
int  ohevc_pic_adopt(ohevc_ctx *ctx, const ohevc_plane planes[3], int width, int height, int chroma_format_idc, int bit_depth);
int  ohevc_pic_release(ohevc_ctx *ctx, int slot);
int  ohevc_pic_upload(ohevc_ctx *ctx, int slot, int plane, const void *host, ptrdiff_t host_stride);
int  ohevc_pic_download(ohevc_ctx *ctx, int slot, int plane, void *host, ptrdiff_t host_stride);
int  ohevc_pic_planes(ohevc_ctx *ctx, int slot, ohevc_plane out[3]);     /* device views (e.g. for an RCCL broadcast) */
/* the three planes of a picture (host[i] NULL: skip) with one wait at the end */
int  ohevc_pic_download_planes(ohevc_ctx *ctx, int slot, void *const host[3], const ptrdiff_t host_stride[3]);
/* ... or QUEUED behind the picture's device work on this context's stream and not waited for: the decoding thread goes on, ohevc_pic_wait_host
 * (below) returns when the samples are in host[].  The planes must be page-locked memory that stays so until then - blocks of ohevc_host_alloc
 * (with pageable memory the runtime turns the call into a synchronous copy; a page lock dropped by ohevc_host_unpin in between is NOT waited
 * for here).  The application's thread then only waits for an event where it used to wait for the device AND issue three copies. */
int  ohevc_pic_download_queue(ohevc_ctx *ctx, int slot, void *const host[3], const ptrdiff_t host_stride[3]);
/* Page-lock application memory that ohevc_pic_download / ohevc_pic_upload will be given again and again - the decoder's frame buffers
 * (alloc_frame, hevc_refs.c:75-114: pass the allocations its buffer pool recycles, AVFrame.buf[i]->data / ->size): the copy-back then is
 * one DMA at the bus rate instead of a staged copy through pageable memory.  A range that overlaps an earlier, different registration
 * replaces it (that memory was freed and allocated again).  OHEVC_ERR_HIP when the runtime refuses: nothing is lost but the speed.
 * ohevc_host_unpin_all before the application frees the memory (the decoder: before avcodec_close). */
int  ohevc_host_pin(ohevc_ctx *ctx, void *ptr, size_t bytes);
int  ohevc_host_unpin_all(ohevc_ctx *ctx);
/* ... or of one allocation only (every registered range overlapping it): the application returned THAT memory to its allocator, e.g. the decoder's
 * buffer pool changed geometry and a buffer address came back with another size.  Copies into other ranges that are in flight are not disturbed. */
int  ohevc_host_unpin(ohevc_ctx *ctx, void *ptr, size_t bytes);
/* Page-locked host memory of the library's own (hipHostMalloc; 64-byte aligned, not cleared): what an application - or the sample hooks'
 * get_buffer2 (integration/hip_hooks.c) - builds the decoder's frame buffers from when it lets the back end allocate them.
 * Memory BORN page-locked needs no ohevc_host_pin, and, unlike a registration of somebody else's allocation, cannot go stale behind the
 * library's back: the reference's frame pool frees and re-creates its buffers in mid-stream (update_frame_pool, utils.c:509-575, reached from
 * every frame thread through one shared FramePool), a large buffer comes back from mmap at the SAME address with the SAME size, and a
 * registration that still names the dead mapping makes the next copy-back a device fault (round 6: 8K frame threads; profiles/r6u_*).
 * ohevc_host_free takes no context: a block may outlive the context (frames the application still holds when the decoder is closed). */
int  ohevc_host_alloc(ohevc_ctx *ctx, size_t bytes, void **out);
int  ohevc_host_free(void *ptr);
int  ohevc_host_alloc_pins(const ohevc_ctx *ctx);  /* 1: this context's ohevc_host_alloc page-locks (it has a device), 0: a record-only context */
int  ohevc_host_block_pinned(const void *ptr);     /* 1: page-locked, 0: plain memory (made by a record-only context), -1: not a block of ohevc_host_alloc */
/* Choices a decoder instance makes for the contexts it creates (value < 0: back to the process default, which include/ohevc_debug.h's setters
 * move for tests).  OHEVC_OPT_LEVEL_LAUNCH: executor of the intra-coded blocks - 0 dependency levels (chain kernel; default), 1 all levels in one
 * launch, 2 chosen per picture, 3 CTB tasks; OHEVC_OPT_FILTERS_ON_DEVICE: 1 deblocking parameters derived on the device from the decoder's maps
 * (default), 0 one job per edge derived on the host. */
enum { OHEVC_OPT_LEVEL_LAUNCH = 0, OHEVC_OPT_FILTERS_ON_DEVICE = 1, OHEVC_OPT_PARK_FRAMES = 2 };   /* PARK_FRAMES: 1 ohevc_frame_end_deferred may park, 0 (default: parking lost its A/B runs) it is ohevc_frame_end */
int  ohevc_ctx_set_option(ohevc_ctx *ctx, int option, int value);
int  ohevc_ctx_get_option(const ohevc_ctx *ctx, int option);
/* Frame-parallel decoding across GPUs (one process per GPU; the reference's counterpart is the shared DPB of its frame threads,
 * pthread_frame.c:479-513 + hevc_await_progress hevc.c:1951-1958): the owner of a picture copies a finished plane out with
 * ohevc_pic_export, the other processes copy it into their own store with ohevc_pic_import; the transport in between (RCCL
 * broadcast over xGMI) is the application's.  Buffers are DEVICE memory of exactly stride x height bytes (ohevc_pic_planes); both
 * calls are ordered against the frames of every context of the store and return when the copy is complete. */
int  ohevc_pic_export(ohevc_ctx *ctx, int slot, int plane, void *device_dst, size_t bytes);
int  ohevc_pic_import(ohevc_ctx *ctx, int slot, int plane, const void *device_src, size_t bytes);
/* The same by CTU-row bands (the reference's frame threads publish a picture row by row, ff_thread_report_progress at the end of every CTB row,
 * hevc_filter.c / hevc.c:2934-2937, and a dependent picture waits for the rows its motion vectors reach, hevc.c:1951-1958): rows
 * [row0, row0 + rows) of the plane; device_plane_base addresses a buffer laid out like the WHOLE plane, the band lies at row0 * stride in it.
 * first != 0 on the first import of a picture: it orders the slot's memory against its earlier readers / writers. */
int  ohevc_pic_export_rows(ohevc_ctx *ctx, int slot, int plane, int row0, int rows, void *device_plane_base);
int  ohevc_pic_import_rows(ohevc_ctx *ctx, int slot, int plane, int row0, int rows, const void *device_plane_base, int first);
/* a whole band - the row ranges of all three planes (device_plane_base[i] NULL: no such plane) - with ONE wait at the end instead of one per plane */
int  ohevc_pic_export_band(ohevc_ctx *ctx, int slot, const int row0[3], const int rows[3], void *const device_plane_base[3]);
int  ohevc_pic_import_band(ohevc_ctx *ctx, int slot, const int row0[3], const int rows[3], const void *const device_plane_base[3], int first);
/* The deepest LUMA row of reference picture `slot` that the motion compensation recorded for the open frame reads, filter taps included
 * (-1: the frame does not predict from it): how much of a remote picture must have arrived before the frame launches. */
int  ohevc_frame_ref_reach(ohevc_ctx *ctx, int slot);
int  ohevc_pic_info(ohevc_ctx *ctx, int slot, int *width, int *height, int *chroma_format_idc, int *bit_depth);

/* SHVC: resample picture src_slot (a base-layer picture of the store) into picture dst_slot, the enhancement layer's
 * inter-layer reference picture -- what the 13 upsample_* slots do in the reference (ohevc_upsample_params in ohevc_hip.h).
 * Ordered against the frames that write src / still read dst's memory like a frame of its own. */
int  ohevc_pic_upsample(ohevc_ctx *ctx, int dst_slot, int src_slot, const ohevc_upsample_params *params);

/* ---- per-frame recording */
int  ohevc_frame_begin(ohevc_ctx *ctx, int slot);
/* transform_add[..] preceded by its inverse transform (kind = OHEVC_TU_*).  `intra` != 0 when the block was just
 * predicted by ohevc_rec_intra at the same position (it then runs right after that prediction's level). */
int  ohevc_rec_tu(ohevc_ctx *ctx, int plane, int x, int y, int log2_size, int kind, const int16_t *coeffs, int intra);
/* ... with the caller's promise that every coefficient outside the top-left cols x rows rectangle of the block is zero: only that rectangle is
 * copied, staged and uploaded (the device rebuilds the dense block, ohevc_expand_rec in ohevc_hip.h).  For inverse-DCT blocks the reference
 * hands its idct slot the bound as col_limit (hevc_cabac.c:1923-1934): cols = min(col_limit, N), rows = min(col_limit + 4, N), the very
 * ranges its own transforms read (hevcdsp_template.c:271-291).  Other kinds are recorded whole whatever is passed. */
int  ohevc_rec_tu_limited(ohevc_ctx *ctx, int plane, int x, int y, int log2_size, int kind, const int16_t *coeffs, int intra, int cols, int rows);
/* a chroma block with cross-component prediction (OHEVC_TU_CROSS): own residual (kind_c, coeffs_c; kind_c = -1 and coeffs_c
 * NULL when the block has no coded coefficients) plus (res_scale_val * luma residual) >> 3, the luma residual being that of
 * (kind_y, coeffs_y), the RAW coefficients of the transform unit's luma block.  Both blocks are copied. */
int  ohevc_rec_tu_cross(ohevc_ctx *ctx, int plane, int x, int y, int log2_size, int kind_c, const int16_t *coeffs_c, int kind_y,
                        const int16_t *coeffs_y, int res_scale_val, int intra);
int  ohevc_rec_mc(ohevc_ctx *ctx, const ohevc_mc_job *job);              /* ref0/ref1 are picture-store slots */
int  ohevc_rec_intra(ohevc_ctx *ctx, const ohevc_intra_job *job);
/* job marked OHEVC_INTRA2_CIP: `cip` is copied and job->cip_index is assigned by the recorder */
int  ohevc_rec_intra_cip(ohevc_ctx *ctx, const ohevc_intra_job *job, const ohevc_intra_cip *cip);
int  ohevc_rec_deblock(ohevc_ctx *ctx, const ohevc_dbk_job *job);
int  ohevc_rec_sao(ohevc_ctx *ctx, const ohevc_sao_job *job);

/* The reference's s->is_pcm map of the picture being recorded (one byte per min-PU block, `stride` bytes per row):
 * samples of flagged blocks come out of SAO with their deblocked value (restore_tqb_pixels, hevc_filter.c:163-193; see
 * ohevc_sao_bypass in ohevc_hip.h).  Call any time between ohevc_frame_begin and ohevc_frame_end when
 * pps->transquant_bypass_enable_flag || (sps->pcm_enabled_flag && sps->pcm.loop_filter_disable_flag); the bytes are copied.
 * exact_reference: see ohevc_sao_bypass (1 = bit-identical with the reference decoder's partial restore). */
int  ohevc_frame_set_bypass_map(ohevc_ctx *ctx, const uint8_t *map, int stride, int width_pu, int height_pu, int log2_min_pu_size,
                                int exact_reference);

/* Bulk forms (one call per CTU row / frame instead of one per block).  Records of different kinds may be handed over
 * in any order EXCEPT that a block's intra job must be recorded before its residual (the residual inherits the
 * prediction's dependency level).  ohevc_rec_tu_bulk: desc = n x {plane, x, y, log2_size, kind, intra} (int32 each),
 * coeffs = the n dense blocks back to back (N*N int16 each, also for OHEVC_TU_DC, which reads only [0]). */
int  ohevc_rec_mc_bulk(ohevc_ctx *ctx, const ohevc_mc_job *jobs, int n);
int  ohevc_rec_intra_bulk(ohevc_ctx *ctx, const ohevc_intra_job *jobs, int n);
int  ohevc_rec_tu_bulk(ohevc_ctx *ctx, int n, const int32_t *desc, const int16_t *coeffs);
int  ohevc_rec_deblock_bulk(ohevc_ctx *ctx, const ohevc_dbk_job *jobs, int n);
int  ohevc_rec_sao_bulk(ohevc_ctx *ctx, const ohevc_sao_job *jobs, int n);
/* The deblocking of the whole current picture as the decoder's maps (ohevc_hip.h: ohevc_dbk_maps, HOST pointers here): the arrays
 * are copied, uploaded at the frame end and the edges derived on the device (ohevc_dev_deblock_maps) - the bulk form of every
 * ohevc_rec_deblock call of the picture.  Needs a device: record-only contexts (ohevc_ctx_has_device() == 0) refuse it. */
int  ohevc_rec_deblock_maps(ohevc_ctx *ctx, const ohevc_dbk_maps *maps);
/* ... with the boundary strengths derived on the device too (ohevc_hip.h, ohevc_dev_boundary_strengths): maps->vertical_bs / horizontal_bs are
 * not read; *bs carries HOST pointers to the picture's motion field and cbf_luma map (copied), and every call the reference would have made
 * of ff_hevc_deblocking_boundary_strengths is recorded with ohevc_rec_bs_call(ctx, x0, y0, log2_size, OHEVC_BS_* flags). */
int  ohevc_rec_deblock_maps_bs(ohevc_ctx *ctx, const ohevc_dbk_maps *maps, const ohevc_bs_maps *bs);
int  ohevc_rec_bs_call(ohevc_ctx *ctx, int x0, int y0, int log2_size, int flags);
int  ohevc_rec_bs_calls(ohevc_ctx *ctx, const ohevc_bs_call *calls, int n);
/* The motion field does not have to travel: every inter prediction block of the frame is recorded as a luma MC job that carries its motion
 * vector (source position, phase) and its reference picture (slot).  After this call (once per frame, after ohevc_frame_begin)
 * ohevc_frame_reconstruct keeps them in a device-side grid of (1 << log2_unit)-sample units (ohevc_dev_motion_grid; log2_unit =
 * sps->log2_min_pu_size) and ohevc_rec_deblock_maps_bs takes bs->mvf == NULL: the frame end derives the boundary strengths from the grid. */
int  ohevc_frame_keep_motion(ohevc_ctx *ctx, int log2_unit);
int  ohevc_ctx_has_device(const ohevc_ctx *ctx);

/* upload + launch prediction/residual work recorded so far (may be called several times per frame) */
int  ohevc_frame_reconstruct(ohevc_ctx *ctx);
/* The same, for the end of a CTU row of a picture WITHOUT inter prediction so far (an intra picture: one long dependency chain on the device
 * that the rest of its GOP waits for): hands the recorded work over if at least min_pending_kib KiB of job records and coefficients are waiting, so that the device
 * works on the picture's first rows while the host parses its last.  A no-op for frames that have recorded inter prediction (their reference
 * pictures' frame ends may not be issued yet: that wait belongs at the frame end), for contexts several threads record into, and for
 * record-only contexts. */
int  ohevc_frame_flush_intra(ohevc_ctx *ctx, int min_pending_kib);
/* reconstruct + in-loop filters (vertical edges, horizontal edges, SAO); the picture is final when this returns OK
 * and the stream has drained (ohevc_ctx_sync / ohevc_pic_download) */
int  ohevc_frame_end(ohevc_ctx *ctx);

/* The same frame end WITHOUT its host-side cost in the calling thread (for the reference's frame threads, pthread_frame.c: several decoding
 * threads, each with a context of one store).  The recorded frame is handed to the store's issuer thread and the call returns; the issuer
 * issues queued frames one after the other - each as soon as the frame ends of the pictures it references have been issued - on streams of
 * its own, and queues the copy-back of the three planes into host[] (NULL: none; page-locked memory, ohevc_host_pin, makes it a DMA) behind
 * it.  The decoding thread can begin its next picture at once.  ohevc_pic_wait_host: the application takes the picture out - returns when
 * the copy has landed.  Errors of an asynchronous frame end mark the picture failed (dependents fail too) and are reported once by
 * ohevc_ctx_async_status.  ohevc_ctx_sync / ohevc_pic_release / ohevc_ctx_destroy wait for submitted frames. */
int  ohevc_frame_end_async(ohevc_ctx *ctx, void *const host[3], const ptrdiff_t host_stride[3]);
int  ohevc_pic_wait_host(ohevc_ctx *ctx, int slot);
int  ohevc_ctx_async_status(ohevc_ctx *ctx);
int  ohevc_ctx_async_profile(ohevc_ctx *ctx, double *busy_seconds, long long *frames);

/* Give up the frame being recorded (a recording error, a failed launch): drops what was recorded and publishes the picture as
 * complete-with-error, so that other contexts of the store that reference it -- they block until its frame end has been issued --
 * fail at once with OHEVC_ERR_STATE instead of timing out.  ohevc_frame_end does this itself when it returns an error. */
int  ohevc_frame_abort(ohevc_ctx *ctx);

/* ohevc_frame_end without the wait for other threads: if the frame end of every reference picture of the open frame has been issued (always so with
 * one decoding thread) this IS ohevc_frame_end.  If not - frame threads: a picture predicts from one another thread is still parsing - the recorded
 * frame is parked on the picture store and the call returns; the thread that issues the last missing reference issues the parked frame right
 * behind it.  The picture's samples are complete where ohevc_pic_download* / ohevc_pic_export* deliver them (they wait, and help).  For callers that
 * copy the picture back later (the sample hooks' deferred copy-back); a caller that needs the picture on the device when the call returns uses
 * ohevc_frame_end.  Parking is an option of the context, OFF by default (ohevc_ctx_set_option(OHEVC_OPT_PARK_FRAMES, 1) / ohhip_options.park_frames switch
 * it on): without it this call is ohevc_frame_end. */
int  ohevc_frame_end_deferred(ohevc_ctx *ctx);

/* statistics of the last ohevc_frame_end, for benches: launches issued, intra dependency levels, bytes uploaded */
typedef struct ohevc_frame_stats {
    int32_t launches, intra_levels;
    int64_t upload_bytes;
    int32_t n_tu, n_mc, n_intra, n_dbk, n_sao;
    int32_t chose_ctbs;                 /* the picture's intra blocks ran as CTB tasks (one launch) rather than as dependency levels */
    int32_t reserved;
    int64_t alg_bytes;                  /* algorithmic HBM bytes of the picture's recorded jobs: the per-unit figures of SURVEY.md 8(d) summed over
                                           the records (residual (2 + 2P) N^2, MC P (w+T-1)(h+T-1) per reference + P w h, intra P (4N+1) + P N^2,
                                           deblocking 2P per touched sample, SAO P (w+2)(h+2) + P w h) -- the traffic floor of the device work */
} ohevc_frame_stats;
int  ohevc_frame_get_stats(ohevc_ctx *ctx, ohevc_frame_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* OHEVC_CTX_H */
