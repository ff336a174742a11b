/*
 * integration/hip_backend.h -- what the APPLICATION side of openHEVC (gpac/modules/openhevc_dec/openHevcWrapper.c, main_hm/main.c) calls to
 * give a decoder instance the gfx950 back end: one ohhip_backend per AVCodecContext it opens.
 *
 * The reference's public API opens several decoders per handle (openHevcWrapper.c:27,47-93: MAX_DECODERS 2, one AVCodecContext each) and its
 * frame threads copy contexts (pthread_frame.c:276: the copies inherit avctx->opaque; hevc.c:4502-4513), and it configures a decoder through
 * AVOptions set before avcodec_open2 (hevc.c:4534-4546: "decode-checksum", "decoder-id", ...).  The back end follows that shape:
 *
 *     ohhip_options o;
 *     ohhip_options_default(&o);                 // environment variables only supply the defaults
 *     o.device = 3;                              // ... everything can be chosen per decoder
 *     ohhip_backend *be = ohhip_backend_new(&o);
 *     ohhip_backend_attach(be, avctx);           // BEFORE avcodec_open2 (sets avctx->opaque, which every thread copy inherits)
 *     avcodec_open2(avctx, codec, NULL);
 *     ... avcodec_decode_video2(avctx, frame, &got, &pkt); ohhip_backend_frame_done(be); ohhip_backend_fetch_output(be, frame->data, frame->linesize); ...
 *     ohhip_backend_pre_close(be); avcodec_close(avctx); ohhip_backend_free(be);
 *
 * Every hook in hip_hooks.c finds its instance through the decoder context it is handed (s->avctx->opaque, checked against the registry of
 * live back ends) - nothing hangs off file-scope state, so two decoders of one process, each with its own threads, never see each other's
 * pictures, errors, modes or device.  A decoder nobody attached a back end to gets the process default (created on first use from the
 * environment defaults; this is what a three-line patch of the reference gets).
 */
#ifndef OHHIP_BACKEND_H
#define OHHIP_BACKEND_H
#include <stdint.h>
#include <stddef.h>
#include "ohevc_ctx.h"
#include "ohevc_frames.h"

#ifdef __cplusplus
extern "C" {
#endif

struct AVCodecContext;
typedef struct ohhip_backend ohhip_backend;

/* ABI rule of this struct (and of ohhip_frames_mode, ohevc_frames.h): struct_size comes FIRST and is sizeof(the struct) as the HOST was compiled
 * (ohhip_options_default fills it in); new fields are only ever APPENDED.  ohhip_backend_new refuses struct_size 0 ("not initialised") and
 * sizes larger than its own; fields beyond the caller's size keep their library defaults (-1 / NULL), so a host built against an older header
 * keeps working and can never be over-read. */
typedef struct ohhip_options {
    size_t struct_size;      /* sizeof(ohhip_options) of the caller; set by ohhip_options_default */
    int device;              /* HIP device ordinal of this decoder                                     (default: OHHIP_DEVICE or 0) */
    int bulk_filters;        /* 1: the in-loop filter drivers in bulk at the frame end, 0: per-edge calls (default 1; OHHIP_BULK_FILTERS) */
    int defer_download;      /* 1: a picture is copied back when the application fetches it (ohhip_backend_fetch_output), 0: in the frame-end hook
                              *                                                                          (default 1; OHHIP_DEFER_DOWNLOAD) */
    int pin_frames;          /* 1: page-lock frame buffers the back end did not make (own_frames 0, or the application's get_buffer2); buffers of the
                              * decoder's own pool lose the lock where the pool frees them (default 0 since round 6 - opt-in; OHHIP_PIN_FRAMES) */
    int async_issue;         /* 1: frame ends issued by the library's issuer threads                   (default 0; OHHIP_ASYNC_ISSUE) */
    int record_only;         /* 1: no device, no pixels: host-side profiling / software-executor tests (default 0; OHHIP_RECORD_ONLY) */
    int test_fail_index;     /* fault injection of the multi-process tests: the owner fails on this picture (default -1; OHHIP_TEST_FAIL_INDEX) */
    int flush_intra_kib;     /* an intra picture's recorded work goes to the device at the end of a CTU row once this many KiB are waiting; 0: only at the frame end; -1 (default): by the picture's size - 0.45 bytes per luma sample, 512 .. 4096 KiB: 911 KiB at 1080p (OHHIP_FLUSH_INTRA_KIB) */
    int level_launch;        /* executor of the intra-coded blocks of THIS decoder's contexts (OHEVC_OPT_LEVEL_LAUNCH, ohevc_ctx.h): 0 levels, 1 one level kernel, 2 chosen per picture, 3 CTB tasks; -1 (default): the library's default (OHHIP_LEVEL_LAUNCH) */
    int device_filters;      /* 1: deblocking parameters derived on the device from the decoder's maps, 0: one job per edge derived on the host; -1 (default): the library's default (OHHIP_DEVICE_FILTERS) */
    int crash_backtrace;     /* 1: SIGSEGV / SIGABRT print a backtrace (debugging aid; default 0; OHHIP_BACKTRACE) */
    const char *trace_path;  /* per-picture host timeline (parse start, hook start, issue end, hook end) written here at free (default OHHIP_TRACE_FRAMES) */
    struct ohhip_backend *base_layer;   /* SHVC: this decoder is an enhancement-layer decoder (decoder-id > 0, openHevcWrapper.c:92) and names the
                                           back end of the decoder its BL_avcontext points at (openHevcWrapper.c:107-108).  The two then share one
                                           device picture store: the inter-layer reference picture is resampled on the device from the base-layer
                                           picture where it lies (hevc.c:2077-2099, hevc_filter.c:1377-1430).  Free the enhancement layer's back end
                                           before the base layer's.  (default NULL) */
    int park_frames;         /* frame threads: 1: a frame end whose reference pictures have not been issued yet is parked instead of making the decoding
                              * thread wait (ohevc_frame_end_deferred, ohevc_ctx.h; needs defer_download), 0: it waits; -1 (default): the library's
                              * default (OHHIP_PARK_FRAMES) */
    int own_frames;          /* 1 (default; OHHIP_OWN_FRAMES): ohhip_backend_attach installs a get_buffer2 that builds the decoder's frame buffers from
                              * page-locked memory of the back end's own (ohevc_host_alloc), recycled by the back end until ohhip_backend_free - unless the
                              * application has installed a get_buffer2 of its own; 0: the decoder's allocator stays, pin_frames page-locks what it hands out.
                              * Why it is the default: the reference's frame pool frees and re-creates its buffers in mid-stream (utils.c:509-575), a
                              * page lock on memory that was freed and mapped again at the same address is dead, and nothing tells the back end */
    int queue_download;      /* 1 (default; OHHIP_QUEUE_DOWNLOAD): with defer_download and own_frames the copy-back of a picture is QUEUED behind its device
                              * work by the decoding thread at the frame end (ohevc_pic_download_queue) and ohhip_backend_fetch_output only waits for
                              * it; 0: fetch_output issues the copies itself (round 5) */
} ohhip_options;

void ohhip_options_default(ohhip_options *o);
size_t ohhip_options_size(void);                            /* sizeof(ohhip_options) as the back end was compiled (a host's struct_size may be smaller, never larger) */
ohhip_backend *ohhip_backend_new(const ohhip_options *o);                 /* NULL: refused (struct_size) or the context could not be made; the reason goes to stderr */
/* the options this back end RUNS with (defaults filled in; a caller's shorter struct completed): out->struct_size says how much of *out may be
 * written (ohhip_options_default(out) first, or set it by hand); returns 0, -1 for a bad handle or size */
int  ohhip_backend_options(const ohhip_backend *be, ohhip_options *out);
int  ohhip_backend_attach(ohhip_backend *be, struct AVCodecContext *avctx);   /* before avcodec_open2 */
/* "frame complete, before output" for one decoding thread (with frame threads the decoder's own end-of-frame report has done it already) */
int  ohhip_backend_frame_done(ohhip_backend *be);
int  ohhip_backend_frame_failed(ohhip_backend *be);         /* avcodec_decode_video2 returned an error after hevc_frame_start */
/* the application takes a picture out: with a deferred / asynchronous copy-back this is where its samples reach the host */
int  ohhip_backend_fetch_output(ohhip_backend *be, uint8_t *const data[3], const int linesize[3]);
void ohhip_backend_pre_close(ohhip_backend *be);            /* before avcodec_close: the page locks of the frame buffers go first */
void ohhip_backend_free(ohhip_backend *be);                 /* after the decoder and its threads are gone */
/* frame-parallel decoding over processes (hip_frames.h): switch on (m != NULL) before the first picture; ..._install after avcodec_open2 */
int  ohhip_backend_frames_mode(ohhip_backend *be, const ohhip_frames_mode *m);
void ohhip_backend_frames_install(ohhip_backend *be, struct AVCodecContext *avctx);
int  ohhip_backend_frame_is_local(ohhip_backend *be, const unsigned char *data0);
int  ohhip_backend_device(const ohhip_backend *be);
void ohhip_frame_pool_trim(void);                           /* own_frames: blocks of closed decoders are kept for the next one (OHHIP_BLOCK_CACHE_MB, default 1024): give them back */
void ohhip_frame_pool_counts(long long *made, long long *live);   /* own_frames: page-locked frame-buffer blocks ever made / existing now, process-wide */
int  ohhip_backend_live_count(void);                        /* back ends alive in this process (tests: open / close must not leak) */

#ifdef __cplusplus
}
#endif
#endif
