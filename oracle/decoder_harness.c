/*
 * oracle/decoder_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A minimal caller of the reference's own decoder API (libavcodec as vendored in /root/reference: avcodec_open2 /
 * avcodec_decode_video2, the same calls gpac/modules/openhevc_dec/openHevcWrapper.c:46-134 makes), compiled against
 * the reference headers and linked with the reference's objects built in place (oracle/Makefile, target `fulldec`).
 * One access unit (Annex-B bytes) per call, single decoding thread.
 *
 * Three libraries share this file:
 *   _ref/libopenhevc_c.so    the untouched pure-C reference decoder (the bitstream-level oracle)
 *   _ref/libopenhevc_gen.so  + synth_gen.c: the stream synthesiser (reference parser driven by a random bin source)
 *   _ref/libopenhevc_hip.so  + hip_hooks.c: the reference front-end with its tables filled by libohevc_hip.so
 *                            (INTEGRATION.md applied by link-time interposition, no reference source is modified)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>

#include "libavcodec/avcodec.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "libavutil/mem.h"

#ifdef OHDEC_HIP
/* integration/hip_backend.h: one back end per decoder instance, attached before avcodec_open2 (the structs are passed through opaquely) */
typedef struct ohhip_backend ohhip_backend;
typedef struct ohdec_options { size_t struct_size; int device, bulk_filters, defer_download, pin_frames, async_issue, record_only, test_fail_index, flush_intra_kib, level_launch, device_filters, crash_backtrace; const char *trace_path; ohhip_backend *base_layer; int park_frames, own_frames, queue_download; } ohdec_options;   /* = ohhip_options, integration/hip_backend.h */
void ohhip_options_default(ohdec_options *o);
size_t ohhip_options_size(void);
ohhip_backend *ohhip_backend_new(const ohdec_options *o);
int  ohhip_backend_attach(ohhip_backend *be, AVCodecContext *avctx);
int  ohhip_backend_frame_done(ohhip_backend *be);
int  ohhip_backend_frame_failed(ohhip_backend *be);
int  ohhip_backend_fetch_output(ohhip_backend *be, uint8_t *const data[3], const int linesize[3]);
void ohhip_backend_pre_close(ohhip_backend *be);
void ohhip_backend_free(ohhip_backend *be);
int  ohhip_backend_frames_mode(ohhip_backend *be, const void *mode);
void ohhip_backend_frames_install(ohhip_backend *be, AVCodecContext *avctx);
int  ohhip_backend_frame_is_local(ohhip_backend *be, const unsigned char *data0);
#else
typedef struct ohhip_backend ohhip_backend;
typedef struct ohdec_options { size_t struct_size; int device, bulk_filters, defer_download, pin_frames, async_issue, record_only, test_fail_index, flush_intra_kib, level_launch, device_filters, crash_backtrace; const char *trace_path; ohhip_backend *base_layer; int park_frames, own_frames, queue_download; } ohdec_options;   /* = ohhip_options, integration/hip_backend.h */
static void ohhip_options_default(ohdec_options *o) { memset(o, 0, sizeof(*o)); o->struct_size = sizeof(*o); }
static size_t ohhip_options_size(void) { return sizeof(ohdec_options); }
static ohhip_backend *ohhip_backend_new(const ohdec_options *o) { (void)o; return NULL; }
static int  ohhip_backend_attach(ohhip_backend *be, AVCodecContext *avctx) { (void)be; (void)avctx; return 0; }
static int  ohhip_backend_frame_done(ohhip_backend *be) { (void)be; return 0; }
static int  ohhip_backend_frame_failed(ohhip_backend *be) { (void)be; return 0; }
static int  ohhip_backend_fetch_output(ohhip_backend *be, uint8_t *const data[3], const int linesize[3]) { (void)be; (void)data; (void)linesize; return 0; }
static void ohhip_backend_pre_close(ohhip_backend *be) { (void)be; }
static void ohhip_backend_free(ohhip_backend *be) { (void)be; }
static int  ohhip_backend_frames_mode(ohhip_backend *be, const void *mode) { (void)be; (void)mode; return -1; }
static void ohhip_backend_frames_install(ohhip_backend *be, AVCodecContext *avctx) { (void)be; (void)avctx; }
static int  ohhip_backend_frame_is_local(ohhip_backend *be, const unsigned char *data0) { (void)be; (void)data0; return 1; }
#endif

/* the reference's own decoded-picture-hash check (hevc.c:4146-4162) reports through av_log only: count its two messages */
static int g_md5_ok, g_md5_bad;
static void log_counter(void *avcl, int level, const char *fmt, va_list vl)
{
    if (fmt && !strncmp(fmt, "Correct MD5", 11))
        __sync_fetch_and_add(&g_md5_ok, 1);
    else if (fmt && !strncmp(fmt, "Incorrect MD5", 13))
        __sync_fetch_and_add(&g_md5_bad, 1);
    if (level <= AV_LOG_ERROR && !(fmt && !strncmp(fmt, "Incorrect MD5", 13)))
        av_log_default_callback(avcl, level, fmt, vl);
}

typedef struct ohdec {
    AVCodecContext *avctx;
    ohhip_backend  *backend;       /* this decoder's instance of the gfx950 back end (NULL in the CPU builds) */
    AVFrame        *frame;
    /* pipelined output (ohdec_set_pipelined): the application takes a picture one call late, so that with ONE decoding thread the device
     * reconstructs picture k while the CPU parses picture k + 1 (with OHHIP_DEFER_DOWNLOAD=1 the frame-end hook only issues the work) */
    AVFrame        *next, *held;
    int             pipelined, held_valid;
    uint8_t        *pkt_buf;
    int             pkt_cap;
    int             have_frame;
    int             threads;
    double          t_decode, t_fetch;     /* seconds of the calling thread inside avcodec_decode_video2 / inside the back end's fetch_output (ohdec_times) */
    long            n_calls;
} ohdec;

static double harness_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
/* where the application thread's time goes: [0] inside the decoder's call, [1] inside ohhip_backend_fetch_output (the wait for the picture's
 * device work + its copy-back), [2] calls - since the decoder was opened */
void ohdec_times(const ohdec *d, double out[3]) { out[0] = d->t_decode; out[1] = d->t_fetch; out[2] = (double)d->n_calls; }

/* thread_type: 1 frame threads, 2 slice/WPP threads, 3 both (the -f option of the reference's CLI, main_hm/getopt.c) */
/* checksum: the reference's `decode-checksum` option, set BEFORE avcodec_open2 the way main_hm/main.c does (libOpenHevcSetCheckMD5
 * between libOpenHevcInit and libOpenHevcStartDecoder, openHevcWrapper.c:429-440) so that frame-thread copies inherit it: every
 * picture is verified against its decoded-picture-hash SEI (hevc.c:4146-4162) */
/* `device` < 0: the back end's defaults (environment); else this decoder's HIP device.  Every decoder gets a back end of its own. */
/* SHVC (openHevcWrapper.c:47-103): the wrapper opens MAX_DECODERS = 2 decoders per handle, sets each one's "decoder-id" option before
 * avcodec_open2 and points the enhancement layer's BL_avcontext at the base layer's context.  `decoder_id` 0 / `base` NULL: a single-layer
 * decoder, as before.  With the gfx950 back end the enhancement-layer decoder's back end shares the base layer's picture store (the
 * inter-layer reference picture is resampled from a base-layer picture on the device). */
ohdec *ohdec_open_layer(int threads, int thread_type, int checksum, int device, int decoder_id, ohdec *base);
/* per-instance options of the NEXT decoder this thread opens (ohhip_options.level_launch / device_filters; -2: leave the default): how a test
 * gives two decoders of one process different options without touching the environment */
static __thread int t_next_level_launch = -2, t_next_device_filters = -2;
void ohdec_set_next_options(int level_launch, int device_filters) { t_next_level_launch = level_launch; t_next_device_filters = device_filters; }

ohdec *ohdec_open_dev(int threads, int thread_type, int checksum, int device) { return ohdec_open_layer(threads, thread_type, checksum, device, 0, NULL); }
ohdec *ohdec_open_ex(int threads, int thread_type, int checksum) { return ohdec_open_dev(threads, thread_type, checksum, -1); }
ohdec *ohdec_open_layer(int threads, int thread_type, int checksum, int device, int decoder_id, ohdec *base)
{
    static int registered;
    ohdec *d = calloc(1, sizeof(*d));
    AVCodec *codec;
    if (!d)
        return NULL;
    if (!registered) {
        avcodec_register_all();
        registered = 1;
    }
    av_log_set_level(AV_LOG_ERROR);
    codec = avcodec_find_decoder(AV_CODEC_ID_HEVC);
    if (!codec)
        goto fail;
    d->avctx = avcodec_alloc_context3(codec);
    d->frame = av_frame_alloc();
    d->next = av_frame_alloc();
    d->held = av_frame_alloc();
    if (!d->avctx || !d->frame || !d->next || !d->held)
        goto fail;
    d->avctx->flags |= CODEC_FLAG_UNALIGNED;
    d->avctx->err_recognition |= AV_EF_EXPLODE;     /* a syntax error fails the call instead of being concealed (hevc.c:3480) */
    av_opt_set(d->avctx, "thread_type", thread_type == 2 ? "slice" : thread_type >= 3 ? "frameslice" : "frame", 0);
    d->threads = threads > 0 ? threads : 1;
    av_opt_set_int(d->avctx, "threads", d->threads, 0);
    if (getenv("OHDEC_DEBUG_THREADS")) {                                         /* the reference's own thread-synchronisation log lines */
        d->avctx->debug |= FF_DEBUG_THREADS;
        av_log_set_level(AV_LOG_DEBUG);
    }
    av_opt_set_int(d->avctx->priv_data, "decoder-id", decoder_id, 0);          /* openHevcWrapper.c:92 */
    d->avctx->quality_id = base || decoder_id ? 1 : 0;                          /* the wrapper's active_layer (openHevcWrapper.c:120) */
    if (checksum) {
        av_opt_set_int(d->avctx->priv_data, "decode-checksum", 1, 0);
        g_md5_ok = g_md5_bad = 0;
        av_log_set_level(AV_LOG_INFO);
        av_log_set_callback(log_counter);
    } else {
        av_log_set_callback(av_log_default_callback);
    }
#ifdef OHDEC_HIP
    {
        ohdec_options o;
        if (ohhip_options_size() != sizeof(o)) {       /* this file mirrors ohhip_options (integration/hip_backend.h) */
            fprintf(stderr, "decoder_harness: ohhip_options is %zu bytes in the back end, %zu here\n", ohhip_options_size(), sizeof(o));
            abort();
        }
        ohhip_options_default(&o);
        if (device >= 0)
            o.device = device;
        if (t_next_level_launch != -2) o.level_launch = t_next_level_launch;
        if (t_next_device_filters != -2) o.device_filters = t_next_device_filters;
        t_next_level_launch = t_next_device_filters = -2;
        /* OHDEC_SHVC_SEPARATE_STORES=1 (a test): the integration mistake of two unrelated back ends - must fail loudly, not decode garbage */
        o.base_layer = base && !getenv("OHDEC_SHVC_SEPARATE_STORES") ? base->backend : NULL;
        if (!(d->backend = ohhip_backend_new(&o)) || ohhip_backend_attach(d->backend, d->avctx) != 0)
            goto fail;
    }
#else
    (void)device;
#endif
    /* openHevcWrapper.c:100-108: decoder i is opened, THEN decoder i + 1's BL_avcontext is set, then decoder i + 1 is opened - i.e. before its
     * avcodec_open2, which is what lets its frame-thread copies inherit the pointer (frame_thread_init copies the context; update_context_from_user,
     * pthread_frame.c:262-300, does not carry this field) */
    if (base)
        d->avctx->BL_avcontext = base->avctx;
    if (avcodec_open2(d->avctx, codec, NULL) < 0)
        goto fail;
    return d;
fail:
    ohhip_backend_free(d->backend);
    if (d->frame)
        av_frame_free(&d->frame);
    if (d->avctx)
        av_free(d->avctx);
    free(d);
    return NULL;
}

ohdec *ohdec_open(int threads, int thread_type) { return ohdec_open_ex(threads, thread_type, 0); }

/* pipelined output: see struct ohdec.  Call right after ohdec_open*. */
void ohdec_set_pipelined(ohdec *d, int on) { d->pipelined = on != 0; }

/* Frame-parallel decoding over processes (integration/hip_frames.h; openhevc_amd/dist.py drives it): `mode` = ohhip_frames_mode *,
 * NULL switches it off.  Call right after ohdec_open*; one decoding thread per process.  Returns 0, or -1 for decoders without
 * the hooks. */
int ohdec_frames_mode(ohdec *d, const void *mode)
{
    if (ohhip_backend_frames_mode(d->backend, mode) != 0)
        return -1;
    if (mode)
        ohhip_backend_frames_install(d->backend, d->avctx);
    return 0;
}

/* the picture ohdec_decode / ohdec_flush just returned: 1 if this process reconstructed it (its samples are valid here) */
int ohdec_frame_is_local(ohdec *d)
{
    return d->have_frame ? ohhip_backend_frame_is_local(d->backend, d->frame->data[0]) : 0;
}

void ohdec_md5_results(ohdec *d, int *ok, int *bad)
{
    (void)d;
    *ok = g_md5_ok;
    *bad = g_md5_bad;
}

/* SHVC: the wrapper's active_layer, written into every decoder's quality_id before each access unit (openHevcWrapper.c:120) */
void ohdec_set_active_layer(ohdec *d, int layer) { d->avctx->quality_id = layer; }

/* SHVC: what libOpenHevcDecode does between the two decoders of an access unit (openHevcWrapper.c:131-132): the enhancement-layer decoder
 * is handed the base-layer picture the base-layer decoder has just reconstructed (its s->ref, hevc.c:3249). */
void ohdec_take_base_frame(ohdec *el, ohdec *bl)
{
    el->avctx->BL_frame = bl->avctx->BL_frame;
}

/* returns 1 when a picture came out (fetch it with ohdec_frame_*), 0 when none, <0 on a decoder / back-end error */
int ohdec_decode(ohdec *d, const uint8_t *au, int len, int64_t pts)
{
    AVPacket pkt;
    int got = 0, ret;

    if (len + FF_INPUT_BUFFER_PADDING_SIZE > d->pkt_cap) {
        free(d->pkt_buf);
        d->pkt_cap = len + FF_INPUT_BUFFER_PADDING_SIZE + 4096;
        d->pkt_buf = malloc(d->pkt_cap);
        if (!d->pkt_buf)
            return -1;
    }
    if (len)
        memcpy(d->pkt_buf, au, len);
    memset(d->pkt_buf + len, 0, FF_INPUT_BUFFER_PADDING_SIZE);

    av_init_packet(&pkt);
    pkt.data = len ? d->pkt_buf : NULL;
    pkt.size = len;
    pkt.pts  = pts;
    if (d->pipelined) {
        int out = 0;
        av_frame_unref(d->next);
        ret = avcodec_decode_video2(d->avctx, d->next, &got, &pkt);
        if (ret < 0) {
            ohhip_backend_frame_failed(d->backend);
            return -2;
        }
        if (ohhip_backend_frame_done(d->backend) < 0)
            return -3;
        av_frame_unref(d->frame);
        if (d->held_valid) {                          /* the picture of the previous call: its device work had a whole parse to finish */
            if (ohhip_backend_fetch_output(d->backend, d->held->data, d->held->linesize) < 0)
                return -3;
            av_frame_move_ref(d->frame, d->held);
            d->held_valid = 0;
            out = 1;
        }
        if (got) {
            av_frame_move_ref(d->held, d->next);
            d->held_valid = 1;
        }
        d->have_frame = out;
        return out;
    }
    av_frame_unref(d->frame);
    {
        const double t0 = harness_now();
        ret = avcodec_decode_video2(d->avctx, d->frame, &got, &pkt);
        d->t_decode += harness_now() - t0;
        d->n_calls++;
    }
    if (ret < 0) {
        ohhip_backend_frame_failed(d->backend);         /* the open frame is aborted and, in frames mode, published as failed */
        return -2;
    }
    /* "frame complete, before output" (INTEGRATION.md section 3): a no-op for the CPU builds */
    if (ohhip_backend_frame_done(d->backend) < 0)
        return -3;
    /* the application takes the picture: with a deferred copy-back this is where its samples reach the host */
    {
        const double t0 = harness_now();
        if (got && ohhip_backend_fetch_output(d->backend, d->frame->data, d->frame->linesize) < 0)
            return -3;
        d->t_fetch += harness_now() - t0;
    }
    d->have_frame = got;
    return got ? 1 : 0;
}

/* drain the decoder: call until it returns 0.  With frame threads an empty packet collects one worker per call and a
 * worker may have nothing to show (pthread_frame.c), so "nothing" only counts after every worker has been asked. */
int ohdec_flush(ohdec *d)
{
    int i, r = 0, workers = d->threads;
    if (d->avctx->thread_count_frame > workers)     /* frame + slice threads: the frame-thread count is derived from the core count (pthread.c:65-71) */
        workers = d->avctx->thread_count_frame;
    for (i = 0; i <= workers + (d->pipelined ? 1 : 0) && r == 0; i++)
        r = ohdec_decode(d, NULL, 0, 0);
    return r;
}

int ohdec_frame_info(ohdec *d, int *w, int *h, int *bit_depth, int *chroma_w_shift, int *chroma_h_shift)
{
    const AVPixFmtDescriptor *desc;
    if (!d->have_frame)
        return -1;
    desc = av_pix_fmt_desc_get(d->frame->format);
    *w = d->frame->width;
    *h = d->frame->height;
    *bit_depth = desc->comp[0].depth_minus1 + 1;
    *chroma_w_shift = desc->log2_chroma_w;
    *chroma_h_shift = desc->log2_chroma_h;
    return 0;
}

/* tightly packed copy of plane c (bytes per sample = 1 or 2) */
int ohdec_frame_copy(ohdec *d, int c, uint8_t *dst)
{
    const AVPixFmtDescriptor *desc;
    int w, h, bps, y;
    if (!d->have_frame)
        return -1;
    desc = av_pix_fmt_desc_get(d->frame->format);
    bps = desc->comp[0].depth_minus1 >= 8 ? 2 : 1;
    w = d->frame->width;
    h = d->frame->height;
    if (c) {
        w = -((-w) >> desc->log2_chroma_w);
        h = -((-h) >> desc->log2_chroma_h);
    }
    for (y = 0; y < h; y++)
        memcpy(dst + (size_t)y * w * bps, d->frame->data[c] + (size_t)y * d->frame->linesize[c], (size_t)w * bps);
    return 0;
}

void ohdec_close(ohdec *d)
{
    if (!d)
        return;
    ohhip_backend_pre_close(d->backend);          /* page locks of the frame buffers go before the buffers do */
    avcodec_close(d->avctx);
    av_free(d->avctx);
    av_frame_free(&d->frame);
    av_frame_free(&d->next);
    av_frame_free(&d->held);
    ohhip_backend_free(d->backend);
    free(d->pkt_buf);
    free(d);
}
