/*
 * integration/hip_hooks.c -- the reference-side half of the drop-in: INTEGRATION.md as code.
 *
 * This file is what a maintainer of openHEVC adds to the decoder (it includes the reference's own headers and is compiled
 * with the reference's flags); together with libohevc_hip.so (include/ohevc_tables.h) it turns the CPU decoder into one
 * whose every pixel is produced by the gfx950 kernels.  Because /root/reference is read-only, the patch is applied at LINK
 * time instead of by editing sources: ONE translation unit (libavcodec/hevc.c) is compiled with call-site renames
 * (-Dff_hevc_dsp_init=ohhip_hevc_dsp_init etc., integration/renames.mk) so that the calls at hevc.c:421-423 (table fill
 * in set_sps), hevc.c:3245/3250 (ff_hevc_set_new_ref / ff_hevc_frame_rps in hevc_frame_start), hevc.c:4148 (right before
 * the decoded-picture-hash check) ... land here.  Each wrapper calls the reference's function and then the libohevc_hip.so
 * hook, i.e. exactly the patch INTEGRATION.md sections 1-3 describes.
 *
 * Links against libohevc_hip.so ONLY: no oracle, no CPU pixel path (`nm -u` of the resulting decoder shows no ohor_* /
 * ohsw_* symbol; tests/test_abi_cpu.py checks that).  Everything the CPU front-end keeps doing (CABAC, MV/merge
 * derivation, bS, SAO parameter parsing, DPB management) is the reference's.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <limits.h>
#include <pthread.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include <stddef.h>

#include "libavcodec/hevc.h"
#include "libavcodec/thread.h"
#include "libavutil/pixdesc.h"
#include "libavutil/buffer.h"
#include "libavutil/buffer_internal.h"      /* struct AVBuffer, BufferPoolEntry: the page locks of decoder-owned frame buffers are dropped where the pool FREES them (guard_pool_entry) */

#include "ohevc_tables.h"
#include "ohevc_debug.h"
#include "hip_backend.h"
#include "hip_frames.h"

/* ---- per-instance state (design 1).  Everything that used to hang off file scope lives in one ohhip_backend per decoder
 * instance; the hooks reach it through the decoder context they are handed (s->avctx->opaque: pthread_frame.c:276 copies it into every
 * frame-thread context) after checking it against the registry of live back ends.  File scope keeps only: that registry, the process
 * default for decoders nobody attached a back end to, and process-wide profiling counters. */
#define MAX_BUFS OHEVC_MAX_PICTURES      /* frame buffers of one decoder = pictures of its store (ohevc_ctx.h): the store's limit, not a second one */
#define MAX_TRACE 8192
typedef struct ohhip_buf {
    const uint8_t *data0;
    int slot, w, h, bd, fmt;
    int poc, seq;              /* the picture that lives in the buffer: HEVCFrame.poc / .sequence when it was registered */
    ohevc_ctx *ctx;            /* the context that is reconstructing (or last reconstructed) this picture */
    /* frame-parallel decoding over processes (hip_frames.h): decoding-order index, owned elsewhere, what has arrived */
    int index, remote, have_motion, have_planes;
    int have_rows;             /* remote picture: luma rows 0 .. have_rows - 1 are in this process's store (bands of the transport, await_rows) */
    /* the page-locked allocations behind this frame (AVFrame.buf[i]): unpinned when the address comes back with another geometry */
    void *pin_ptr[3];
    size_t pin_bytes[3];
    int copy_queued;           /* the picture's copy-back was queued at its frame end (queue_download): fetch_output only waits for it */
} ohhip_buf;

typedef struct ohhip_trace_rec { int tid, poc; double t_start, t_hook, t_issued, t_end; } ohhip_trace_rec;

struct ohhip_backend {
    uint32_t           magic;
    unsigned           id;             /* never reused: thread-local caches name (pointer, id) */
    ohhip_options      opt;
    ohevc_ctx         *root;           /* owns the picture store; one context per decoding thread shares it (ohevc_ctx_create_shared) */
    ohevc_ctx        **all;                      /* every per-thread context this instance made (grows) */
    int                cap_all;
    int                nall;
    ohevc_ctx         *spare[64];      /* contexts made at attach time, one per decoding thread the decoder will start: a context is a stream,
                                          events and page-locked buffers - milliseconds of driver calls that would otherwise sit in front of
                                          each thread's first picture */
    int                nspare;
    pthread_mutex_t    lock;
    volatile int       error;          /* failures seen by threads that do not own a picture (slice workers) */
    int                async_used;     /* some frame end went through the issuer: fetch_output waits for copy-backs (several decoding threads set it: atomic accesses) */
    double             issuer_s0;
    long long          issuer_f0;
    ohhip_buf          bufs[MAX_BUFS];
    int                nbufs;
    /* frames mode (hip_frames.h) */
    ohhip_frames_mode  fm;
    int                fm_on, fm_index, fm_segment;       /* fm_segment: IDR pictures seen so far - 1 (segment_ownership) */
    int (*execute)(AVCodecContext *, int (*)(AVCodecContext *, void *), void *, int *, int, int);
    int (*execute2)(AVCodecContext *, int (*)(AVCodecContext *, void *, int, int), void *, int *, int);
    /* OHHIP_TRACE_FRAMES: host timeline of every picture */
    ohhip_trace_rec   *trace;
    int                ntrace;
    int                pinned_blocks;  /* this back end's contexts make page-locked blocks (a record-only back end: plain memory) */
    int                nthreads;       /* decoding threads of the decoder this back end is attached to */
    pthread_t          prefetch_thread;            /* own_frames: makes the first blocks ahead of the decoder's requests (pool_prefetch_run) */
    int                prefetch_started, prefetch_size, prefetch_want, prefetch_cap;
    struct ohhip_frame_pool *pool;     /* own_frames: the page-locked blocks this decoder's frame buffers are made of (outlives the back end while blocks are out) */
    struct { int w, h, fmt, planes, linesize[4], size[4]; ptrdiff_t off[4]; } layouts[4];      /* frame layouts seen (ohhip_get_buffer2) */
    int                nlayouts;
    struct ohhip_backend *next;
};
#define OHHIP_MAGIC 0x6f686862u

/* ---- own_frames: the decoder's frame buffers out of page-locked memory of the back end's own ----
 * Blocks are recycled by exact size (a decoder has one size per plane) and given back to the runtime when the back end is freed; a block the
 * application still holds then (a frame it has not released) frees itself when it is released.  The pool object outlives the back end for that. */
typedef struct ohhip_block { void *ptr; int size; int out; struct ohhip_frame_pool *pool; struct ohhip_block *next; } ohhip_block;      /* out: plane buffers of it the decoder / the application still holds */
typedef struct ohhip_frame_pool {
    pthread_mutex_t m;
    int alive, refs;                   /* refs: the back end + every block that is out */
    int stop_prefetch;
    ohhip_block *free_list;
    ohhip_block **blocks;              /* every block that exists (free or out): whose buffer is it? (pool_owns: pointer comparison only) */
    int nblocks, cap_blocks;
    long long bytes;
} ohhip_frame_pool;

/* Blocks outlive their decoder in a process-wide cache (by size and kind; OHHIP_BLOCK_CACHE_MB, default 1024, 0: none): an application that
 * opens a decoder after closing one - a seek, the next file of a play-list, every repetition of a benchmark - gets its frame buffers without
 * forty page-locking calls in front of its first pictures (0.33 ms per 1080p frame, 7 ms per 8K frame: profiles/r6zc_host_block_probe.txt).
 * ohhip_frame_pool_trim() gives the cache back to the runtime. */
static pthread_mutex_t g_cache_lock = PTHREAD_MUTEX_INITIALIZER;
static struct { void *ptr; int size; } *g_cache;
static int g_ncache, g_cap_cache;
static long long g_cache_bytes, g_cache_limit = -1;
static void *cache_take(int size, int pinned)
{
    void *ptr = NULL;
    int i;
    pthread_mutex_lock(&g_cache_lock);
    for (i = g_ncache - 1; i >= 0 && !ptr; i--)
        if (g_cache[i].size == size && ohevc_host_block_pinned(g_cache[i].ptr) == pinned) {
            ptr = g_cache[i].ptr;
            g_cache[i] = g_cache[--g_ncache];
            g_cache_bytes -= size;
        }
    pthread_mutex_unlock(&g_cache_lock);
    return ptr;
}
static void cache_put(void *ptr, int size)         /* or back to the runtime, when the cache is full */
{
    int kept = 0;
    pthread_mutex_lock(&g_cache_lock);
    if (g_cache_limit < 0) {
        const char *v = getenv("OHHIP_BLOCK_CACHE_MB");
        g_cache_limit = (v && v[0] ? atoll(v) : 1024) << 20;
    }
    if (g_cache_bytes + size <= g_cache_limit) {
        if (g_ncache == g_cap_cache) {
            const int cap = g_cap_cache ? 2 * g_cap_cache : 64;
            void *grown = realloc(g_cache, (size_t)cap * sizeof(*g_cache));
            if (grown) { g_cache = grown; g_cap_cache = cap; }
        }
        if (g_ncache < g_cap_cache) {
            g_cache[g_ncache].ptr = ptr; g_cache[g_ncache].size = size; g_ncache++;
            g_cache_bytes += size;
            kept = 1;
        }
    }
    pthread_mutex_unlock(&g_cache_lock);
    if (!kept)
        ohevc_host_free(ptr);
}
void ohhip_frame_pool_trim(void)
{
    pthread_mutex_lock(&g_cache_lock);
    while (g_ncache > 0)
        ohevc_host_free(g_cache[--g_ncache].ptr);
    g_cache_bytes = 0;
    pthread_mutex_unlock(&g_cache_lock);
}

static long long g_pool_made, g_pool_live;          /* blocks ever made / existing now, process-wide (ohhip_frame_pool_counts: tests, leak checks) */
void ohhip_frame_pool_counts(long long *made, long long *live)
{
    if (made) *made = __atomic_load_n(&g_pool_made, __ATOMIC_RELAXED);
    if (live) *live = __atomic_load_n(&g_pool_live, __ATOMIC_RELAXED);
}

static void pool_forget_locked(ohhip_frame_pool *p, ohhip_block *b)
{
    __atomic_fetch_sub(&g_pool_live, 1, __ATOMIC_RELAXED);
    int i;
    for (i = 0; i < p->nblocks; i++)
        if (p->blocks[i] == b) { p->blocks[i] = p->blocks[--p->nblocks]; break; }
    p->bytes -= b->size;
}
static void pool_destroy(ohhip_frame_pool *p) { pthread_mutex_destroy(&p->m); free(p->blocks); free(p); }

/* the AVBuffer free callback of a plane of a block; the last one puts the block back onto the free list, or - the back end is gone - gives it
 * back to the runtime */
static void pool_release(void *opaque, uint8_t *data)
{
    ohhip_block *b = opaque;
    ohhip_frame_pool *p = b->pool;
    int dead, last;
    (void)data;
    if (__atomic_sub_fetch(&b->out, 1, __ATOMIC_ACQ_REL) > 0)
        return;
    pthread_mutex_lock(&p->m);
    dead = !p->alive;
    if (dead)
        pool_forget_locked(p, b);
    else {
        b->next = p->free_list;
        p->free_list = b;
    }
    last = --p->refs == 0;
    pthread_mutex_unlock(&p->m);
    if (dead) {
        cache_put(b->ptr, b->size);
        free(b);
    }
    if (last)
        pool_destroy(p);
}

/* a new block of the pool (listed in blocks[]); `out`: handed to the caller (counted in refs) instead of put onto the free list */
static ohhip_block *pool_make_block(ohhip_frame_pool *p, ohevc_ctx *ctx, int size, int out, int pinned)
{
    ohhip_block *b;
    void *mem = NULL;
    /* (not cleared, unlike the decoder's own pool - av_buffer_allocz, utils.c:558-560: every sample of the picture area arrives by the
     * copy-back, the edge around it is read by nobody - motion compensation runs on the device picture) */
    if (!(mem = cache_take(size, pinned)) && ohevc_host_alloc(ctx, (size_t)size, &mem) != OHEVC_OK)
        return NULL;
    if (!(b = calloc(1, sizeof(*b)))) { cache_put(mem, size); return NULL; }
    b->ptr = mem; b->size = size; b->pool = p;
    pthread_mutex_lock(&p->m);
    if (p->nblocks == p->cap_blocks) {
        const int cap = p->cap_blocks ? 2 * p->cap_blocks : 64;
        ohhip_block **grown = realloc(p->blocks, (size_t)cap * sizeof(*grown));
        if (!grown) { pthread_mutex_unlock(&p->m); cache_put(mem, size); free(b); return NULL; }
        p->blocks = grown; p->cap_blocks = cap;
    }
    p->blocks[p->nblocks++] = b;
    p->bytes += size;
    __atomic_fetch_add(&g_pool_made, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_pool_live, 1, __ATOMIC_RELAXED);
    if (out)
        p->refs++;
    else {
        b->next = p->free_list;
        p->free_list = b;
    }
    pthread_mutex_unlock(&p->m);
    return b;
}

/* one block = one frame (its planes back to back, each at a multiple of 64 bytes): ONE allocation per frame buffer the decoder ever needs */
static ohhip_block *pool_get(ohhip_backend *be, int size)
{
    ohhip_frame_pool *p = be->pool;
    ohhip_block *b = NULL, **pp;
    pthread_mutex_lock(&p->m);
    for (pp = &p->free_list; *pp; pp = &(*pp)->next)
        if ((*pp)->size == size) { b = *pp; *pp = b->next; break; }
    if (b)
        p->refs++;
    pthread_mutex_unlock(&p->m);
    if (!b && !(b = pool_make_block(p, be->root, size, 1, be->pinned_blocks)))
        return NULL;
    b->out = 1;                         /* the caller's own hold: dropped (pool_release) once the plane buffers are made */
    return b;
}

/* Page-locking a frame's worth of memory costs the runtime a fraction of a millisecond (at 8K: several), and a frame thread asks for its frame
 * in the picture's serial prologue (hevc_frame_start, in front of ff_thread_finish_setup; get_buffer under the decoder's buffer_mutex,
 * pthread_frame.c:902): during a decoder's first pictures every frame start waited for one such allocation - a fresh decoder's first pass
 * at 16 frame threads ran at 2200-2400 pictures a second where the round-5 form (page locks taken on the decoder's own buffers) reached
 * 2600-2800 (profiles/r6za_*).  So the first frame of a geometry starts a helper that makes the blocks the decoder is about to ask for -
 * one per decoding thread and a picture buffer's worth - while the threads parse; it ends when it has made them or the back end is freed. */
static void *pool_prefetch_run(void *arg)
{
    ohhip_backend *be = arg;
    ohhip_frame_pool *p = be->pool;
    int made = 0;
    while (made < be->prefetch_want) {
        int go;
        pthread_mutex_lock(&p->m);
        go = p->alive && !p->stop_prefetch && p->nblocks < be->prefetch_cap;
        pthread_mutex_unlock(&p->m);
        if (!go || !pool_make_block(p, be->root, be->prefetch_size, 0, be->pinned_blocks))
            break;
        made++;
    }
    return NULL;
}

static void pool_prefetch_start(ohhip_backend *be, int size)        /* be->lock held (once per back end: the first geometry) */
{
    long long budget = 2048ll << 20;            /* at most 2 GiB ahead of demand */
    int want = be->nthreads + 8;
    if (be->prefetch_started || !be->pool || be->opt.record_only)
        return;
    if ((long long)want * size > budget)
        want = (int)(budget / size);
    if (want < 2)
        return;
    be->prefetch_size = size;
    be->prefetch_want = want;
    be->prefetch_cap = want + 4;
    if (pthread_create(&be->prefetch_thread, NULL, pool_prefetch_run, be) == 0)
        be->prefetch_started = 1;
}

static void pool_prefetch_stop(ohhip_backend *be)                    /* before the contexts go (ohhip_backend_free) */
{
    if (!be->prefetch_started)
        return;
    pthread_mutex_lock(&be->pool->m);
    be->pool->stop_prefetch = 1;
    pthread_mutex_unlock(&be->pool->m);
    pthread_join(be->prefetch_thread, NULL);
    be->prefetch_started = 0;
}

/* is this AVBufferRef one of the pool's blocks?  (its opaque pointer is compared, never followed) */
static int pool_owns(ohhip_backend *be, const AVBufferRef *ref)
{
    ohhip_frame_pool *p = be->pool;
    const void *o;
    int i, found = 0;
    if (!p || !ref)
        return 0;
    o = av_buffer_get_opaque(ref);
    pthread_mutex_lock(&p->m);
    for (i = 0; i < p->nblocks && !found; i++)
        found = p->blocks[i] == o && (uint8_t *)p->blocks[i]->ptr <= ref->data && ref->data < (uint8_t *)p->blocks[i]->ptr + p->blocks[i]->size;
    pthread_mutex_unlock(&p->m);
    return found;
}

static void pool_close(ohhip_backend *be)       /* ohhip_backend_free: after the contexts (their streams have drained) */
{
    ohhip_frame_pool *p = be->pool;
    ohhip_block *list, *b;
    int last;
    if (!p)
        return;
    be->pool = NULL;
    pthread_mutex_lock(&p->m);
    p->alive = 0;
    list = p->free_list;
    p->free_list = NULL;
    for (b = list; b; b = b->next)
        pool_forget_locked(p, b);
    last = --p->refs == 0;
    pthread_mutex_unlock(&p->m);
    while ((b = list)) {
        list = b->next;
        cache_put(b->ptr, b->size);
        free(b);
    }
    if (last)
        pool_destroy(p);
}

static pthread_mutex_t     g_reg_lock = PTHREAD_MUTEX_INITIALIZER;
static ohhip_backend      *g_backends;             /* live back ends */
static ohhip_backend      *g_default;              /* for decoders without an attached back end (ohdec_backend_open / first use) */
static volatile unsigned   g_epoch = 1;            /* bumped whenever the registry changes: invalidates the per-thread look-up caches */
static unsigned            g_next_id = 1;
/* process-wide profiling counters (all instances; ohdec_backend_profile) */
static pthread_mutex_t     g_prof_lock = PTHREAD_MUTEX_INITIALIZER;
static double              g_end_frame_s;      /* wall time inside the frame-end hook (upload, launches, drain, copy-back) */
static long long           g_counts[8];        /* frames, launches, tu, mc, intra, dbk, sao jobs, upload bytes */
static long long           g_alg_bytes;        /* algorithmic HBM bytes of the recorded jobs (ohevc_frame_stats.alg_bytes), same period */

/* ---- what the calling thread is doing right now ---- */
static __thread ohhip_backend *t_be;   /* the instance of this thread's open frame */
static __thread ohevc_ctx *t_ctx;
static __thread int        t_frame_open;
static __thread int        t_error;    /* a hook of the OPEN frame failed on this thread: reported by this picture's frame end, nobody else's */
static __thread HEVCContext *t_s;      /* the decoder context this thread's open frame belongs to */
static __thread ThreadFrame *t_bl_tf;   /* SHVC: the base-layer picture's ThreadFrame of the picture this thread is parsing (ohhip_cabac_init, set_new_ref) */
static __thread int        t_remote;       /* the picture being parsed is reconstructed by another process: skip its slice data */
static __thread int        t_publish;      /* the open frame is exchanged at its end (index of its bufs entry + 1) */
static __thread int        t_publish_index;        /* its decoding-order index and the size of its motion field: what a failure report needs */
static __thread size_t     t_publish_mvf_bytes;
static __thread double     t_trace_start;
static __thread int        t_trace_poc;
/* this thread's context in each instance it has decoded for (an application thread may drive several one-thread decoders in turn) */
static __thread struct { ohhip_backend *be; unsigned id; ohevc_ctx *ctx; } t_ctxs[4];
/* look-up cache: the decoder context this thread was last asked about */
static __thread struct { const void *avctx; unsigned epoch; ohhip_backend *be; } t_seen;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static ohhip_backend *default_backend(void);

/* the instance a decoder context belongs to: avctx->opaque if it names a live back end, else the process default */
static ohhip_backend *backend_of(const AVCodecContext *avctx)
{
    ohhip_backend *b, *found = NULL;
    /* the epoch BEFORE the lookup: a back end freed or made between the lookup and the stamp must leave a stale stamp, not a stale pointer
     * under the current one (ohhip_cabac_init reads it the same way) */
    const unsigned epoch = g_epoch;
    if (avctx && t_seen.avctx == avctx && t_seen.epoch == epoch)
        return t_seen.be;
    pthread_mutex_lock(&g_reg_lock);
    for (b = g_backends; b && avctx; b = b->next)
        if ((void *)b == avctx->opaque)
            found = b;
    pthread_mutex_unlock(&g_reg_lock);
    if (!found)
        found = default_backend();
    if (avctx) {
        t_seen.avctx = avctx; t_seen.epoch = epoch; t_seen.be = found;
    }
    return found;
}

static void note_error(ohhip_backend *be)
{
    /* the picture's own thread marks its picture; anybody else (slice workers record into a picture they do not own) marks the instance */
    if (t_frame_open)
        t_error = 1;
    else if (be)
        be->error = 1;
}

int ohdec_backend_frame_done(void);

/* frames mode: every rank issues exactly one collective per exchanged picture.  A picture its owner cannot complete is published all the
 * same, marked failed (hip_frames.h), so that the subscribers' receives complete and their waits fail at once instead of hanging. */
static void publish_failed(ohhip_backend *be)
{
    const int i = t_publish - 1;
    t_publish = 0;
    if (i < 0 || !be || !be->fm_on)
        return;
    if (be->fm.publish(be->fm.user, t_publish_index, t_ctx, be->bufs[i].slot, NULL, t_publish_mvf_bytes, 1) != 0)
        fprintf(stderr, "ohhip: publishing the failure of picture %d failed: %s\n", t_publish_index, ohevc_last_error());
}

/* this thread's context in instance `be` (created on first use; all of them share the root's picture store) */
static void apply_ctx_options(const ohhip_backend *be, ohevc_ctx *ctx);
static ohevc_ctx *new_thread_ctx(ohhip_backend *be)
{
    ohevc_ctx *ctx = NULL;
    if (ohevc_ctx_create_shared(&ctx, be->opt.device, be->root) != OHEVC_OK)
        return NULL;
    /* stay bit-identical with the CTB lag of hevc_filter.c:1027-1063 (see ohevc_tables.h) */
    if (ohevc_tables_bind(ctx) != OHEVC_OK || ohevc_tables_emulate_filter_lag(ctx, 1) != OHEVC_OK) {
        ohevc_ctx_destroy(ctx);
        return NULL;
    }
    apply_ctx_options(be, ctx);
    pthread_mutex_lock(&be->lock);
    if (be->nall == be->cap_all) {
        const int cap = be->cap_all ? 2 * be->cap_all : 32;
        ohevc_ctx **grown = realloc(be->all, (size_t)cap * sizeof(*grown));
        if (!grown) {                           /* a context the instance cannot list would never be destroyed: fail instead */
            pthread_mutex_unlock(&be->lock);
            ohevc_ctx_destroy(ctx);
            fprintf(stderr, "ohhip: out of memory listing a per-thread context\n");
            return NULL;
        }
        be->all = grown;
        be->cap_all = cap;
    }
    be->all[be->nall++] = ctx;
    pthread_mutex_unlock(&be->lock);
    return ctx;
}

static ohevc_ctx *thread_ctx(ohhip_backend *be)
{
    int k, free_k = -1;
    ohevc_ctx *ctx = NULL;
    if (!be || !be->root)
        return NULL;
    for (k = 0; k < 4; k++) {
        if (t_ctxs[k].be == be && t_ctxs[k].id == be->id)
            return t_ctxs[k].ctx;
        if (free_k < 0 && !t_ctxs[k].be)
            free_k = k;
    }
    if (free_k < 0) {             /* four other instances used this thread before: forget the dead ones, else the oldest */
        unsigned live[4] = {0, 0, 0, 0};
        ohhip_backend *b;
        pthread_mutex_lock(&g_reg_lock);
        for (b = g_backends; b; b = b->next)
            for (k = 0; k < 4; k++)
                if (t_ctxs[k].be == b && t_ctxs[k].id == b->id)
                    live[k] = 1;
        for (k = 0; k < 4 && free_k < 0; k++)
            if (!live[k])
                free_k = k;
        if (free_k < 0) {
            /* all four belong to live instances: entry 0 goes, and its context goes back to its instance's spare list (under the registry lock:
             * the instance cannot die meanwhile) - whoever next needs a context for that instance takes it from there instead of making one */
            ohhip_backend *old = t_ctxs[0].be;
            free_k = 0;
            pthread_mutex_lock(&old->lock);
            if (old->nspare < (int)(sizeof(old->spare) / sizeof(old->spare[0])))
                old->spare[old->nspare++] = t_ctxs[0].ctx;
            pthread_mutex_unlock(&old->lock);
        }
        pthread_mutex_unlock(&g_reg_lock);
        t_ctxs[free_k].be = NULL;
    }
    pthread_mutex_lock(&be->lock);
    if (be->nspare > 0)
        ctx = be->spare[--be->nspare];                              /* (registered in be->all when it was made) */
    pthread_mutex_unlock(&be->lock);
    if (!ctx && !(ctx = new_thread_ctx(be))) {
        fprintf(stderr, "ohhip: per-thread context failed: %s\n", ohevc_last_error());
        note_error(be);
        return NULL;
    }
    t_ctxs[free_k].be = be; t_ctxs[free_k].id = be->id; t_ctxs[free_k].ctx = ctx;
    return ctx;
}

/* ---- the reference's own entry points (their call sites in hevc.c were renamed, the definitions were not) ---- */
void ohhip_hevc_dsp_init(HEVCDSPContext *c, int bit_depth)
{
    ff_hevc_dsp_init(c, bit_depth);                                 /* hevcdsp.c:1071 */
    ohevc_hevcdsp_init_hip((ohevc_HEVCDSPContext *)c, bit_depth);    /* INTEGRATION.md section 1 */
}

void ohhip_videodsp_init(VideoDSPContext *c, int bpc)
{
    ff_videodsp_init(c, bpc);                                       /* videodsp.c:38 */
    ohevc_videodsp_init_hip((ohevc_VideoDSPContext *)c, bpc);
}

/* intra_pred takes the decoder context, so the stub lives on the reference side */
static void intra_pred_hip(HEVCContext *s, int x0, int y0, int log2_size, int c_idx)
{
    const HEVCLocalContext *lc = s->HEVClc;
    ohevc_intra_geom g;
    memset(&g, 0, sizeof(g));
    g.width                    = s->sps->width;
    g.height                   = s->sps->height;
    g.chroma_format_idc        = s->sps->chroma_array_type ? s->sps->chroma_array_type : 1;
    g.log2_ctb_size            = s->sps->log2_ctb_size;
    g.log2_min_tb_size         = s->sps->log2_min_tb_size;
    g.strong_intra_smoothing   = s->sps->sps_strong_intra_smoothing_enable_flag;
    g.intra_smoothing_disabled = s->sps->spsRext.intra_smoothing_disabled_flag;
    g.constrained_intra_pred   = s->pps->constrained_intra_pred_flag;
    if (ohevc_tables_intra_pred_cip(&g, s->sps->log2_min_pu_size,
                                    (const uint8_t *)&s->ref->tab_mvf[0].pred_flag, sizeof(MvField), PF_INTRA,
                                    x0, y0, log2_size, c_idx,
                                    c_idx ? lc->tu.intra_pred_mode_c : lc->tu.intra_pred_mode,
                                    lc->na.cand_bottom_left, lc->na.cand_left, lc->na.cand_up_left,
                                    lc->na.cand_up, lc->na.cand_up_right) != OHEVC_OK)
        note_error(backend_of(s->avctx));
}
#define STUB(n) static void intra_pred_##n(HEVCContext *s, int x0, int y0, int c) { intra_pred_hip(s, x0, y0, n, c); }
STUB(2) STUB(3) STUB(4) STUB(5)

void ohhip_hevc_pred_init(HEVCPredContext *hpc, int bit_depth)
{
    ff_hevc_pred_init(hpc, bit_depth);                              /* hevcpred.c:47 */
    hpc->intra_pred[0] = intra_pred_2;
    hpc->intra_pred[1] = intra_pred_3;
    hpc->intra_pred[2] = intra_pred_4;
    hpc->intra_pred[3] = intra_pred_5;
}

/* the picture-store slot of a host frame (allocated on first sight or when the geometry of the buffer changed).  Returns the bufs
 * index with be->lock HELD, or -1 (unlocked). */
static int slot_of_frame_locked(ohhip_backend *be, ohevc_ctx *ctx, const HEVCContext *s, const AVFrame *f, int *fresh)
{
    int i, k, slot, cfmt = s->sps->chroma_array_type ? s->sps->chroma_array_type : 1;
    pthread_mutex_lock(&be->lock);
    for (i = 0; i < be->nbufs; i++)
        if (be->bufs[i].data0 == f->data[0])
            break;
    if (i < be->nbufs && (be->bufs[i].w != s->sps->width || be->bufs[i].h != s->sps->height ||
                          be->bufs[i].bd != s->sps->bit_depth || be->bufs[i].fmt != cfmt)) {
        ohevc_tables_unregister_picture(ctx, be->bufs[i].slot);
        ohevc_pic_release(ctx, be->bufs[i].slot);
        /* The decoder dropped its buffer pool with the old geometry and this address came back from the allocator: the page locks taken for
         * the OLD picture of this buffer go - only those.  (Round 3 dropped every page lock of the store here, under frame threads while
         * other threads' copy-backs into their still-living buffers were in flight.)  Nobody copies into this buffer now: the decoder
         * recycles a buffer only after the application has let go of the picture. */
        for (k = 0; k < 3; k++)
            if (be->bufs[i].pin_ptr[k])
                ohevc_host_unpin(ctx, be->bufs[i].pin_ptr[k], be->bufs[i].pin_bytes[k]);
        be->bufs[i] = be->bufs[--be->nbufs];
        i = be->nbufs;
    }
    *fresh = i == be->nbufs;
    if (i == be->nbufs) {
        slot = be->nbufs < MAX_BUFS ? ohevc_pic_alloc(ctx, s->sps->width, s->sps->height, cfmt, s->sps->bit_depth) : -1;
        if (slot < 0) {
            const int full = be->nbufs >= MAX_BUFS;
            pthread_mutex_unlock(&be->lock);
            if (full)
                fprintf(stderr, "ohhip: the decoder holds %d frame buffers: more than a picture store takes (OHEVC_MAX_PICTURES)\n", MAX_BUFS);
            else
                fprintf(stderr, "ohhip: pic_alloc failed: %s\n", ohevc_last_error());
            note_error(be);
            return -1;
        }
        memset(&be->bufs[i], 0, sizeof(be->bufs[i]));
        be->bufs[i].data0 = f->data[0];
        be->bufs[i].slot = slot;
        be->bufs[i].w = s->sps->width;
        be->bufs[i].h = s->sps->height;
        be->bufs[i].bd = s->sps->bit_depth;
        be->bufs[i].fmt = cfmt;
        be->bufs[i].poc = INT_MIN;
        be->bufs[i].seq = -1;
        be->bufs[i].index = -1;
        be->bufs[i].remote = 0;
        be->bufs[i].have_motion = be->bufs[i].have_planes = 1;
        be->nbufs++;
    }
    return i;
}

static int find_buf_locked(ohhip_backend *be, const uint8_t *data0)
{
    int i;
    for (i = 0; i < be->nbufs; i++)
        if (be->bufs[i].data0 == data0)
            return i;
    return -1;
}

/* pin_frames on buffers of the decoder's OWN frame pool (own_frames 0, the default allocator): the pool frees them in mid-stream when it is
 * re-created (utils.c:555-560), and a page lock that outlives its memory is worse than none - the runtime keeps believing the range is
 * page-locked, and whatever the allocator puts there next (another frame buffer at the same address: the round-6 device fault; or, after the
 * decoder is long gone, any array a later copy-back lands in) is written through a mapping that no longer exists.  The pool keeps, per
 * buffer, the callback that really frees it (BufferPoolEntry.free, buffer_internal.h:60-75; called by buffer_pool_free, buffer.c:227-238):
 * the hooks put their own in front of it - drop the page lock, forget the address, then let the original free the memory. */
static int ohhip_get_buffer2(AVCodecContext *avctx, AVFrame *frame, int flags);
typedef struct pin_guard { void (*free)(void *opaque, uint8_t *data); void *opaque; ohhip_backend *be; unsigned be_id; size_t size; } pin_guard;
static void guarded_free(void *opaque, uint8_t *data)
{
    pin_guard *g = opaque;
    ohhip_backend *b, *alive = NULL;
    int i, k;
    pthread_mutex_lock(&g_reg_lock);
    for (b = g_backends; b; b = b->next)
        if (b == g->be && b->id == g->be_id)
            alive = b;
    if (alive && alive->root) {                 /* (a back end that is gone dropped all of its page locks in ohhip_backend_pre_close) */
        ohevc_host_unpin(alive->root, data, g->size);
        pthread_mutex_lock(&alive->lock);
        for (i = 0; i < alive->nbufs; i++)
            for (k = 0; k < 3; k++)
                if (alive->bufs[i].pin_ptr[k] == (void *)data)
                    alive->bufs[i].pin_ptr[k] = NULL;
        pthread_mutex_unlock(&alive->lock);
    }
    pthread_mutex_unlock(&g_reg_lock);
    g->free(g->opaque, data);
    free(g);
}
static void guard_pool_entry(ohhip_backend *be, const AVBufferRef *ref)
{
    BufferPoolEntry *e = ref && ref->buffer ? ref->buffer->opaque : NULL;
    pin_guard *g;
    if (!e || e->data != ref->data || !e->pool || !e->free || e->free == guarded_free)      /* (not a pool buffer, or guarded already) */
        return;
    if (!(g = malloc(sizeof(*g))))
        return;
    g->free = e->free; g->opaque = e->opaque; g->be = be; g->be_id = be->id; g->size = (size_t)ref->size;
    e->opaque = g;
    e->free = guarded_free;
}

/* INTEGRATION.md section 3, rows alloc_frame + hevc_frame_start */
static int device_bs_frame(const HEVCContext *s);
/* boundary-strength calls the picture's own thread records (ohhip_deblocking_boundary_strengths below) */
static __thread ohevc_bs_call *t_bs_buf;
static __thread int            t_bs_n, t_bs_cap;
static __thread const HEVCContext *t_bs_direct;       /* the context whose calls this thread records directly, NULL: none */
int ohhip_set_new_ref(HEVCContext *s, AVFrame **frame, int poc)
{
    int ret = ff_hevc_set_new_ref(s, frame, poc);                   /* hevc_refs.c */
    ohhip_backend *be = backend_of(s->avctx);
    const AVFrame *f;
    ohevc_ctx *ctx;
    int i, k, slot, fresh;
    if (be && be->fm_on && t_publish && t_be == be)    /* the previous picture of this thread never reached its frame end (a decoding error) */
        publish_failed(be);
    if (ret < 0)
        return ret;
    if (!be || !(ctx = thread_ctx(be)))
        return AVERROR(ENOMEM);
    /* an application thread may drive several decoders in turn: this thread's table calls belong to THIS instance's context from here on */
    if (t_ctx != ctx && ohevc_tables_bind(ctx) != OHEVC_OK)
        return AVERROR(EINVAL);
    t_be = be;
    t_ctx = ctx;
    t_error = 0;
    t_trace_start = be->trace ? now_s() : 0;
    t_trace_poc = poc;
    f = s->ref->frame;
    if ((i = slot_of_frame_locked(be, ctx, s, f, &fresh)) < 0)
        return AVERROR(ENOMEM);
    be->bufs[i].copy_queued = 0;        /* (a picture nobody took out: its queued copy is ordered in front of the new picture's work by the store) */
    /* INTEGRATION.md section 3, row alloc_frame: page-lock the buffers the decoder's pool recycles (hevc_refs.c:75-114, get_buffer.c), so
     * that the copy-back of every picture is a DMA.  One hipHostRegister per pool buffer, ever: known ranges return at once.  (After the
     * slot look-up: a buffer that came back with another geometry had its old page locks dropped there.) */
    if (be->opt.pin_frames && ohevc_ctx_has_device(ctx) && !pool_owns(be, f->buf[0])) {     /* (own_frames: born page-locked) */
        for (k = 0; k < 3 && k < AV_NUM_DATA_POINTERS && f->buf[k]; k++) {
            if (be->bufs[i].pin_ptr[k] == f->buf[k]->data && be->bufs[i].pin_bytes[k] == (size_t)f->buf[k]->size)
                continue;
            if (ohevc_host_pin(ctx, f->buf[k]->data, f->buf[k]->size) != OHEVC_OK) {
                if (be->opt.pin_frames == 1)
                    fprintf(stderr, "ohhip: frame buffers stay pageable: %s\n", ohevc_last_error());
                be->opt.pin_frames = 2;                  /* say it once */
                break;
            }
            be->bufs[i].pin_ptr[k] = f->buf[k]->data;
            be->bufs[i].pin_bytes[k] = (size_t)f->buf[k]->size;
            /* (buffers of the decoder's own pool: the default allocator - a decoder nobody attached a back end to, own_frames 0 - or
             * this back end's allocator falling back to it: a fifth geometry, no page-locked memory left) */
            if (s->avctx->get_buffer2 == avcodec_default_get_buffer2 || s->avctx->get_buffer2 == ohhip_get_buffer2)
                guard_pool_entry(be, f->buf[k]);        /* the page lock goes where the decoder's pool frees the buffer */
        }
    }
    /* (Taking these locks later - at the picture's frame end, out of this serial prologue - was tried at the end of round 4 and gave a fresh
     * decoder's first pass 4.5 ms back (profiles/r14_*), but a page lock wants the store's table exclusively, i.e. waits for every copy-back in
     * flight, and at a frame end other threads are already waiting for THIS picture: when the decoder re-created its buffer pool under four frame
     * threads, one run in six stood still until the reference wait timed out.  Here, before anything can depend on the picture, the wait is harmless.) */
    slot = be->bufs[i].slot;
    if (be->fm_on && be->bufs[i].remote && be->bufs[i].index >= 0 && be->fm.release) {
        /* the buffer last held a remote picture: whatever is still in flight for it (planes nobody predicted from, a motion field
         * nobody asked for) is waited for and freed before the slot's memory gets a new picture */
        const int old = be->bufs[i].index;
        be->bufs[i].index = -1;
        pthread_mutex_unlock(&be->lock);
        if (be->fm.release(be->fm.user, old) != 0)
            note_error(be);
        pthread_mutex_lock(&be->lock);
        i = find_buf_locked(be, f->data[0]);             /* the table may have been compacted meanwhile */
        if (i < 0) {
            pthread_mutex_unlock(&be->lock);
            return AVERROR(EINVAL);
        }
    }
    be->bufs[i].ctx = ctx;
    be->bufs[i].poc = s->ref->poc;
    be->bufs[i].seq = s->ref->sequence;
    be->bufs[i].index = -1;
    be->bufs[i].remote = 0;
    be->bufs[i].have_motion = be->bufs[i].have_planes = 1;
    t_remote = 0;
    if (be->fm_on) {
        /* a picture nothing can predict from: sub-layer non-reference (even nal_unit_type below 16, H.265 table 7-1) in the
         * highest temporal sub-layer -- it is not exchanged */
        const int exchanged_pic = !(s->nal_unit_type < 16 && !(s->nal_unit_type & 1) && s->temporal_id == s->sps->max_sub_layers - 1);
        const size_t mvf_bytes = (size_t)s->sps->min_pu_width * s->sps->min_pu_height * sizeof(MvField);     /* hevc.c:178 */
        int exchanged = exchanged_pic, owner;
        be->bufs[i].index = be->fm_index++;
        /* Who reconstructs the picture.  Per picture: decoding-order index mod world - every exchanged picture crosses the wire.  Per IDR segment
         * (ohhip_frames_mode.segment_ownership): an IDR picture empties the decoded picture buffer and nothing after it predicts from
         * anything before it (H.265 8.3.1, 8.3.2), so a segment - an IDR picture and everything up to the next one - is decoded by ONE
         * rank, start to end, and nothing of it is needed anywhere else: no exchange at all, the ranks work on different segments of the
         * stream at the same time (the random-access points every broadcast / streaming encoder puts in once a second or two). */
        /* BLA pictures open a segment too: like an IDR picture they empty the decoded picture buffer (hevc.c:561 clears the references for both), what
         * their leading pictures name from before is generated, not read.  A CRA picture in mid-stream does NOT: its RASL pictures predict from
         * pictures before it (H.265 8.3.3), which another rank would own - an open-GOP stream without IDR / BLA pictures stays on one rank under
         * this rule (use per-picture ownership for it); the first picture of a stream opens segment 0 whatever it is. */
        if (IS_IDR(s) || IS_BLA(s) || be->fm_segment < 0)
            be->fm_segment++;
        owner = be->fm.segment_ownership ? be->fm_segment % be->fm.world : be->bufs[i].index % be->fm.world;
        if (be->fm.segment_ownership)
            exchanged = 0;
        be->bufs[i].remote = owner != be->fm.rank;
        /* a picture that is not exchanged has nothing to wait for - H.265 8.3.2 only bars it from the Curr sets, a stream may keep it in
         * a Foll set of later pictures */
        be->bufs[i].have_motion = be->bufs[i].have_planes = !be->bufs[i].remote || !exchanged;
        be->bufs[i].have_rows = 0;
        if (!exchanged)
            be->bufs[i].index = -1;                  /* (nothing to release either) */
        if (be->bufs[i].remote) {
            const int index = be->bufs[i].index;
            pthread_mutex_unlock(&be->lock);
            t_remote = 1;
            t_frame_open = 0;
            t_s = s;
            /* later pictures name this one by its host planes (the MC wrappers' src pointers): keep the pointer -> slot map */
            if (ohevc_tables_register_picture(ctx, slot, (uint8_t *const *)f->data, f->linesize) != OHEVC_OK ||
                (exchanged && be->fm.subscribe(be->fm.user, index, ctx, slot, mvf_bytes) != 0)) {
                fprintf(stderr, "ohhip: subscribing to remote picture %d failed: %s\n", index, ohevc_last_error());
                note_error(be);
                return AVERROR(EINVAL);
            }
            return 0;
        }
        t_publish = exchanged ? i + 1 : 0;
        t_publish_index = be->bufs[i].index;
        t_publish_mvf_bytes = mvf_bytes;
    }
    pthread_mutex_unlock(&be->lock);
    /* slice threads: the WPP-row / tile workers of this picture all record into ctx (ohhip_cabac_init binds them) */
    ohevc_tables_set_concurrent(ctx, (s->threads_type & FF_THREAD_SLICE) && s->threads_number > 1);
    if (ohevc_tables_register_picture(ctx, slot, (uint8_t *const *)f->data, f->linesize) != OHEVC_OK ||
        ohevc_tables_begin_frame(ctx, slot) != OHEVC_OK) {
        fprintf(stderr, "ohhip: begin_frame failed: %s\n", ohevc_last_error());
        note_error(be);
        return AVERROR(EINVAL);
    }
    t_frame_open = 1;
    t_s = s;
    t_bl_tf = s->BL_frame ? &s->BL_frame->tf : NULL;
    if (device_bs_frame(s) == 2 && ohevc_tables_keep_motion(ctx, s->sps->log2_min_pu_size) != OHEVC_OK)     /* boundary strengths from the MC jobs */
        note_error(be);
    t_bs_n = 0;
    t_bs_direct = device_bs_frame(s) && !((s->threads_type & FF_THREAD_SLICE) && s->threads_number > 1) ? s : NULL;
    return 0;
}

/* INTEGRATION.md section 3, row generate_missing_ref.  ff_hevc_frame_rps (hevc_refs.c:637, called at hevc.c:3250 right after
 * ff_hevc_set_new_ref) synthesises every reference picture the stream names but the DPB does not hold (a stream that starts at a
 * CRA, lost pictures): generate_missing_ref (hevc_refs.c:538-598) allocates a frame and fills its HOST planes with mid-grey.  No
 * table call and no frame_begin / frame_end ever touches such a picture, so its samples must be sent to the device here: every
 * reference of the new picture whose buffer does not hold a picture this back-end reconstructed (other poc / sequence tag, or
 * never seen) is registered and uploaded. */
int ohhip_frame_rps(HEVCContext *s)
{
    int ret = ff_hevc_frame_rps(s);
    ohhip_backend *be = backend_of(s->avctx);
    ohevc_ctx *ctx;
    int t, k, c;
    if (ret < 0 || !s->ref)
        return ret;
    if (!be || !(ctx = thread_ctx(be)))
        return AVERROR(ENOMEM);
    for (t = 0; t < NB_RPS_TYPE; t++)
        for (k = 0; k < s->rps[t].nb_refs; k++) {
            const HEVCFrame *ref = s->rps[t].ref[k];
            int i, slot, fresh, known;
            if (!ref || ref == s->ref || !ref->frame || !ref->frame->data[0])
                continue;
            if ((i = slot_of_frame_locked(be, ctx, s, ref->frame, &fresh)) < 0)
                return AVERROR(ENOMEM);
            known = !fresh && be->bufs[i].poc == ref->poc && be->bufs[i].seq == ref->sequence;
            slot = be->bufs[i].slot;
            if (known && be->fm_on && !t_remote && be->bufs[i].remote && !be->bufs[i].have_motion && (t == ST_CURR_BEF || t == ST_CURR_AFT || t == LT_CURR)) {
                /* (only pictures of the Curr sets can be the collocated picture or a prediction reference, H.265 8.3.2) */
                /* the wait of the reference's frame threads for a collocated picture's motion field (hevc_mvs.c) */
                const int index = be->bufs[i].index;
                be->bufs[i].have_motion = 1;
                pthread_mutex_unlock(&be->lock);
                if (be->fm.await_motion(be->fm.user, index, ref->tab_mvf, (size_t)s->sps->min_pu_width * s->sps->min_pu_height * sizeof(MvField)) != 0) {
                    fprintf(stderr, "ohhip: the motion field of remote picture %d did not arrive\n", index);
                    note_error(be);
                    return AVERROR(EINVAL);
                }
                continue;
            }
            if (!known) {
                be->bufs[i].poc = ref->poc;
                be->bufs[i].seq = ref->sequence;
                be->bufs[i].ctx = ctx;
                /* a generated reference lives here now, not the (possibly remote, possibly un-awaited) picture the buffer held before */
                be->bufs[i].remote = 0;
                be->bufs[i].index = -1;
                be->bufs[i].have_motion = be->bufs[i].have_planes = 1;
            }
            pthread_mutex_unlock(&be->lock);
            if (known)
                continue;
            if (ohevc_tables_register_picture(ctx, slot, (uint8_t *const *)ref->frame->data, ref->frame->linesize) != OHEVC_OK) {
                note_error(be);
                return AVERROR(EINVAL);
            }
            /* SHVC: the inter-layer reference picture (ff_hevc_set_new_iter_layer_ref, hevc_refs.c:149-180) has no samples yet - the first
             * prediction unit that names it makes the up-sampling slots fill it (hevc.c:2077-2099), on the device, whole */
            if (ref == s->inter_layer_ref)
                continue;
            for (c = 0; c < 3; c++)
                if (ref->frame->data[c] && ohevc_pic_upload(ctx, slot, c, ref->frame->data[c], ref->frame->linesize[c]) != OHEVC_OK) {
                    fprintf(stderr, "ohhip: upload of a generated reference picture failed: %s\n", ohevc_last_error());
                    note_error(be);
                    return AVERROR(EINVAL);
                }
        }
    return ret;
}

/* Slice threads.  The worker entry functions (hls_decode_entry_wpp / hls_decode_entry_tiles, hevc.c:2744-2920) run on pool
 * threads with a per-thread COPY of the decoder context (s1->sList[self_id]); INTEGRATION.md section 3 puts one
 * `ohevc_tables_bind(s->hip)` at their top.  Here the same effect comes from renaming the first call every CTB makes from
 * hevc.c with the context in hand, ff_hevc_cabac_init (hevc.c:2666,2785,2873): the wrapper binds the calling thread to the
 * context that is reconstructing s->ref, once per picture and thread. */
void ohhip_cabac_init(HEVCContext *s, int ctb_addr_ts)
{
    /* SHVC: the base-layer frame whose row progress THIS thread's prediction units wait for (ohhip_await_progress).  Set here, on every thread
     * that parses CTBs - with frame x slice threads the prediction units run on WPP pool workers that never opened a frame (t_s is NULL there) */
    t_bl_tf = s->BL_frame ? &s->BL_frame->tf : NULL;
    if ((s->threads_type & FF_THREAD_SLICE) && s->threads_number > 1 && s->ref && s->ref->frame) {
        /* looked up every time (once per CTB): a host buffer address names a different picture -- and, with frame + slice
         * threads, a different context -- every time the decoder's pool recycles it */
        const uint8_t *d0 = s->ref->frame->data[0];
        static __thread const uint8_t *seen_d0;       /* the last answer of this thread: asked once per CTB, and with frame x slice threads */
        static __thread int seen_poc, seen_seq;       /* 64 threads would queue on the lock for it */
        static __thread ohevc_ctx *seen_ctx;
        static __thread unsigned seen_epoch;          /* contexts die with their instance: the registry's epoch changes at every open / close */
        ohevc_ctx *ctx = NULL;
        int i;
        if (seen_epoch == g_epoch && seen_d0 == d0 && seen_poc == s->ref->poc && seen_seq == s->ref->sequence && seen_ctx) {
            ctx = seen_ctx;
        } else {
            ohhip_backend *be = backend_of(s->avctx);
            const unsigned epoch = g_epoch;
            if (be) {
                pthread_mutex_lock(&be->lock);
                if ((i = find_buf_locked(be, d0)) >= 0)
                    ctx = be->bufs[i].ctx;
                pthread_mutex_unlock(&be->lock);
            }
            seen_d0 = d0; seen_poc = s->ref->poc; seen_seq = s->ref->sequence; seen_ctx = ctx; seen_epoch = epoch;
        }
        if (ctx && ctx != t_ctx) {            /* a pool thread: it never owns a context, it borrows the picture's */
            if (ohevc_tables_bind(ctx) != OHEVC_OK)
                note_error(backend_of(s->avctx));
            t_ctx = ctx;
        }
    }
    ff_hevc_cabac_init(s, ctb_addr_ts);                             /* hevc_cabac.c */
}

/* Cross-component prediction: hls_cross_component_pred (hevc.c:1186-1200, static) = the two calls below; INTEGRATION.md adds
 * `ohevc_tables_cross_component(lc->tu.res_scale_val)` at its end, here the renamed call sites rebuild the value. */
static __thread int t_log2_res_scale_abs_plus1;
int ohhip_log2_res_scale_abs(HEVCContext *s, int idx)
{
    int v = ff_hevc_log2_res_scale_abs(s, idx);                     /* hevc_cabac.c */
    t_log2_res_scale_abs_plus1 = v;
    if (v == 0)
        ohevc_tables_cross_component(0);
    return v;
}
int ohhip_res_scale_sign_flag(HEVCContext *s, int idx)
{
    int f = ff_hevc_res_scale_sign_flag(s, idx);
    ohevc_tables_cross_component((1 << (t_log2_res_scale_abs_plus1 - 1)) * (1 - 2 * f));      /* hevc.c:1192-1193 */
    return f;
}

/* OHHIP_BACKTRACE=1: native backtrace of a crashing thread on stderr (resolve the offsets with addr2line on the same .so) */
static void crash_handler(int sig)
{
    void *frames[64];
    int n = backtrace(frames, 64);
    static const char msg[] = "ohhip: fatal signal, native backtrace:\n";
    if (write(2, msg, sizeof(msg) - 1) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

/* SURVEY.md 8f-3, host half: the in-loop filter drivers are skipped -- ff_hevc_hls_filters /
 * ff_hevc_hls_filter (hevc_filter.c:1027-1064; call sites hevc.c:2690-2695,2809-2818,2892-2901,3002-3012 renamed to these) keep
 * only their progress reports -- and the frame-end hook hands the maps they would have read to ohevc_tables_derive_filters.
 * 16x16-CTB streams with SAO (output depends on the ORDER of the driver calls: filter lag) have that order replayed by the bulk form. */
static int bulk_filters(const HEVCContext *s)
{
    const ohhip_backend *be = backend_of(s->avctx);
    /* 16x16 CTBs with SAO: the reference's output depends on the order of its driver calls (filter lag); the bulk form replays that
     * order for one decoding thread per picture, slice threads keep the drivers (their order is whatever the row threads make it) */
    return be && be->opt.bulk_filters &&
           !(s->sps->log2_ctb_size == 4 && s->sps->sao_enabled && (s->threads_type & FF_THREAD_SLICE) && s->threads_number > 1);
}

void ohhip_hls_filter(HEVCContext *s, int x, int y, int ctb_size)
{
    if (!bulk_filters(s)) {
        ff_hevc_hls_filter(s, x, y, ctb_size);
        return;
    }
    /* what other frame threads wait for (hevc_filter.c:1038-1050): rows whose CTBs have all been through here */
    if (s->threads_type & FF_THREAD_FRAME) {
        const int x_end = x >= s->sps->width - ctb_size, y_end = y >= s->sps->height - ctb_size;
        if (s->sps->sao_enabled) {
            if (y && x_end)
                ff_thread_report_progress(&s->ref->tf, y - ctb_size, 0);
            if (x_end && y_end)
                ff_thread_report_progress(&s->ref->tf, y, 0);
        } else if (y && x_end) {
            ff_thread_report_progress(&s->ref->tf, y, 0);
        }
    }
}

/* End of a CTU row of a picture this thread parses alone: what has been recorded of an intra picture goes to the device now
 * (ohevc_frame_flush_intra: a no-op once the picture has inter prediction, and until flush_intra_kib KiB of records and coefficients are
 * waiting - a flush cuts the picture's dependency chain into bands that run one after the other, so it only pays for pictures whose
 * parsing takes much longer than their chain: dense residuals, 4K / 8K; ohevc_ctx.h has the measurement). */
static void row_end(HEVCContext *s, int x_ctb, int ctb_size)
{
    int kib;
    if (!(t_frame_open && t_ctx && t_be && s == t_s && t_be->opt.flush_intra_kib != 0 && x_ctb >= s->sps->width - ctb_size) ||
        ((s->threads_type & FF_THREAD_SLICE) && s->threads_number > 1))
        return;
    kib = t_be->opt.flush_intra_kib;
    if (kib < 0) {
        /* by the picture's size: two or three hand-overs per encoder-like intra picture (its records and coefficients are ~0.8 bytes per luma
           sample), so that the first half's dependency chain runs while the second half is parsed; more of them and the bands' chains add up
           (a picture's levels run along diagonals through all of its CTU rows: DESIGN.md 5h) */
        const long long px = (long long)s->sps->width * s->sps->height;
        kib = (int)(px * 45 / 100 / 1024);
        kib = kib < 512 ? 512 : kib > 4096 ? 4096 : kib;
    }
    if (ohevc_frame_flush_intra(t_ctx, kib) != OHEVC_OK)
        note_error(t_be);
}

void ohhip_hls_filters(HEVCContext *s, int x_ctb, int y_ctb, int ctb_size)
{
    row_end(s, x_ctb, ctb_size);
    if (!bulk_filters(s)) {
        ff_hevc_hls_filters(s, x_ctb, y_ctb, ctb_size);
        return;
    }
    {   /* the reference's own dispatch, hevc_filter.c:1053-1063 */
        const int x_end = x_ctb >= s->sps->width - ctb_size, y_end = y_ctb >= s->sps->height - ctb_size;
        if (y_ctb && x_ctb)
            ohhip_hls_filter(s, x_ctb - ctb_size, y_ctb - ctb_size, ctb_size);
        if (y_ctb && x_end)
            ohhip_hls_filter(s, x_ctb, y_ctb - ctb_size, ctb_size);
        if (x_ctb && y_end)
            ohhip_hls_filter(s, x_ctb - ctb_size, y_ctb, ctb_size);
    }
}

/* SURVEY.md 8f-3, second half: the boundary strengths (ff_hevc_deblocking_boundary_strengths, hevc_filter.c:805-941: 15-18 % of the front end's
 * time on 1080p inter content) are derived on the device from the motion field.  The call sites (hevc.c:1578,1607,2400,2484) land here: the call
 * is recorded - position, size, the slice / tile flags of its CTB - and the frame-end hook hands s->ref->tab_mvf and s->cbf_luma over instead of
 * s->horizontal_bs / vertical_bs.  Where the reference's filter drivers still run on the host (they read those arrays) the reference's function
 * is called as before. */
static int device_bs(const HEVCContext *s)
{
    /* (asked by every thread that parses a part of the picture - slice threads have the picture's context bound by ohhip_cabac_init - and the
     * answer depends on the picture alone) */
    return t_ctx && !(s->pps->tiles_enabled_flag && s->threads_number != 1) &&       /* tiles_filters (hevc.c:2967) rewrites entries afterwards */
           bulk_filters(s) && ohevc_tables_bs_wanted(t_ctx, s->sps->log2_ctb_size, s->sps->sao_enabled, s->sps->chroma_array_type, 1);
}

static int device_bs_frame(const HEVCContext *s)       /* the same decision at the frame end (the frame is no longer "open" there) */
{
    if (!t_ctx || (s->pps->tiles_enabled_flag && s->threads_number != 1) || !bulk_filters(s))
        return 0;
    return ohevc_tables_bs_wanted(t_ctx, s->sps->log2_ctb_size, s->sps->sao_enabled, s->sps->chroma_array_type, 1);     /* 1: from tab_mvf, 2: from the MC jobs */
}

/* One call costs the reference ~40 ns on 1080p inter content (10 000 calls, 0.4 ms per picture): recording it must cost far less than that.
 * The thread that owns the picture (no slice threads: it makes every call) appends 8 bytes to a buffer of its own - the decision was taken at
 * the picture's start (t_bs_direct) - and the frame-end hook hands the buffer over in one piece; slice threads go through the picture's
 * context (ohevc_tables_bs_call, a recorder per thread). */

void ohhip_deblocking_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size)
{
    const HEVCLocalContext *lc = s->HEVClc;
    if (s == t_bs_direct) {
        ohevc_bs_call *b;
        if (t_bs_n == t_bs_cap) {
            int cap = t_bs_cap ? 2 * t_bs_cap : 16384;
            ohevc_bs_call *nb = realloc(t_bs_buf, (size_t)cap * sizeof(*nb));
            if (!nb) { note_error(t_be); return; }
            t_bs_buf = nb; t_bs_cap = cap;
        }
        b = &t_bs_buf[t_bs_n++];
        b->x0 = (uint16_t)x0; b->y0 = (uint16_t)y0; b->log2_size = (uint8_t)log2_trafo_size; b->reserved = 0;
        b->flags = (uint8_t)((lc->slice_or_tiles_up_boundary & 3) | ((lc->slice_or_tiles_left_boundary & 3) << 2) |
                             (s->sh.slice_loop_filter_across_slices_enabled_flag ? OHEVC_BS_ACROSS_SLICES : 0));
        return;
    }
    if (!device_bs(s)) {
        ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_trafo_size);
        return;
    }
    if (ohevc_tables_bs_call(x0, y0, log2_trafo_size, (lc->slice_or_tiles_up_boundary & 3) | ((lc->slice_or_tiles_left_boundary & 3) << 2) |
                                                      (s->sh.slice_loop_filter_across_slices_enabled_flag ? OHEVC_BS_ACROSS_SLICES : 0)) != OHEVC_OK)
        note_error(backend_of(s->avctx));
}

/* Boundary strengths from the picture's own MC jobs (mode 2) rest on "every inter prediction unit made its MC calls".  The reference writes
 * tab_mvf BEFORE hls_prediction_unit gives up on a unit whose reference picture is missing (hevc.c:2068-2075,2089-2091: `if (!ref0) return;`):
 * such a unit keeps its inter pred_flag in tab_mvf but never reaches a table slot, the device-side grid would show it as intra (bS 2) and
 * the deblocking would differ from the reference's on damaged streams.  A picture whose slices name a reference that is not there therefore
 * hands its motion field over (mode 1) - decided here, at the frame end, when all of its slice headers have been seen. */
static int any_missing_reference(const HEVCContext *s)
{
    int i, l, k;
    for (i = 0; i <= s->slice_idx && i < MAX_SLICES_IN_FRAME; i++) {
        const RefPicList *rpl = s->ref->refPicList[i];
        if (!rpl)
            continue;
        for (l = 0; l < 2; l++)
            for (k = 0; k < rpl[l].nb_refs && k < MAX_REFS; k++)
                if (!rpl[l].ref[k])
                    return 1;
    }
    return 0;
}

static int derive_filters(HEVCContext *s)
{
    ohevc_filter_maps m;
    memset(&m, 0, sizeof(m));
    m.width = s->sps->width; m.height = s->sps->height;
    m.log2_ctb_size = s->sps->log2_ctb_size; m.log2_min_cb_size = s->sps->log2_min_cb_size; m.log2_min_pu_size = s->sps->log2_min_pu_size;
    m.chroma_format_idc = s->sps->chroma_array_type;
    m.cb_qp_offset = s->pps->cb_qp_offset; m.cr_qp_offset = s->pps->cr_qp_offset;
    m.sao_enabled = s->sps->sao_enabled;
    m.tiles_enabled = s->pps->tiles_enabled_flag; m.loop_filter_across_tiles = s->pps->loop_filter_across_tiles_enabled_flag;
    m.pcm_or_bypass = (s->sps->pcm_enabled_flag && s->sps->pcm.loop_filter_disable_flag) || s->pps->transquant_bypass_enable_flag;
    m.horizontal_bs = s->horizontal_bs; m.vertical_bs = s->vertical_bs; m.bs_width = s->bs_width;
    m.qp_y_tab = s->qp_y_tab; m.min_cb_width = s->sps->min_cb_width;
    m.deblock = (const int8_t *)s->deblock; m.deblock_stride = sizeof(DBParams);
    m.sao = (const ohevc_SAOParams *)s->sao;
    m.filter_slice_edges = s->filter_slice_edges; m.tab_slice_address = s->tab_slice_address;
    m.ctb_addr_rs_to_ts = s->pps->ctb_addr_rs_to_ts; m.tile_id = s->pps->tile_id;
    m.is_pcm = s->is_pcm; m.min_pu_width = s->sps->min_pu_width; m.min_pu_height = s->sps->min_pu_height;
    m.emulate_filter_lag = 1; m.ctb_addr_ts_to_rs = s->pps->ctb_addr_ts_to_rs;
    if (device_bs_frame(s)) {
        if (s == t_bs_direct && t_bs_n > 0 && ohevc_tables_bs_calls(t_bs_buf, t_bs_n) != OHEVC_OK)
            return OHEVC_ERR_STATE;
        t_bs_n = 0;
        m.tab_mvf = device_bs_frame(s) == 2 && !any_missing_reference(s) ? NULL : s->ref->tab_mvf; m.mvf_stride = sizeof(MvField);
        m.mvf_off_mv = offsetof(MvField, mv); m.mvf_off_poc = offsetof(MvField, poc); m.mvf_off_pred_flag = offsetof(MvField, pred_flag);
        m.mvf_pred_flag_bytes = sizeof(((MvField *)0)->pred_flag);
        m.cbf_luma = s->cbf_luma; m.min_tb_width = s->sps->min_tb_width; m.min_tb_height = s->sps->min_tb_height; m.log2_min_tb_size = s->sps->log2_min_tb_size;
    }
    return ohevc_tables_derive_filters(t_ctx, &m);
}

/* ---- instances ---- */
/* The environment supplies DEFAULTS for ohhip_options and nothing else: this is the only place of the back end that looks at it.  An
 * application that fills ohhip_options itself (hip_backend.h) never depends on the environment of its process. */
static const char *env_str(const char *name) { const char *v = getenv(name); return v && v[0] ? v : NULL; }
static int env_int(const char *name, int dflt) { const char *v = env_str(name); return v ? atoi(v) : dflt; }

size_t ohhip_options_size(void) { return sizeof(ohhip_options); }     /* for hosts that mirror the struct instead of including hip_backend.h */

void ohhip_options_default(ohhip_options *o)
{
    memset(o, 0, sizeof(*o));
    o->struct_size = sizeof(*o);
    o->device = env_int("OHHIP_DEVICE", 0);
    o->bulk_filters = env_int("OHHIP_BULK_FILTERS", 1) != 0;
    /* 1 since round 5: ending a frame only ISSUES its device work, the copy-back waits until the application takes the picture
     * (ohhip_backend_fetch_output, part of the recipe in INTEGRATION.md 2) - the decoding thread goes on parsing instead of waiting for the
     * device: +18-24 % on an all-intra stream at 16 frame threads, +3-12 % on an encoder-like one (profiles/r5z_intra16_switches.txt).
     * An application that cannot make that call sets 0. */
    o->defer_download = env_int("OHHIP_DEFER_DOWNLOAD", 1) != 0;
    /* 0 since round 6: frame buffers the back end did not make (own_frames 0, a decoder nobody attached a back end to, an application's own
     * allocator) stay pageable unless asked for.  A page lock on memory that belongs to somebody else is only as good as the promise that it
     * is dropped before the memory goes; the hooks keep that promise for the decoder's own pool (guard_pool_entry), but the runtime's
     * un-registration is lazy (hipHostUnregister returns in a microsecond) and one whole-suite run in five still died in a LATER, unrelated
     * copy into memory that had been page-locked and given back minutes before. */
    o->pin_frames = env_int("OHHIP_PIN_FRAMES", 0) != 0;
    o->async_issue = env_int("OHHIP_ASYNC_ISSUE", 0);
    o->record_only = env_str("OHHIP_RECORD_ONLY") ? (atoi(env_str("OHHIP_RECORD_ONLY")) == 2 ? 2 : 1) : 0;
    o->test_fail_index = env_int("OHHIP_TEST_FAIL_INDEX", -1);
    o->trace_path = env_str("OHHIP_TRACE_FRAMES");
    o->flush_intra_kib = env_int("OHHIP_FLUSH_INTRA_KIB", -1);
    o->level_launch = env_int("OHHIP_LEVEL_LAUNCH", -1);
    o->device_filters = env_int("OHHIP_DEVICE_FILTERS", -1);
    o->crash_backtrace = env_str("OHHIP_BACKTRACE") != NULL;
    o->park_frames = env_int("OHHIP_PARK_FRAMES", -1);
    o->own_frames = env_int("OHHIP_OWN_FRAMES", 1) != 0;
    o->queue_download = env_int("OHHIP_QUEUE_DOWNLOAD", 1) != 0;
}

/* this instance's choices on a context it has made (the library's process-wide debug setters stay what they are: defaults for tests) */
static void apply_ctx_options(const ohhip_backend *be, ohevc_ctx *ctx)
{
    ohevc_ctx_set_option(ctx, OHEVC_OPT_LEVEL_LAUNCH, be->opt.level_launch);
    ohevc_ctx_set_option(ctx, OHEVC_OPT_FILTERS_ON_DEVICE, be->opt.device_filters);
    ohevc_ctx_set_option(ctx, OHEVC_OPT_PARK_FRAMES, be->opt.park_frames);
}

ohhip_backend *ohhip_backend_new(const ohhip_options *o)
{
    ohhip_options def;
    ohhip_backend *be;
    /* the caller's struct may be OLDER (smaller) than this library's: its fields are laid over the defaults, the ones it does not have keep
     * them.  struct_size 0 = a zero-initialised / never-defaulted struct (every field would read as 0: host-side deblock derivation, no
     * deferred copy-back, "-1 = library default" lost), larger than ours = a host newer than the library: both refused, nothing is read. */
    ohhip_options_default(&def);
    if (o) {
        if (o->struct_size < offsetof(ohhip_options, device) + sizeof(int) || o->struct_size > sizeof(ohhip_options)) {
            fprintf(stderr, "ohhip: ohhip_backend_new: ohhip_options.struct_size is %zu (this library: %zu): call ohhip_options_default() first\n",
                    o->struct_size, sizeof(ohhip_options));
            return NULL;
        }
        memcpy(&def, o, o->struct_size);
        def.struct_size = sizeof(def);
    }
    o = &def;
    be = calloc(1, sizeof(*be));
    if (!be)
        return NULL;
    if (o->crash_backtrace) {
        signal(SIGSEGV, crash_handler);
        signal(SIGABRT, crash_handler);
    }
    be->magic = OHHIP_MAGIC;
    be->opt = *o;
    /* an asynchronous frame end queues the copy-back behind the picture's launches itself: a second, deferred one at fetch time would write
     * the same host planes from another thread at the same moment (ThreadSanitizer pass, tools/hipemu_tsan.sh) */
    if (be->opt.async_issue > 0)
        be->opt.defer_download = 0;
    pthread_mutex_init(&be->lock, NULL);
    /* host-side profiling / host-logic tests without a device (include/ohevc_debug.h): record, produce no pixels.  A test may
     * have switched record-only mode on itself (and installed a frame sink) before opening the decoder: leave that alone. */
    if (o->record_only)
        ohevc_debug_set_record_only(o->record_only);
    if (o->base_layer && (o->base_layer->magic != OHHIP_MAGIC || !o->base_layer->root)) {
        fprintf(stderr, "ohhip: base_layer does not name a live back end\n");
        pthread_mutex_destroy(&be->lock);
        free(be);
        return NULL;
    }
    if (o->base_layer)
        be->opt.device = o->base_layer->opt.device;
    /* SHVC: the enhancement layer's contexts use the base layer's picture store (and, through it, its table-level picture registry:
     * the up-sampling slots find the base-layer picture by its host address) */
    if ((o->base_layer ? ohevc_ctx_create_shared(&be->root, be->opt.device, o->base_layer->root) : ohevc_ctx_create(&be->root, o->device)) != OHEVC_OK) {
        fprintf(stderr, "ohhip: ctx_create failed: %s\n", ohevc_last_error());
        pthread_mutex_destroy(&be->lock);
        free(be);
        return NULL;
    }
    apply_ctx_options(be, be->root);
    be->pinned_blocks = ohevc_host_alloc_pins(be->root);
    if (be->opt.own_frames && (be->pool = calloc(1, sizeof(*be->pool)))) {
        pthread_mutex_init(&be->pool->m, NULL);
        be->pool->alive = be->pool->refs = 1;
    }
    if (o->trace_path)
        be->trace = calloc(MAX_TRACE, sizeof(*be->trace));
    pthread_mutex_lock(&g_reg_lock);
    be->id = g_next_id++;
    be->next = g_backends;
    g_backends = be;
    g_epoch++;
    pthread_mutex_unlock(&g_reg_lock);
    return be;
}

int ohhip_backend_options(const ohhip_backend *be, ohhip_options *out)
{
    size_t n;
    if (!be || be->magic != OHHIP_MAGIC || !out)
        return -1;
    n = out->struct_size;
    if (n < offsetof(ohhip_options, device) + sizeof(int) || n > sizeof(ohhip_options))
        return -1;
    memcpy(out, &be->opt, n);
    out->struct_size = n;
    return 0;
}

/* own_frames: AVCodecContext.get_buffer2 of a decoder with this back end (installed by ohhip_backend_attach; thread-safe: frame threads call
 * it directly, pthread_frame.c:903-908).  The LAYOUT of a frame - plane sizes, line sizes, where the picture starts behind its edge - is the
 * decoder's own: the first frame of every geometry is made by avcodec_default_get_buffer2 (utils.c:735) and measured; its buffers, and those
 * of every later frame, are then blocks of the back end's page-locked pool.  Nothing here restates how the reference lays a frame out. */
static int get_buffer2_impl(AVCodecContext *avctx, AVFrame *frame, int flags);
static double g_getbuf_s; static long g_getbuf_n;       /* OHHIP_TRACE_POOL: time inside get_buffer2, process-wide */
static int ohhip_get_buffer2(AVCodecContext *avctx, AVFrame *frame, int flags)
{
    static int trace = -1;
    double t0;
    int ret;
    if (trace < 0)
        trace = getenv("OHHIP_TRACE_POOL") != NULL;
    if (!trace)
        return get_buffer2_impl(avctx, frame, flags);
    t0 = now_s();
    ret = get_buffer2_impl(avctx, frame, flags);
    pthread_mutex_lock(&g_prof_lock);
    g_getbuf_s += now_s() - t0; g_getbuf_n++;
    if (g_getbuf_n % 33 == 0)
        fprintf(stderr, "pool: %ld frames from get_buffer2, %.3f ms each\n", g_getbuf_n, 1e3 * g_getbuf_s / g_getbuf_n);
    pthread_mutex_unlock(&g_prof_lock);
    return ret;
}
static int get_buffer2_impl(AVCodecContext *avctx, AVFrame *frame, int flags)
{
    ohhip_backend *be = NULL, *b;
    AVBufferRef *refs[4] = { NULL, NULL, NULL, NULL };
    int i, k, li = -1, ret;
    pthread_mutex_lock(&g_reg_lock);
    for (b = g_backends; b; b = b->next)
        if ((void *)b == avctx->opaque)
            be = b;
    pthread_mutex_unlock(&g_reg_lock);
    if (!be || !be->pool || !be->root || avctx->codec_type != AVMEDIA_TYPE_VIDEO)
        return avcodec_default_get_buffer2(avctx, frame, flags);
    pthread_mutex_lock(&be->lock);
    for (i = 0; i < be->nlayouts; i++)
        if (be->layouts[i].w == frame->width && be->layouts[i].h == frame->height && be->layouts[i].fmt == frame->format)
            li = i;
    pthread_mutex_unlock(&be->lock);
    if (li < 0) {
        int planes = 0, ok = 1;
        if ((ret = avcodec_default_get_buffer2(avctx, frame, flags)) < 0)
            return ret;
        while (planes < 4 && frame->buf[planes])
            planes++;
        /* one buffer per plane, the plane inside its buffer, nothing beyond buf[]: anything else keeps the decoder's frame as it is */
        ok = planes > 0 && planes <= 3 && !frame->buf[planes < 4 ? planes : 3] && frame->extended_data == frame->data && !frame->nb_extended_buf;
        for (k = 0; k < planes && ok; k++)
            ok = frame->data[k] >= frame->buf[k]->data && frame->data[k] < frame->buf[k]->data + frame->buf[k]->size && frame->linesize[k] > 0;
        for (k = planes; k < 4 && ok; k++)
            ok = !frame->data[k];
        pthread_mutex_lock(&be->lock);
        if (ok && be->nlayouts < 4) {
            li = be->nlayouts;
            be->layouts[li].w = frame->width; be->layouts[li].h = frame->height; be->layouts[li].fmt = frame->format; be->layouts[li].planes = planes;
            for (k = 0; k < 4; k++) {
                be->layouts[li].linesize[k] = frame->linesize[k];
                be->layouts[li].size[k] = k < planes ? frame->buf[k]->size : 0;
                be->layouts[li].off[k] = k < planes ? frame->data[k] - frame->buf[k]->data : 0;
            }
            be->nlayouts++;
            if (li == 0) {
                int total = 0;
                for (k = 0; k < planes; k++)
                    total += (be->layouts[0].size[k] + 63) & ~63;
                pool_prefetch_start(be, total);
            }
        }
        pthread_mutex_unlock(&be->lock);
        if (li < 0)
            return 0;                           /* a layout this allocator does not take over (or a fifth geometry): the decoder's own frame */
    }
    {
        ohhip_block *blk;
        int total = 0, at = 0, failed = 0;
        for (k = 0; k < be->layouts[li].planes; k++)
            total += (be->layouts[li].size[k] + 63) & ~63;
        if (!(blk = pool_get(be, total)))       /* no page-locked memory: the decoder's own frame (kept if it was just made to measure the layout) */
            return frame->buf[0] ? 0 : avcodec_default_get_buffer2(avctx, frame, flags);
        for (k = 0; k < be->layouts[li].planes && !failed; k++) {
            __atomic_add_fetch(&blk->out, 1, __ATOMIC_ACQ_REL);
            if (!(refs[k] = av_buffer_create((uint8_t *)blk->ptr + at, be->layouts[li].size[k], pool_release, blk, 0))) {
                __atomic_sub_fetch(&blk->out, 1, __ATOMIC_ACQ_REL);
                failed = 1;
            }
            at += (be->layouts[li].size[k] + 63) & ~63;
        }
        if (failed) {
            for (k = 0; k < 4; k++)
                av_buffer_unref(&refs[k]);
            pool_release(blk, NULL);
            return frame->buf[0] ? 0 : avcodec_default_get_buffer2(avctx, frame, flags);
        }
        pool_release(blk, NULL);                /* (this function's own hold) */
    }
    for (k = 0; k < 4; k++) {
        av_buffer_unref(&frame->buf[k]);        /* (the measured frame's buffers go back to the decoder's pool) */
        frame->buf[k] = refs[k];
        frame->data[k] = refs[k] ? refs[k]->data + be->layouts[li].off[k] : NULL;
        frame->linesize[k] = be->layouts[li].linesize[k];
    }
    frame->extended_data = frame->data;
    return 0;
}

int ohhip_backend_attach(ohhip_backend *be, AVCodecContext *avctx)
{
    if (!be || be->magic != OHHIP_MAGIC || !avctx)
        return -1;
    avctx->opaque = be;          /* inherited by every frame-thread copy (pthread_frame.c:276 and the `*copy = *src` of its init) */
    /* own_frames: unless the application brought an allocator of its own (the field is the default's after avcodec_alloc_context3) */
    be->nthreads = avctx->thread_count > 1 ? avctx->thread_count : 1;
    if (be->pool && avctx->get_buffer2 == avcodec_default_get_buffer2) {
        avctx->get_buffer2 = ohhip_get_buffer2;
        avctx->thread_safe_callbacks = 1;
    }
    /* one context per decoding thread the decoder is about to start (the "threads" option is set before avcodec_open2, like this call):
       made here, at open time, instead of in front of every thread's first picture */
    if (be->root && !be->opt.record_only) {
        int want = avctx->thread_count > 1 ? avctx->thread_count : 1, have;
        if (want > 64)
            want = 64;
        pthread_mutex_lock(&be->lock);
        have = be->nspare;
        pthread_mutex_unlock(&be->lock);
        for (; have < want; have++) {
            ohevc_ctx *ctx = new_thread_ctx(be);
            if (!ctx)
                break;                      /* (a thread that finds no spare context makes its own and reports the failure) */
            pthread_mutex_lock(&be->lock);
            be->spare[be->nspare++] = ctx;
            pthread_mutex_unlock(&be->lock);
        }
        ohevc_tables_bind(t_ctx);           /* (making a context binds the calling thread's table calls to it: back to what they were) */
    }
    return 0;
}

int ohhip_backend_device(const ohhip_backend *be) { return be ? be->opt.device : -1; }

int ohhip_backend_live_count(void)
{
    int n = 0;
    ohhip_backend *b;
    pthread_mutex_lock(&g_reg_lock);
    for (b = g_backends; b; b = b->next)
        n++;
    pthread_mutex_unlock(&g_reg_lock);
    return n;
}

static ohhip_backend *default_backend(void)
{
    ohhip_backend *be;
    pthread_mutex_lock(&g_reg_lock);
    be = g_default;
    pthread_mutex_unlock(&g_reg_lock);
    if (be)
        return be;
    be = ohhip_backend_new(NULL);
    pthread_mutex_lock(&g_reg_lock);
    if (!g_default) {
        g_default = be;
        be = NULL;
    }
    pthread_mutex_unlock(&g_reg_lock);
    if (be)                        /* another thread was faster */
        ohhip_backend_free(be);
    return g_default;
}

/* before the decoder frees its frame buffers (avcodec_close): their page locks go first */
void ohhip_backend_pre_close(ohhip_backend *be)
{
    int i, k;
    if (!be || !be->root)
        return;
    if (!be->opt.base_layer) {
        ohevc_host_unpin_all(be->root);
        return;
    }
    /* SHVC enhancement layer: the store is the base layer's too, whose decoder may still be decoding - only this decoder's buffers */
    pthread_mutex_lock(&be->lock);
    for (i = 0; i < be->nbufs; i++)
        for (k = 0; k < 3; k++)
            if (be->bufs[i].pin_ptr[k]) {
                ohevc_host_unpin(be->root, be->bufs[i].pin_ptr[k], be->bufs[i].pin_bytes[k]);
                be->bufs[i].pin_ptr[k] = NULL;
            }
    pthread_mutex_unlock(&be->lock);
}

/* after the decoder (and its threads) are gone */
void ohhip_backend_free(ohhip_backend *be)
{
    ohhip_backend **pp;
    int i, k;
    if (!be || be->magic != OHHIP_MAGIC)
        return;
    pthread_mutex_lock(&g_reg_lock);
    for (pp = &g_backends; *pp; pp = &(*pp)->next)
        if (*pp == be) {
            *pp = be->next;
            break;
        }
    if (g_default == be)
        g_default = NULL;
    g_epoch++;
    pthread_mutex_unlock(&g_reg_lock);
    if (be->trace && be->opt.trace_path) {
        FILE *f = fopen(be->opt.trace_path, "a");
        if (f) {
            for (i = 0; i < be->ntrace && i < MAX_TRACE; i++)
                fprintf(f, "%u %d %d %.6f %.6f %.6f %.6f\n", be->id, be->trace[i].tid, be->trace[i].poc, be->trace[i].t_start, be->trace[i].t_hook,
                        be->trace[i].t_issued, be->trace[i].t_end);
            fclose(f);
        }
    }
    free(be->trace);
    pool_prefetch_stop(be);
    if (be->opt.base_layer && be->root)         /* the store outlives this back end: give its pictures back */
        for (i = 0; i < be->nbufs; i++) {
            ohevc_tables_unregister_picture(be->root, be->bufs[i].slot);
            ohevc_pic_release(be->root, be->bufs[i].slot);
        }
    for (i = 0; i < be->nall; i++)
        ohevc_ctx_destroy(be->all[i]);
    free(be->all);
    if (be->root)
        ohevc_ctx_destroy(be->root);
    pool_close(be);                 /* (after the contexts: no copy into a block is in flight) */
    for (k = 0; k < 4; k++)
        if (t_ctxs[k].be == be)
            t_ctxs[k].be = NULL;
    if (t_be == be) {
        t_be = NULL;
        t_ctx = NULL;
        t_frame_open = 0;
    }
    be->magic = 0;
    pthread_mutex_destroy(&be->lock);
    free(be);
}

/* ---- frame-parallel decoding over processes (hip_frames.h) ---- */
int ohhip_backend_frames_mode(ohhip_backend *be, const ohhip_frames_mode *m)
{
    if (!be)
        return -1;
    be->fm_on = 0;
    be->fm_index = 0;
    be->fm_segment = -1;
    if (!m)
        return 0;
    /* struct_size first (ohevc_frames.h): a caller compiled against an older header has fewer trailing fields - they read as NULL / 0 here and
     * the caller's memory is never read beyond what it says it has */
    if (m->struct_size < offsetof(ohhip_frames_mode, await_planes) + sizeof(m->await_planes) || m->struct_size > sizeof(ohhip_frames_mode)) {
        fprintf(stderr, "ohhip: ohhip_backend_frames_mode: ohhip_frames_mode.struct_size is %zu (this library: %zu)\n", m->struct_size, sizeof(ohhip_frames_mode));
        return -1;
    }
    memset(&be->fm, 0, sizeof(be->fm));
    memcpy(&be->fm, m, m->struct_size);
    be->fm.struct_size = sizeof(be->fm);
    m = &be->fm;
    if (m->world < 1 || m->rank < 0 || m->rank >= m->world || !m->publish || !m->subscribe || !m->await_motion || !m->await_planes)
        return -1;
    be->fm_on = m->world > 1;
    return 0;
}

/* hls_slice_data (hevc.c:3017-3095) hands the slice data to avctx->execute (one decoding thread) or execute2 (slice threads) and
 * takes the last CTB address from ret[]: a remote picture reports "all CTBs done" without parsing anything */
static int frames_execute(AVCodecContext *c, int (*func)(AVCodecContext *, void *), void *arg, int *ret, int count, int size)
{
    ohhip_backend *be = backend_of(c);
    int i;
    if (!t_remote)
        return be->execute(c, func, arg, ret, count, size);
    for (i = 0; ret && i < count; i++)
        ret[i] = INT_MAX / 2;
    return 0;
}
static int frames_execute2(AVCodecContext *c, int (*func)(AVCodecContext *, void *, int, int), void *arg, int *ret, int count)
{
    ohhip_backend *be = backend_of(c);
    int i;
    if (!t_remote)
        return be->execute2(c, func, arg, ret, count);
    for (i = 0; ret && i < count; i++)
        ret[i] = INT_MAX / 2;
    return 0;
}
void ohhip_backend_frames_install(ohhip_backend *be, AVCodecContext *avctx)
{
    if (be && avctx->execute != frames_execute) {
        be->execute = avctx->execute;
        be->execute2 = avctx->execute2;
        avctx->execute = frames_execute;
        avctx->execute2 = frames_execute2;
    }
}

int ohhip_backend_frame_is_local(ohhip_backend *be, const unsigned char *data0)
{
    int i, local = 1;
    if (!be)
        return 1;
    pthread_mutex_lock(&be->lock);
    if ((i = find_buf_locked(be, data0)) >= 0)
        local = !be->bufs[i].remote;
    pthread_mutex_unlock(&be->lock);
    return local;
}

/* the wait of the reference's frame threads for the rows their motion vectors point at (hevc_await_progress, hevc.c:1951-1958),
 * per picture: every reference picture of the frame that is about to launch must have its samples in this process's store */
static int frames_await_planes(ohhip_backend *be, HEVCContext *s)
{
    int t, k;
    for (t = 0; t < NB_RPS_TYPE; t++)
        for (k = 0; k < s->rps[t].nb_refs; k++) {
            const HEVCFrame *ref = s->rps[t].ref[k];
            int i, index = -1, slot = -1, have = 0, height = 0, need, bi = -1;
            if (t != ST_CURR_BEF && t != ST_CURR_AFT && t != LT_CURR)
                continue;
            if (!ref || ref == s->ref || !ref->frame || !ref->frame->data[0])
                continue;
            pthread_mutex_lock(&be->lock);
            for (i = 0; i < be->nbufs; i++)
                if (be->bufs[i].data0 == ref->frame->data[0] && be->bufs[i].poc == ref->poc && be->bufs[i].seq == ref->sequence &&
                    be->bufs[i].remote && !be->bufs[i].have_planes) {
                    bi = i;
                    index = be->bufs[i].index;
                    slot = be->bufs[i].slot;
                    have = be->bufs[i].have_rows;
                    height = be->bufs[i].h;
                }
            pthread_mutex_unlock(&be->lock);
            if (index < 0)
                continue;
            /* With a transport that moves pictures in bands of CTU rows (await_rows), wait for the rows this picture's motion compensation reads
             * - the recorder kept the deepest one per reference picture, filter taps included - not for the whole picture: what
             * hevc_await_progress (hevc.c:1951-1958) does per prediction block.  A picture of the Curr sets nothing was predicted from is
             * not waited for at all (a later picture that does predict from it asks again). */
            need = be->fm.await_rows ? ohevc_frame_ref_reach(t_ctx, slot) : height;
            if (need < 0 || need < have)
                continue;
            if (need >= height - 1)
                need = -1;                                   /* all of it */
            {
                const int rc = be->fm.await_rows ? be->fm.await_rows(be->fm.user, index, t_ctx, slot, need) : be->fm.await_planes(be->fm.user, index, t_ctx, slot);
                if (rc < 0) {
                    fprintf(stderr, "ohhip: the planes of remote picture %d did not arrive: %s\n", index, ohevc_last_error());
                    return -1;
                }
                if (rc > 0)
                    need = -1;                                   /* the bands asked for happened to be the last ones: the picture is complete */
            }
            pthread_mutex_lock(&be->lock);
            if (be->bufs[bi].index == index) {
                if (need < 0)
                    be->bufs[bi].have_planes = 1;
                else
                    be->bufs[bi].have_rows = need + 1;
            }
            pthread_mutex_unlock(&be->lock);
        }
    return 0;
}

/* A failure is reported ONCE, by the frame end of the picture it happened in (t_error: set by hooks on the picture's own thread), and then
 * forgotten: the pictures that predict from the failed one fail on their own (the library marks it, ohevc_frame_abort), and after the next
 * IDR picture the decoder is whole again - a damaged access unit must not silence the rest of the stream.  Failures of threads that own no
 * picture (slice workers) are the instance's and surface at the next frame end of that instance. */
static int take_error(ohhip_backend *be)
{
    int e = t_error;
    t_error = 0;
    if (be && be->error) {
        be->error = 0;
        e = 1;
    }
    return e ? -1 : 0;
}

/* INTEGRATION.md section 3, last row: run the recorded jobs, copy the picture back for output.  Runs on the thread that
 * decoded the picture: called by the application after avcodec_decode_video2 (one decoding thread) or from the decoder's own
 * end-of-frame progress report (frame threads, below). */
static int frame_done(ohhip_backend *be)
{
    int st, async;
    double t0, t1, t_issued = 0;
    ohevc_frame_stats fs;
    if (!be)
        return 0;
    if (!t_frame_open || t_be != be)
        return take_error(be);
    /* restore_tqb_pixels (hevc_filter.c:163-193) ran on host pixels nobody reads: hand its map to the back-end instead */
    if (t_s && t_s->sps && t_s->pps && t_s->is_pcm &&
        (t_s->pps->transquant_bypass_enable_flag || (t_s->sps->pcm_enabled_flag && t_s->sps->pcm.loop_filter_disable_flag)) &&
        ohevc_tables_set_bypass_map(t_ctx, t_s->is_pcm, t_s->sps->min_pu_width, t_s->sps->min_pu_height,
                                    t_s->sps->log2_min_pu_size) != OHEVC_OK)
        t_error = 1;
    if (be->fm_on && t_s && frames_await_planes(be, t_s) < 0)
        t_error = 1;
    t0 = now_s();
    if (t_s && t_s->sps && t_s->pps && bulk_filters(t_s) && derive_filters(t_s) != OHEVC_OK) {
        fprintf(stderr, "ohhip: filter derivation failed: %s\n", ohevc_last_error());
        t_error = 1;
    }
    t_frame_open = 0;
    /* Frame threads: the issue of the frame end (stage, upload, launches) and the copy-back leave the decoding thread (ohevc_frame_end_async);
     * the picture's samples are waited for where it leaves the decoder (ohhip_backend_fetch_output).  Not with the decoded-picture-hash check
     * on (hevc.c:4146-4162 reads the host planes in this thread right behind this call) and not in frames mode over processes (the picture is
     * exported right below). */
    async = be->opt.async_issue > 0;
    if (async && ((t_s && t_s->decode_checksum_sei) || be->fm_on || !ohevc_ctx_has_device(t_ctx)))
        async = 0;
    ohevc_ctx_set_option(t_ctx, OHEVC_OPT_PARK_FRAMES, be->fm_on ? 0 : be->opt.park_frames);      /* frames mode over processes exports the picture right below: no parking */
    /* The copy-back: 1 = issued and waited for here (defer_download 0); 0 = left to ohhip_backend_fetch_output; 2 (round 6) = QUEUED here behind
     * the picture's device work, on this thread's stream, and only WAITED for at fetch - the application's thread, which every picture of
     * every decoding thread passes through in output order, no longer issues three copies and a wait per picture (0.12 ms of its 0.39 ms per
     * picture on the encoder-like stream at 16 frame threads, 0.66 of 1.2 ms on the all-intra one: profiles/r6y_app_thread_split.txt).  Only
     * into frame buffers of the back end's own (page-locked for good), without parked frame ends, not in frames mode. */
    {
        int mode = !be->opt.defer_download;
        if (!async && !mode && be->opt.queue_download && be->opt.park_frames <= 0 && !be->fm_on && t_s && t_s->ref && t_s->ref->frame &&
            ohevc_ctx_has_device(t_ctx) && pool_owns(be, t_s->ref->frame->buf[0]))
            mode = 2;
        st = async ? ohevc_tables_end_frame_async(t_ctx, 1) : ohevc_tables_end_frame2(t_ctx, mode, &t_issued);
        if (mode == 2 && st == OHEVC_OK) {
            int i;
            pthread_mutex_lock(&be->lock);
            if ((i = find_buf_locked(be, t_s->ref->frame->data[0])) >= 0)
                be->bufs[i].copy_queued = 1;
            pthread_mutex_unlock(&be->lock);
        }
    }
    if (async)
        __atomic_store_n(&be->async_used, 1, __ATOMIC_RELAXED);
    t1 = now_s();
    if (be->fm_on && t_publish && be->opt.test_fail_index >= 0 && be->opt.test_fail_index == t_publish_index)
        st = OHEVC_ERR_STATE;                   /* fault injection of tests/test_dist_cpu.py: the owner fails on this picture */
    if (st == OHEVC_OK && ohevc_frame_get_stats(t_ctx, &fs) == OHEVC_OK) {
        pthread_mutex_lock(&g_prof_lock);
        g_end_frame_s += t1 - t0;
        g_counts[0]++;
        g_counts[1] += fs.launches; g_counts[2] += fs.n_tu; g_counts[3] += fs.n_mc; g_counts[4] += fs.n_intra;
        g_counts[5] += fs.n_dbk; g_counts[6] += fs.n_sao; g_counts[7] += fs.upload_bytes;
        g_alg_bytes += fs.alg_bytes;
        pthread_mutex_unlock(&g_prof_lock);
    }
    if (be->trace) {
        int k;
        pthread_mutex_lock(&be->lock);
        k = be->ntrace < MAX_TRACE ? be->ntrace++ : -1;
        pthread_mutex_unlock(&be->lock);
        if (k >= 0) {
            static int next_tid;
            static __thread int my_tid;
            if (!my_tid)
                my_tid = __sync_add_and_fetch(&next_tid, 1);
            ohhip_trace_rec r = { my_tid, t_trace_poc, t_trace_start, t0, t_issued ? t_issued : t1, t1 };
            be->trace[k] = r;
        }
    }
    if (st == OHEVC_OK)
        st = ohevc_tables_status(t_ctx);
    if (st != OHEVC_OK) {
        fprintf(stderr, "ohhip: frame failed (%d): %s\n", st, ohevc_last_error());
        if (be->fm_on && t_publish)
            publish_failed(be);
        take_error(be);
        return -1;
    }
    if (be->fm_on && t_publish && t_s && t_s->ref) {
        /* hand the picture to the other processes: ohevc_pic_export orders its copy behind the picture's `written` event (and waits
         * for it), so this works with and without the deferred copy-back */
        const int i = t_publish - 1;
        t_publish = 0;
        if (be->fm.publish(be->fm.user, be->bufs[i].index, t_ctx, be->bufs[i].slot, t_s->ref->tab_mvf,
                           (size_t)t_s->sps->min_pu_width * t_s->sps->min_pu_height * sizeof(MvField), 0) != 0) {
            fprintf(stderr, "ohhip: publishing picture %d failed: %s\n", be->bufs[i].index, ohevc_last_error());
            t_error = 1;
        }
    }
    return take_error(be);
}

int ohhip_backend_frame_done(ohhip_backend *be) { return frame_done(be ? be : t_be); }

/* hip_frames.h: the decoder gave up on the picture it was decoding */
int ohhip_backend_frame_failed(ohhip_backend *be)
{
    if (!be)
        be = t_be;
    if (t_frame_open && t_ctx && t_be == be) {
        t_frame_open = 0;
        ohevc_frame_abort(t_ctx);
    }
    if (be && be->fm_on && t_publish && t_be == be)
        publish_failed(be);
    t_error = 0;
    return 0;
}

/* INTEGRATION.md section 3, last row, one decoding thread: hevc_decode_frame checks the decoded-picture-hash SEI on the HOST planes
 * right after decode_nal_units (hevc.c:4146-4162; `decode-checksum` option, libOpenHevcSetCheckMD5).  The frame-end hook belongs
 * in front of that check; its first statement is the only av_pix_fmt_desc_get call of hevc.c that runs (hevc.c:4148), so that call
 * site carries the hook here.  (With frame threads the frame has already ended in ohhip_report_progress.) */
const AVPixFmtDescriptor *ohhip_pix_fmt_desc_get(enum AVPixelFormat pix_fmt)
{
    HEVCContext *s = t_s;
    ohhip_backend *be = t_be;
    if (be && frame_done(be) < 0)
        t_error = 1;                                                /* (reported by the application's own ohhip_backend_frame_done call) */
    if (be && be->opt.defer_download && s && s->ref && s->ref->frame)           /* the check reads the host planes now */
        if (ohhip_backend_fetch_output(be, s->ref->frame->data, s->ref->frame->linesize) < 0)
            t_error = 1;
    return av_pix_fmt_desc_get(pix_fmt);
}

/* frame threads: decode_nal_units() ends with ff_thread_report_progress(&s->ref->tf, INT_MAX, 0) (hevc.c:4026-4027),
 * the moment other threads may consider the picture complete -- the frame-end hook of INTEGRATION.md section 3.  The
 * call sites in hevc.c are renamed to this wrapper; the per-row reports of hevc_filter.c are untouched. */
void ohhip_report_progress(ThreadFrame *f, int progress, int field)
{
    /* what other threads wait for on the CPU (motion fields, DPB state) is complete now; the samples are ordered on the
     * device by the library (ohevc_ctx_create_shared), so the report need not wait for the GPU */
    ff_thread_report_progress(f, progress, field);
    if (progress == INT_MAX && t_be && frame_done(t_be) < 0)
        t_be->error = 1;             /* nobody reads this thread's return value: the instance's next frame end reports it */
}

/* INTEGRATION.md section 3, "before output".  With defer_download ending a frame only ISSUES its device work and the
 * copy-back happens here, when the application takes the picture out of the decoder (the harness calls this right after
 * avcodec_decode_video2 handed it a frame; in openHEVC proper the place is libOpenHevcGetOutput, openHevcWrapper.c:353-398):
 * parsing of the next picture then overlaps the device work of this one even with a single decoding thread, and pictures
 * that are never output are never copied.  (Not inside ff_hevc_output_frame: with no reordering the decoder "outputs" the
 * current picture at hevc_frame_start, before it is decoded, and relies on the shared buffer being filled afterwards.) */
int ohhip_backend_fetch_output(ohhip_backend *be, uint8_t *const data[3], const int linesize[3])
{
    ohevc_ctx *ctx;
    int i, slot = -1, queued = 0;
    if (!be)
        be = t_be;
    if (!be || (!be->opt.defer_download && !__atomic_load_n(&be->async_used, __ATOMIC_RELAXED)) || !be->root || !data[0])
        return 0;
    ctx = t_be == be && t_ctx ? t_ctx : be->root;
    pthread_mutex_lock(&be->lock);
    if ((i = find_buf_locked(be, data[0])) >= 0) {
        slot = be->bufs[i].slot;
        queued = be->bufs[i].copy_queued;
        be->bufs[i].copy_queued = 0;
    }
    pthread_mutex_unlock(&be->lock);
    if (slot < 0) {
        fprintf(stderr, "ohhip: output picture is not in the picture store\n");
        return -1;
    }
    if (queued) {                               /* the copy-back was queued at the picture's frame end: wait until it has landed */
        if (ohevc_tables_fetch_picture(ctx, slot) != OHEVC_OK) {
            fprintf(stderr, "ohhip: queued copy-back failed: %s\n", ohevc_last_error());
            return -1;
        }
        return 0;
    }
    if (__atomic_load_n(&be->async_used, __ATOMIC_RELAXED) && !be->opt.defer_download) {     /* the copy-back was queued by the issuer: wait until it has landed */
        if (ohevc_tables_fetch_picture(ctx, slot) != OHEVC_OK || ohevc_ctx_async_status(ctx) != OHEVC_OK) {
            fprintf(stderr, "ohhip: asynchronous frame end failed: %s\n", ohevc_last_error());
            return -1;
        }
        return 0;
    }
    {
        void *const host[3] = { data[0], data[1], data[2] };
        const ptrdiff_t strides[3] = { linesize[0], linesize[1], linesize[2] };
        if (ohevc_pic_download_planes(ctx, slot, host, strides) != OHEVC_OK) {
            fprintf(stderr, "ohhip: download failed: %s\n", ohevc_last_error());
            return -1;
        }
    }
    return 0;
}

/* hevc_await_progress() (hevc.c:1951-1958) makes a frame thread wait until the rows its motion vectors point at have
 * been RECONSTRUCTED by the thread decoding the reference picture.  With the GPU back-end no sample is read on the CPU
 * and the device-side ordering is per picture, so this wait is only lost time: the call sites in hevc.c are renamed to
 * this no-op.  (The waits for collocated motion vectors in hevc_mvs.c are untouched.) */
void ohhip_await_progress(ThreadFrame *f, int progress, int field)
{
    /* SHVC: the one wait that is about HOST data - the enhancement layer scales the base-layer picture's motion field into the inter-layer
     * picture's (ff_upscale_mv_block, hevc_filter.c:1312-1375) once the base-layer decoder's thread has parsed those rows
     * (hevc_await_progress_bl, hevc.c:1959-1966) */
    if (t_bl_tf && f == t_bl_tf)
        ff_thread_await_progress(f, progress, field);
}

/* SHVC, quality (SNR) scalability: at ratio 1 ff_upsample_block (hevc_filter.c:1377-1430; call sites hevc.c:2082,2097) copies the base-layer
 * CTB into the inter-layer reference picture with memcpy (copy_block, hevc_filter.c:1165-1173,1189-1192,1262-1266) - no table slot is called,
 * so the device picture would stay empty.  The wrapper lets the reference do its host work (the motion field of the inter-layer picture) and,
 * at ratio 1, resamples the picture on the device once: the general filter at phase 0 is the identity ({0,0,0,64,0,0,0,0}, 64 * 64 * x + 2048 >> 12). */
void ohhip_upsample_block(HEVCContext *s, HEVCFrame *ref0, int x0, int y0, int nPbW, int nPbH)
{
    ff_upsample_block(s, ref0, x0, y0, nPbW, nPbH);
    if (s->up_filter_inf.idx == SNR && s->BL_frame && ref0 && ref0->frame) {      /* (also from a slice worker: ohhip_cabac_init bound it to the picture's context) */
        ohevc_UpsamplInf u;
        /* the reference's ratio-1 path (copy_block, hevc_filter.c:1165-1192) copies base-layer (x0, y0) to the same position with memcpy: the
         * scaled reference layer offsets of the SPS play no part in it - a zero window makes the general filter that copy */
        ohevc_HEVCWindow win = { 0, 0, 0, 0 };
        /* a COPY whatever cross_layer_phase_alignment_flag says (the reference tests the scale alone, hevc.c:486-487): the offsets of phase
         * alignment 0 (set_sps, hevc.c:476-484: the chroma row offset of a quarter sample is taken back by the "- 4" of the chroma mapping) */
        u.scaleXLum = u.scaleYLum = u.scaleXCr = u.scaleYCr = 65536;
        u.addXLum = u.addYLum = u.addXCr = 1 << 11;
        u.addYCr = ((1 * 65536 + 2) >> 2) + (1 << 11);
        u.idx = DEFAULT;
        if (ohevc_tables_upsample_frame(ref0->frame->data[0], s->BL_frame->frame->data[0], &win, &u) != OHEVC_OK)
            note_error(backend_of(s->avctx));
    }
}

/* ---- process-wide profiling (all instances) ---- */
/* cumulative since the last call: seconds inside the frame-end hook and job / launch / upload counters */
void ohdec_backend_profile(double *end_frame_s, long long counts[8])
{
    ohhip_backend *b;
    pthread_mutex_lock(&g_reg_lock);
    pthread_mutex_lock(&g_prof_lock);
    for (b = g_backends; b; b = b->next)
        if (b->root && __atomic_load_n(&b->async_used, __ATOMIC_RELAXED)) {           /* the issuer's seconds belong to the frame ends too (they just do not block a decoding thread) */
            double bs = 0;
            long long fr = 0;
            if (ohevc_ctx_async_profile(b->root, &bs, &fr) == OHEVC_OK) {
                g_end_frame_s += bs - b->issuer_s0;
                b->issuer_s0 = bs;
                b->issuer_f0 = fr;
            }
        }
    *end_frame_s = g_end_frame_s;
    memcpy(counts, g_counts, sizeof(g_counts));
    g_end_frame_s = 0;
    memset(g_counts, 0, sizeof(g_counts));
    pthread_mutex_unlock(&g_prof_lock);
    pthread_mutex_unlock(&g_reg_lock);
}

/* algorithmic HBM bytes of the jobs recorded since the last call (the device-side traffic floor of those pictures) */
long long ohdec_backend_alg_bytes(void)
{
    long long v;
    pthread_mutex_lock(&g_prof_lock);
    v = g_alg_bytes;
    g_alg_bytes = 0;
    pthread_mutex_unlock(&g_prof_lock);
    return v;
}
