// ctx.hip -- host side of libohevc_hip.so: picture store, job recorder, phase-ordered executor (include/ohevc_ctx.h).
// Host code only; the kernels it launches live in the *_kernels.hip files and are reached through the same
// ohevc_dev_* entry points an external caller would use.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <atomic>
#include <string.h>
#include <vector>
#include "common.hpp"
#include "ohevc_ctx.h"
#include "ohevc_debug.h"

namespace {

struct Picture {
    bool used = false, owned = true;
    bool single = false;                  // the three planes are one allocation, back to back (alloc_picture): one copy moves the picture
    int w = 0, h = 0, cfi = 1, bd = 8;
    ohevc_plane planes[3] = {};
    // cross-ctx ordering (contexts of several decoding threads share one store and run on their own streams):
    bool end_issued = true;               // false between frame_begin and the frame_end that reconstructs this picture
    bool failed = false;                  // that frame_end gave up (ohevc_frame_abort): dependents fail at once instead of waiting
    hipEvent_t written = nullptr;         // recorded on the writer's stream by that frame_end
    std::vector<hipEvent_t> readers;      // frame-end events of pictures that read this one since it was written
    // asynchronous frame ends (ohevc_frame_end_async): the copy-back into the application's planes
    // A slot holds one picture after the other.  `gen` counts them (frame_begin), `issued_gen` is the newest one whose frame end has been
    // issued: a queued frame end names the VERSION of each reference picture it reads, because the decoder may recycle a reference's
    // buffer - and begin a new picture in its slot - once the thread that decoded the reader is done with it, i.e. before the reader's
    // frame end has been issued.
    uint32_t gen = 0, issued_gen = 0;
    bool host_copy_issued = true;         // false between the submission of a frame end with a copy-back and the issue of that copy
    hipEvent_t host_copy = nullptr;       // fires when the copy has landed
};

struct PicStore;
// the thread that issues asynchronous frame ends (ohevc_frame_end_async, below)
struct Issuer {
    std::vector<std::thread> th;           // OHEVC_ISSUER_THREADS of them (default 4): one thread issues ~1000 1080p frame ends per second
    std::vector<ohevc_ctx *> executing;    // frames taken from the queue whose frame end has not been published yet
    std::mutex m;
    std::condition_variable cv;
    std::deque<ohevc_ctx *> queue;         // executor contexts holding a submitted frame, in submission order
    std::vector<ohevc_ctx *> execs;        // all executor contexts (free ones have exec_busy == false)
    int in_flight = 0;
    bool stop = false;
    int error = OHEVC_OK;                  // sticky: first failure of an asynchronous frame end (ohevc_ctx_async_status)
    char error_text[256] = {};
    double busy_s = 0;                     // seconds the issuer spent issuing (OHEVC_TRACE=timing)
    long frames = 0;
    int device = 0;
    struct PicStore *store = nullptr;
};

// The device picture store = the decoded picture buffer.  One per ohevc_ctx_create, shared by ohevc_ctx_create_shared.
constexpr int kMaxPics = OHEVC_MAX_PICTURES;      // (ohevc_ctx.h; every per-slot table below and in tables.hip / hip_hooks.c is sized by it)
struct PicStore {
    std::mutex m;
    std::condition_variable cv;           // signalled when a picture's end_issued turns true
    Picture pics[kMaxPics];               // fixed array: pointers to entries stay valid while other threads allocate
    std::atomic<int> npics{0};            // grows under `m`; read without it by every context of the store (get_pic)
    unsigned version = 0;                 // bumped whenever a slot's planes change (contexts re-upload their MC table)
    // Page locks: taken and dropped under the exclusive lock; a copy-back into application memory holds the shared lock from its issue to
    // its completion, so dropping a page lock (which first drains the device) can never pull a range from under a copy in flight.
    std::shared_mutex pin_m;
    std::vector<std::pair<uintptr_t, size_t>> pinned;      // host ranges page-locked through ohevc_host_pin
    Issuer *issuer = nullptr;             // ohevc_frame_end_async: the thread that issues frame ends (created by the first submission)
    // Device pictures come in batches: one hipMalloc, one memset and one wait for 4, 8, 16, 32 pictures of a size instead of one of each per
    // picture.  A decoder's pool of frame buffers grows through its first dozens of pictures, every new buffer wants a device picture, and the
    // sample hooks ask for it in the serial prologue of the picture (hevc_frame_start, before the next access unit is let in): 0.3-0.5 ms of
    // driver calls there spaced a fresh decoder's picture starts 0.8-1.2 ms apart instead of 0.43 (profiles/r13_*).  Pieces are zeroed when their
    // batch is made; a piece whose picture is released waits in `dirty` and is zeroed when it is handed out again (take_piece), before any new
    // batch is made.
    std::mutex spare_m;
    struct Spare { std::vector<unsigned char *> pieces, dirty; int next_batch = 4; };
    std::map<size_t, Spare> spare;        // by piece size: zeroed pieces nobody uses yet (two layers of an SHVC stream share a store: two sizes take turns)
    std::vector<void *> batches;          // the allocations behind all pieces ever made
    // every stream of every live context of this store: what "wait until nothing of this decoder is in flight" means (store_sync) - the other
    // decoders of the process, on their own stores and streams, are not waited for (hipDeviceSynchronize used to do that)
    std::mutex streams_m;
    std::vector<hipStream_t> streams;
};
static void store_add_stream(PicStore &st, hipStream_t s) { if (s) { std::lock_guard<std::mutex> g(st.streams_m); st.streams.push_back(s); } }
static void store_remove_stream(PicStore &st, hipStream_t s)
{
    std::lock_guard<std::mutex> g(st.streams_m);
    st.streams.erase(std::remove(st.streams.begin(), st.streams.end(), s), st.streams.end());
}
static hipError_t store_sync(PicStore &st)
{
    std::lock_guard<std::mutex> g(st.streams_m);          // (held throughout: a context that dies meanwhile waits with destroying its streams)
    hipError_t rc = hipSuccess;
    for (hipStream_t s : st.streams) { const hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess && rc == hipSuccess) rc = e; }
    return rc;
}

struct DevBuf {                       // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return OHEVC_OK;
        if (p) OHEVC_HIP_TRY(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = std::max(n, (size_t)1 << 20);
        want = (want + (want >> 1) + 255) & ~(size_t)255;
        OHEVC_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return OHEVC_OK;
    }
};

struct PinnedBuf {                    // grow-only pinned host staging buffer
    unsigned char *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n, size_t at_least = (size_t)1 << 20)
    {
        if (n <= cap) return OHEVC_OK;
        if (p) OHEVC_HIP_TRY(hipHostFree(p));
        p = nullptr; cap = 0;
        size_t want = std::max(n, at_least);
        want = (want + (want >> 1) + 255) & ~(size_t)255;
        OHEVC_HIP_TRY(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
        cap = want;
        return OHEVC_OK;
    }
};

// jobs of one intra dependency level (level 0 = residuals of inter blocks).  Bins keep their capacity from picture to
// picture; `touched` lists the (size, kind) bins in use so that clearing and staging never walk the empty ones.
struct LevelBins {
    std::vector<ohevc_tu_job> tu[4][OHEVC_TU_NKINDS];
    std::vector<ohevc_intra_job> intra;
    std::vector<ohevc_tu_job> intra_res;   // parallel to intra: the block's own residual (reserved0 = kind + 1) or zeros (ohevc_dev_intra_recon_batch)
    uint64_t touched = 0;             // bit (log2 - 2) * 16 + kind
};

}  // namespace

static void async_drain(PicStore &st);
static void issuer_shutdown(PicStore &st);
static void issuer_help(PicStore &st);
static void settle_slot(ohevc_ctx *c, int slot);
static inline Issuer *get_issuer(PicStore &st) { return __atomic_load_n(&st.issuer, __ATOMIC_ACQUIRE); }
static std::atomic<uint64_t> g_ctx_gen{1};
// executor of the intra-coded blocks (ohevc_debug_set_level_launch):
//   0  one prediction launch and one residual launch per dependency level;   1  all levels inside one ohevc_dev_levels launch;
//   3  one ohevc_dev_ctbs launch per picture: CTBs as tasks, their samples in LDS, operations in decoding order;
//   0 is the default since round 4 (the chain kernel takes the levels of a picture in one launch or a few; recording both forms costs the
//      parser 1-5 %, profiles/r4q_levelmode_ab_summary.txt).
//   2  both forms are recorded and the cheaper one is chosen per picture from the recorded work itself: the CTB form when
//      its longest chain of dependent CTBs is short (sparse intra blocks: encoder-like inter pictures), the level form otherwise.
// Pictures whose intra jobs name no CTB size always take the level form.
void ohevc_mc_forget_stream(void *stream);      // mc_kernels.hip: per-stream scratch of the MC redo pass
static int g_record_only = 0;        // ohevc_debug_set_record_only (2: record the DEVICE forms - maps instead of per-edge jobs - and drop them: profiling of the recording path on a box without a GPU)
static int g_compact_coeffs = 2;     // ohevc_debug_set_compact_coeffs: 2 = the non-zero 4x4 groups of an inverse-DCT block travel (round 6), 1 = its col_limit rectangle (round 5), 0 = every block whole (rounds 1-4; A/B and tests)
extern "C" int ohevc_debug_set_compact_coeffs(int on) { g_compact_coeffs = on < 0 ? 0 : on > 2 ? 2 : on; return OHEVC_OK; }
static int g_fuse_intra = 1;   // ohevc_debug_set_fuse_intra: a block's residual runs in its prediction's wavefront
static std::atomic<int> g_level_launch{0};      // ohevc_debug_set_level_launch (the sample hooks set it, to the same value, from every decoder that is opened: atomic)
// The widest level a chain takes.  Inside the chain kernel a level costs ~2 us plus ~1.5 us per further pass of its 8-wavefront workgroup; as a
// launch of its own ~6.6 us of kernel plus 2 - 4 us until the next one starts, whatever its width: up to four passes the chain is cheaper.
static int g_intra_chain_waves = 32;             // ohevc_debug_set_intra_chain_limits
static int g_intra_chain_min_run = 2;      // shortest run (levels) worth a chain launch
static int g_intra_chain = 1; // ohevc_debug_set_intra_chain: runs of narrow levels in one launch (ohevc_dev_intra_chain)
static bool g_reverse_levels = false;       // ohevc_debug_set_reverse_levels: the jobs of every level in reverse order (tests: their order must not matter)
extern "C" int ohevc_debug_set_reverse_levels(int on) { g_reverse_levels = on != 0; return OHEVC_OK; }
static int g_intra_pack = 1;   // ohevc_debug_set_intra_pack: the packed intra kernel (N lanes per block) serves the levels
static const bool g_trace_order = ohevc::config().trace_order;         // OHEVC_TRACE=order / timing (common.hpp: Config)
static const bool g_trace_timing = ohevc::config().trace_timing;
// how long a frame thread waits for another thread to issue the frame end of a reference picture before it gives up (a decoding thread
// that died would otherwise hang the pool).  The sanitizer build of the kernel emulator needs minutes where a device needs milliseconds.
static const int g_ref_wait_s = ohevc::config().ref_wait_seconds;
// OHEVC_TRACE=at=plane:x:y: print every recorded job whose block covers that sample (diagnosis of a mismatching block)
static const int *const g_trace_at = ohevc::config().trace_at;
static const bool g_trace_at_on = ohevc::config().trace_at[0] >= 0;
static inline bool trace_hit(int plane, int x, int y, int w, int h)
{
    return g_trace_at_on && plane == g_trace_at[0] && g_trace_at[1] >= x && g_trace_at[1] < x + w && g_trace_at[2] >= y && g_trace_at[2] < y + h;
}
static void trace_dbk(int target, const ohevc_dbk_job &j)
{
    const bool v = j.flags & OHEVC_DBK_VERTICAL_EDGE;      // an edge segment touches up to 4 samples either side of its line
    if (trace_hit(j.plane, v ? j.x - 4 : j.x, v ? j.y : j.y - 4, v ? 8 : 8, v ? 8 : 8))
        fprintf(stderr, "trace: target %d dbk plane %d x %d y %d flags 0x%x beta %d tc %d %d\n", target, j.plane, j.x, j.y, j.flags, j.beta, j.tc[0], j.tc[1]);
}
static void trace_sao(int target, const ohevc_sao_job &j)
{
    if (trace_hit(j.plane, j.x - 1, j.y - 1, j.w + 2, j.h + 2))
        fprintf(stderr, "trace: target %d sao plane %d x %d y %d w %d h %d type %d klass %d borders 0x%x restore %d edges 0x%x quirks 0x%x off %d %d %d %d\n", target,
                j.plane, j.x, j.y, j.w, j.h, j.type, j.klass, j.borders, j.restore, j.edges, j.quirks, j.offset_val[1], j.offset_val[2], j.offset_val[3], j.offset_val[4]);
}
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// What the ohevc_rec_* calls fill.  The context itself is one; with ohevc_ctx_set_concurrent every further thread that
// records into the context (the reference's slice threads: WPP rows / tiles of ONE picture) gets a private one, merged into
// the context's own at the next frame_reconstruct -- no lock and no shared cache line on the recording path.
struct Rec {
    std::vector<ohevc_mc_job> mc, mc_small;              // tiles of at most 16x16 / at most 8x8 samples
    std::vector<LevelBins> levels;                         // [level]; entries 0..max_level are live
    int max_level = -1;
    // intra work in recording (= decoding) order, for the CTB executor (ohevc_dev_ctbs): one word per operation as the kernel reads it
    // (bit 31 residual / prediction, size, kind, index into the arrays below) next to the CTB it belongs to
    std::vector<ohevc_intra_job> ctb_intra;
    std::vector<ohevc_tu_job> ctb_tu;
    std::vector<std::pair<uint32_t, uint32_t>> ctb_ops;    // (CTB raster index, operation word)
    // Coefficients cross the bus COMPACT: of an inverse-DCT block only the top-left cols x rows rectangle that can hold non-zero coefficients (the
    // reference computes the bound from the last significant coefficient, hevc_cabac.c:1923-1934, and its own transforms skip what lies
    // outside, hevcdsp_template.c:271-277,288-291); everything else whole.  `coeffs` is that stream, `expand` says where each piece goes in the
    // DENSE arena the kernels index (ohevc_tu_job.coeff_off: block-major N x N int16, as before) - the device rebuilds it (ohevc_dev_expand_coeffs).
    std::vector<int16_t> coeffs;
    std::vector<ohevc_expand_rec> expand;
    uint32_t dense = 0;                                    // int16 elements of the dense arena so far
    std::vector<ohevc_intra_cip> cips;                     // side records of constrained-intra jobs
    std::vector<ohevc_dbk_job> dbk_v, dbk_h;
    std::vector<ohevc_bs_call> bs_calls;                   // ohevc_rec_bs_call: the picture's calls of ff_hevc_deblocking_boundary_strengths (device-side boundary strengths)
    std::vector<ohevc_sao_job> sao;
    bool sao_lagged = false;          // some recorded SAO job carries OHEVC_SAO_LAG_*
    int nstat[5] = {};                // tu, mc, intra, dbk, sao calls
    int64_t alg = 0;                  // algorithmic bytes of the recorded jobs (ohevc_frame_stats.alg_bytes)
    struct { int level = -1, index = 0, plane = 0, x = 0, y = 0, log2 = 0; } last_intra;   // the most recent intra job of this recorder (levels form)
    int16_t reach[OHEVC_MAX_PICTURES + 1];               // [reference slot]: the deepest LUMA row of that picture the recorded motion compensation reads, -1: none (ohevc_frame_ref_reach)
    Rec() { for (int16_t &v : reach) v = -1; }
};

struct ohevc_ctx : Rec {
    bool dry = false;                 // record-only profiling mode: no device, no pixels (ohevc_debug.h)
    bool dry_as_device = false;       // ... that records what a context WITH a device records (ohevc_debug_set_record_only(2))
    std::vector<int16_t> dense_host;  // ohevc_debug_arena: the dense arena for host-side consumers of the recorded jobs
    int device = 0;
    hipStream_t stream = nullptr;     // where this context's frames are issued: stream_norm, or - pictures with long dependency chains - stream_long (select_stream)
    hipStream_t stream_norm = nullptr, stream_long = nullptr;
    hipEvent_t switch_ev = nullptr;   // hand-over between the two: what was issued on the one is ordered before what follows on the other
    // Two upload lanes - host staging buffer, device buffer, "copied" event - one for the job arrays of ohevc_frame_reconstruct, one for the
    // filter maps of the frame end.  With one lane the second staging copy of a picture had to wait on the host until the first H2D copy
    // had run, and that copy sits in the stream BEHIND the waits for the reference pictures' completion: under frame threads every
    // decoding thread stood still in the middle of its frame end until its references were reconstructed on the device.
    hipEvent_t staged[2] = {nullptr, nullptr};      // recorded after the last H2D copy out of stage[k]
    bool staged_pending[2] = {false, false};
    // The uploads run on a stream of their own.  In `stream` they sat behind the waits for the reference pictures' frame ends
    // (hipStreamWaitEvent on events of other decoding threads' streams), and hipMemcpyAsync behind an unresolved cross-stream wait does not
    // return on this runtime until the wait is over - and then only after a wake-up latency of ~4 ms during which NOTHING is submitted
    // (profiles/r5b_*: three decoding threads inside hipMemcpyAsync for 7.7-9.8 ms, the device idle for the last 4.1 ms of it, once per
    // level of the GOP's reference hierarchy).  An upload depends on nothing but the earlier readers of its device buffer (lane_done).
    hipStream_t up_stream = nullptr;
    hipEvent_t lane_done[2] = {nullptr, nullptr};   // recorded in `stream` behind the last kernel that reads d_jobs[k]
    bool lane_done_pending[2] = {false, false};
    std::shared_ptr<PicStore> store;
    unsigned table_version = ~0u;     // store->version the device MC table was built from
    int cur = -1;
    hipEvent_t ring[16] = {};         // frame-end events handed to the store (a re-recorded event only waits longer)
    int ring_next = 0;
    std::vector<int> ref_slots;       // reference pictures the stream already waits for in this frame
    bool target_guarded = false;      // the stream already waits for earlier readers/writers of the target picture
    Picture twin;                     // deblocked copy for SAO (the reference's sao_frame, hevc.c:369-385)
    Picture lag;                      // picture between the two deblocking passes (only for OHEVC_SAO_LAG_* jobs)

    // concurrent recording (ohevc_ctx_set_concurrent)
    bool concurrent = false;
    uint64_t gen = 0;                                      // identity for the per-thread cache (addresses get reused)
    std::atomic<uint64_t> epoch{0};                        // bumped by frame_begin: the owner thread may change from picture to picture
    std::thread::id owner;                                 // the thread that called frame_begin records into the context itself
    std::mutex side_m;
    std::vector<std::pair<std::thread::id, std::unique_ptr<Rec>>> side;

    std::vector<ohevc_intra_chain_level> chain_tab;       // scratch of frame_reconstruct: runs of narrow levels
    std::vector<int> chain_first, chain_len;
    std::vector<ohevc_level_phase> phases;                // scratch of frame_reconstruct
    std::vector<uint32_t> need, sync_zero;
    std::vector<uint8_t> dbk_blob;                         // ohevc_rec_deblock_maps: the copied maps back to back (empty = none)
    ohevc_dbk_maps dbk_maps = {};                          // geometry; the pointers hold offsets into dbk_blob
    ohevc_bs_maps bs_maps = {};                            // device-side boundary strengths: geometry; mvf / cbf_luma hold offsets into dbk_blob
    bool have_bs = false;
    DevBuf d_bs;                                           // the two boundary-strength arrays the kernel fills
    DevBuf d_grid;                                         // ohevc_frame_keep_motion: the motion field rebuilt from the luma MC jobs
    int keep_motion_l2 = 0;                                // log2 of the grid's unit; 0: the frame keeps none
    bool grid_zeroed = false;                              // ... and it has been cleared for this frame
    size_t grid_bs_off = 0, grid_bs_cap = 0;               // ... together with room behind it for the boundary-strength arrays
    std::vector<uint8_t> bypass;                           // ohevc_frame_set_bypass_map: is_pcm bytes, row length bypass_w (empty = none)
    int bypass_w = 0, bypass_l2 = 0, bypass_exact = 0;
    std::vector<uint16_t> level_map[3];
    int lm_w[3] = {}, lm_h[3] = {};
    int recon_lane = 0, last_recon_lane = 0;      // the staging / device buffer pair the next / the last ohevc_frame_reconstruct upload takes
    int flushed_intra = 0;            // ohevc_frame_flush_intra: intra jobs of this frame already handed to the device by an early flush
    bool flush_closed = false;        // ... and no further early flush for this frame (it has inter prediction: its references may not be issued yet)
    int frame_mode = 0;               // the executor of the intra-coded blocks as chosen at frame_begin (one executor per picture)
    int opt[3] = { -1, -1, -1 };          // ohevc_ctx_set_option: OHEVC_OPT_LEVEL_LAUNCH, OHEVC_OPT_FILTERS_ON_DEVICE, OHEVC_OPT_PARK_FRAMES (-1: the process default)
    int log2_ctb = 0;                 // CTB size named by the picture's intra jobs (0: none seen yet, -1: they disagree)
    std::vector<ohevc_ctb_task> ctb_tasks;                // scratch of frame_reconstruct
    std::vector<uint32_t> ctb_opwords, ctb_sync_zero;
    std::vector<int32_t> ctb_task_of;

    // asynchronous frame ends: an EXECUTOR context (owned by the store's issuer) takes over the recorded frame of a decoding thread's context
    bool is_exec = false, exec_busy = false;
    bool parked = false;                                   // executor: the frame it holds was parked by ohevc_frame_end_deferred (statistics are added, not assigned)
    ohevc_frame_stats parked_stats = {};                   // recording context: statistics of its parked frames issued since the last ohevc_frame_get_stats (stats_m)
    long n_parked = 0;
    ohevc_ctx *async_from = nullptr;                       // the context the frame was recorded into (receives the statistics)
    std::vector<std::pair<int, uint32_t>> async_refs;      // (slot, version) of the reference pictures of the queued frame: it is issued once their frame ends are
    uint32_t my_gen = 0;                                   // version of the target picture this context is recording / executing
    void *async_host[3] = {nullptr, nullptr, nullptr};     // copy-back destination, NULL = none
    ptrdiff_t async_stride[3] = {0, 0, 0};
    hipEvent_t dl_ring[8] = {};
    int dl_next = 0;
    DevBuf d_jobs[2], d_dense[2], d_table, d_upsample;
    // SHVC: the tap maps in d_upsample belong to these parameters (a stream resamples every picture with the same ones: one upload per geometry)
    ohevc_upsample_params up_prm = {};
    bool up_valid = false;
    size_t up_off_cols[3] = {}, up_off_colof[3] = {}, up_off_rows[3] = {};
    int up_src_cols[3] = {}, up_src_rows[3] = {};
    PinnedBuf stage[2], table_stage;
    ohevc_frame_stats stats = {}, last_stats = {};
    std::mutex stats_m;                    // last_stats: written by the context's own thread or, for an asynchronous frame end, by the issuer thread; read by ohevc_frame_get_stats
    double t_wait_refs = 0, t_issue = 0;   // OHEVC_TRACE=timing: host seconds blocked on other threads' frame ends / spent issuing
    // the filter maps / records of the frame end, staged by frame_end_impl BEFORE it calls ohevc_frame_reconstruct so that they travel in the
    // same host-to-device copy as the job arrays (tail_base: where they landed in that upload; SIZE_MAX: they did not travel yet)
    std::vector<std::pair<const void *, size_t>> tail_parts;
    size_t tail_total = 0, tail_base = SIZE_MAX;
    double t_f[6] = {0, 0, 0, 0, 0, 0};     // ... the filter calls one by one: boundary strengths, deblocking (vertical), deblocking (horizontal), the deblocked copy, SAO, the rest
    double t_part[5] = {0, 0, 0, 0, 0};    // ... of which: staging copies, copy / launch calls of the reconstruction, the same of the filters, waits for a free staging buffer, copy-back
    int n_frames = 0, n_map_frames = 0;
};

using namespace ohevc;

// store: where a picture that was cut out of a batch (take_piece) gives its piece back.  The caller has made sure that nothing on the device
// still reads or writes the picture (ohevc_pic_release drains; ensure_like synchronises the one stream that used the copy).
static int free_picture(Picture &p, bool dry = false, PicStore *store = nullptr)
{
    if (p.single) {
        if (p.planes[0].data && p.owned && !dry) OHEVC_HIP_TRY(hipFree(p.planes[0].data));
        if (p.planes[0].data && !p.owned && !dry && store) {
            // (a released piece used to be dropped until the store died: every release + alloc pair - a decoder's pool changing geometry, an
            // enhancement layer reopened on a live base store, the scratch copies of ensure_like - grew device memory by one picture)
            const size_t bytes = ((size_t)((unsigned char *)p.planes[2].data - (unsigned char *)p.planes[0].data) + (size_t)p.planes[2].stride * p.planes[2].height + 4095) & ~(size_t)4095;
            std::lock_guard<std::mutex> g(store->spare_m);
            store->spare[bytes].dirty.push_back(static_cast<unsigned char *>(p.planes[0].data));
        }
        for (auto &pl : p.planes) pl = ohevc_plane{};
    } else {
        for (auto &pl : p.planes) {
            if (pl.data && p.owned && !dry) OHEVC_HIP_TRY(hipFree(pl.data));
            pl = ohevc_plane{};
        }
    }
    p.used = false; p.owned = true; p.single = false;
    return OHEVC_OK;
}

// OHEVC_PICTURE_BATCH=0: every picture its own allocation (the AddressSanitizer pass over the emulated device code wants red zones around each)
static const int g_picture_batch = ohevc::config().picture_batch;

// a zeroed piece of `bytes` bytes out of the store's batches (PicStore::spare); nullptr: none to be had, allocate the old way
static unsigned char *take_piece(PicStore &st, size_t bytes, hipStream_t stream)
{
    if (!g_picture_batch || bytes > ((size_t)256 << 20)) return nullptr;
    bytes = (bytes + 4095) & ~(size_t)4095;
    std::lock_guard<std::mutex> g(st.spare_m);
    PicStore::Spare &sp = st.spare[bytes];
    if (sp.pieces.empty() && !sp.dirty.empty()) {
        // a released picture's piece: nothing on the device touches it any more (free_picture's contract); zero it like a fresh batch's
        unsigned char *d = sp.dirty.back();
        if (hipMemsetAsync(d, 0, bytes, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) { sp.dirty.pop_back(); return d; }
        (void)hipGetLastError();
    }
    if (sp.pieces.empty()) {
        const int n = (int)std::max<size_t>(1, std::min<size_t>((size_t)sp.next_batch, ((size_t)1 << 30) / bytes));
        void *m = nullptr;
        if (hipMalloc(&m, n * bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemsetAsync(m, 0, n * bytes, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(m); return nullptr; }
        st.batches.push_back(m);
        for (int i = n - 1; i >= 0; i--) sp.pieces.push_back(static_cast<unsigned char *>(m) + (size_t)i * bytes);
        sp.next_batch = std::min(sp.next_batch * 2, 32);
    }
    unsigned char *d = sp.pieces.back();
    sp.pieces.pop_back();
    return d;
}

extern "C" int ohevc_debug_picture_batches(ohevc_ctx *c)
{
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->store->spare_m);
    return (int)c->store->batches.size();
}

static int alloc_picture(Picture &p, int width, int height, int cfi, int bd, bool dry = false, PicStore *store = nullptr, hipStream_t stream = nullptr)
{
    const int ps = bd > 8 ? 2 : 1;
    p.w = width; p.h = height; p.cfi = cfi; p.bd = bd;
    size_t off[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 3; i++) {
        const int hs = i ? (cfi == 1 || cfi == 2) : 0, vs = i ? (cfi == 1) : 0;
        const int w = width >> hs, h = height >> vs;
        const int stride = (w * ps + 255) & ~255;          // 256-byte pitch: whole 128-byte lines per row segment
        p.planes[i] = ohevc_plane{ nullptr, stride, w, h };
        off[i + 1] = off[i] + (size_t)stride * h;
    }
    unsigned char *d = reinterpret_cast<unsigned char *>((uintptr_t)0x1000000);      // never dereferenced in record-only mode
    unsigned char *piece = !dry && store ? take_piece(*store, off[3], stream) : nullptr;
    if (piece) {
        d = piece;
    } else if (!dry) {
        // one allocation, the planes back to back: the deblocked copy SAO reads (and the filter-lag snapshot) is one device copy, not three
        void *m = nullptr;
        const hipError_t e = hipMalloc(&m, off[3]);
        if (e != hipSuccess) {
            set_error("picture allocation failed: %s", hipGetErrorString(e));
            for (auto &pl : p.planes) pl = ohevc_plane{};
            return OHEVC_ERR_HIP;
        }
        d = static_cast<unsigned char *>(m);
    }
    for (int i = 0; i < 3; i++) p.planes[i].data = dry ? reinterpret_cast<void *>((uintptr_t)0x1000000 * (i + 1)) : static_cast<void *>(d + off[i]);
    if (ohevc::config().trace_pin && !dry) fprintf(stderr, "pin: device picture %p + %zu (%s)\n", (void *)d, off[3], piece ? "piece" : "own allocation");
    p.used = true; p.single = !dry;
    p.owned = piece == nullptr;           // a piece belongs to its batch (freed with the store)
    return OHEVC_OK;
}

static void unpin_locked(PicStore &st, size_t i);

extern "C" int ohevc_ctx_create(ohevc_ctx **out, int device)
{
    OHEVC_REQUIRE(out != nullptr, "out");
    return ohevc_ctx_create_shared(out, device, nullptr);
}

// The first use of things costs milliseconds on this runtime - the first host-to-device copy of a process 8.7 ms, the second context's 5.6 ms,
// the first device-to-host copy 7.7 ms, the first kernel launch 3.5 ms (code object load), every page-locked staging buffer 0.5-2 ms
// (profiles/r5f_*: all of it inside the first pictures' frame ends).  A context does them when it is made - the sample hooks make one per
// decoding thread when the decoder is opened (ohhip_backend_attach) - instead of in front of its thread's first picture.  Best effort: a
// failure here shows up again, with its message, where the buffers are needed.
// Pictures with a long chain of dependency levels (an intra picture: ~1000 levels, one 8-wavefront workgroup for 3-5 ms) are issued on a
// stream of their own, created at the highest stream priority.  Why: the runtime spreads the streams of one priority over a pool of 4 hardware
// queues, least-used first; a context makes a kernel stream and an upload stream, so the kernel streams of a frame-threaded decoder all land on
// two of the four queues, a hardware queue runs its packets in order, and whatever shares a queue with such a chain waits for it - an
// all-intra stream ran two pictures at a time on 16 frame threads (345 fps against the reference's 2066 on its SSE tables;
// profiles/r5b_overlap_intra_only_16.jsonl: share of time with n chains running {1: 0.25, 2: 0.75}).  Raising GPU_MAX_HW_QUEUES fixes that
// stream and costs every other one 15-50 % (profiles/r5c_stream_priority_hw_queues_ab.txt).  Streams of another priority come out of another
// pool: long chains get up to four queues of their own and leave the regular ones to the short frames.
// ohevc_debug_set_long_chain_levels: a frame whose recorded dependency levels reach this many goes to the long-chain stream (0: never).
static int g_long_chain_levels = 96;
static int g_long_chain_pools = 2;       // ohevc_debug_set_long_chain_pools (1: the highest priority only; 3: a hardware queue of its own per context, below; 4: the three priorities in turn)
extern "C" int ohevc_debug_set_long_chain_pools(int n) { g_long_chain_pools = n < 1 ? 1 : n > 4 ? 4 : n; return OHEVC_OK; }
extern "C" int ohevc_debug_set_long_chain_levels(int levels) { g_long_chain_levels = levels < 0 ? 0 : levels; return OHEVC_OK; }
static int select_stream(ohevc_ctx *c, bool long_chain)
{
    if (long_chain && !c->stream_long) {
        if (g_long_chain_pools == 3) {
            // A stream created with a compute-unit mask is given a hardware queue of ITS OWN by the runtime (not one of the four the streams of a
            // priority share): with every unit enabled the mask restricts nothing, but sixteen decoding threads' long chains - one workgroup
            // each, milliseconds long - then run sixteen at a time instead of eight (two priority pools of four queues, the round-5 form).
            hipDeviceProp_t prop;
            OHEVC_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
            const unsigned words = ((unsigned)prop.multiProcessorCount + 31u) / 32u;
            std::vector<uint32_t> mask(words ? words : 1u, 0xffffffffu);
            if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
            OHEVC_HIP_TRY(hipExtStreamCreateWithCUMask(&c->stream_long, (uint32_t)mask.size(), mask.data()));
        } else {
            int least = 0, greatest = 0;
            OHEVC_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            // (contexts alternate between the highest and the lowest priority: two more pools, eight hardware queues for long chains)
            static std::atomic<unsigned> n_long{0};
            const unsigned k = n_long.fetch_add(1);
            // (pools 4: every third context's long chains share the NORMAL priority's queues with the short frames - twelve chains at a time on an
            // all-intra stream, where the regular streams carry next to nothing)
            const int prio = g_long_chain_pools == 4 ? (k % 3u == 0 ? greatest : k % 3u == 1 ? least : (least + greatest) / 2)
                                                     : ((k & 1u) && g_long_chain_pools > 1 ? least : greatest);
            OHEVC_HIP_TRY(hipStreamCreateWithPriority(&c->stream_long, hipStreamNonBlocking, prio));
        }
        store_add_stream(*c->store, c->stream_long);
    }
    hipStream_t want = long_chain ? c->stream_long : c->stream_norm;
    if (want == c->stream) return OHEVC_OK;
    OHEVC_HIP_TRY(hipEventRecord(c->switch_ev, c->stream));
    OHEVC_HIP_TRY(hipStreamWaitEvent(want, c->switch_ev, 0));
    c->stream = want;
    return OHEVC_OK;
}

static const int g_prewarm_kib = ohevc::config().prewarm_kib;      // 0: off; the upload buffers' first size (x 1.5)
static void prewarm(ohevc_ctx *c)
{
    if (g_prewarm_kib <= 0) return;
    for (int lane = 0; lane < 2; lane++)
        if (c->stage[lane].reserve((size_t)g_prewarm_kib << 10) != OHEVC_OK || c->d_jobs[lane].reserve((size_t)g_prewarm_kib << 10) != OHEVC_OK) return;
    memset(c->stage[0].p, 0, 4096);
    if (hipMemcpyAsync(c->d_jobs[0].p, c->stage[0].p, 4096, hipMemcpyHostToDevice, c->up_stream) != hipSuccess) return;
    if (hipStreamSynchronize(c->up_stream) != hipSuccess) return;
    if (ohevc_dev_copy(static_cast<unsigned char *>(c->d_jobs[1].p), c->d_jobs[0].p, 4096, c->stream) != OHEVC_OK) return;
    if (hipMemcpy2DAsync(c->stage[1].p, 1024, c->d_jobs[1].p, 1024, 1024, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return;
    (void)hipStreamSynchronize(c->stream);
}


extern "C" int ohevc_ctx_create_shared(ohevc_ctx **out, int device, ohevc_ctx *share_with)
{
    OHEVC_REQUIRE(out != nullptr, "out");
    if (g_record_only || (share_with && share_with->dry)) {
        ohevc_ctx *c = new ohevc_ctx();
        c->gen = g_ctx_gen.fetch_add(1);
        c->dry = true;
        c->dry_as_device = share_with ? share_with->dry_as_device : g_record_only == 2;
        c->store = share_with ? share_with->store : std::make_shared<PicStore>();
        *out = c;
        return OHEVC_OK;
    }
    if (share_with) device = share_with->device;
    int rc = ohevc_set_device(device);
    if (rc != OHEVC_OK) return rc;
    ohevc_ctx *c = new ohevc_ctx();
    c->gen = g_ctx_gen.fetch_add(1);
    c->device = device;
    c->store = share_with ? share_with->store : std::make_shared<PicStore>();
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->switch_ev, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->staged[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->staged[1], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->lane_done[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->lane_done[1], hipEventDisableTiming) == hipSuccess;
    for (auto &e : c->ring) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        set_error("stream/event creation failed");
        delete c;
        return OHEVC_ERR_HIP;
    }
    c->stream_norm = c->stream;
    store_add_stream(*c->store, c->stream);
    store_add_stream(*c->store, c->up_stream);
    prewarm(c);
    *out = c;
    return OHEVC_OK;
}

extern "C" const void *ohevc_ctx_store_id(ohevc_ctx *c) { return c ? (const void *)c->store.get() : nullptr; }

extern "C" void ohevc_tables_forget(ohevc_ctx *ctx);      // tables.hip: drop the pointer registry of this ctx

extern "C" void ohevc_ctx_destroy(ohevc_ctx *c)
{
    if (!c) return;
    ohevc_tables_forget(c);
    if (g_trace_timing && c->n_parked)
        fprintf(stderr, "timing: ctx %p parked %ld of its frame ends (issued by the thread that issued their last missing reference)\n", (void *)c, c->n_parked);
    if (g_trace_timing && c->n_frames)
        fprintf(stderr, "timing: ctx %p %d frames: frame_end %.3f ms/frame of which waiting for reference frames %.3f ms; deblocking derived on the device in %d; "
                        "staging copies %.3f, reconstruction calls %.3f, filter calls %.3f (bs %.3f, vertical edges %.3f, horizontal edges %.3f, copy %.3f, SAO %.3f), "
                        "waiting for the staging buffer %.3f, copy-back incl. wait %.3f ms/frame\n",
                (void *)c, c->n_frames, 1e3 * c->t_issue / c->n_frames, 1e3 * c->t_wait_refs / c->n_frames, c->n_map_frames, 1e3 * c->t_part[0] / c->n_frames,
                1e3 * c->t_part[1] / c->n_frames, 1e3 * c->t_part[2] / c->n_frames, 1e3 * c->t_f[0] / c->n_frames, 1e3 * c->t_f[1] / c->n_frames, 1e3 * c->t_f[2] / c->n_frames,
                1e3 * c->t_f[3] / c->n_frames, 1e3 * c->t_f[4] / c->n_frames, 1e3 * c->t_part[3] / c->n_frames, 1e3 * c->t_part[4] / c->n_frames);
    if (c->dry) { delete c; return; }
    // teardown: an error here has nowhere to go
    (void)hipSetDevice(c->device);
    if (!c->is_exec && c->store->issuer) {
        async_drain(*c->store);
        if (c->store.use_count() == 1 + (long)c->store->issuer->execs.size()) issuer_shutdown(*c->store);      // the last recording context goes
    }
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    for (hipStream_t st : { c->stream_norm, c->stream_long }) if (st) (void)hipStreamSynchronize(st);
    if (c->store.use_count() == 1) {            // last context of this store: the pictures go with it
        (void)store_sync(*c->store);
        {
            std::unique_lock<std::shared_mutex> g(c->store->pin_m);
            while (!c->store->pinned.empty()) unpin_locked(*c->store, c->store->pinned.size() - 1);
        }
        for (int i = 0; i < c->store->npics; i++) if (c->store->pics[i].used) free_picture(c->store->pics[i]);
        {
            std::lock_guard<std::mutex> g(c->store->spare_m);
            for (void *b : c->store->batches) (void)hipFree(b);
            c->store->batches.clear(); c->store->spare.clear();
        }
    }
    {   // pictures of the shared store may still name this context's events (the stream has drained: they have all fired)
        std::lock_guard<std::mutex> g(c->store->m);
        for (int i = 0; i < c->store->npics; i++) {
            Picture &p = c->store->pics[i];
            for (hipEvent_t e : c->ring) {
                if (!e) continue;
                if (p.written == e) p.written = nullptr;
                if (p.host_copy == e) p.host_copy = nullptr;
                p.readers.erase(std::remove(p.readers.begin(), p.readers.end(), e), p.readers.end());
            }
        }
    }
    for (auto &e : c->ring) if (e) (void)hipEventDestroy(e);
    // (scratch copies cut out of the store's batches go back to it: the store may outlive this context.  The stream has drained above.)
    if (c->twin.used) free_picture(c->twin, false, c->store.get());
    if (c->lag.used) free_picture(c->lag, false, c->store.get());
    for (DevBuf &b : c->d_jobs) if (b.p) (void)hipFree(b.p);
    for (DevBuf &b : c->d_dense) if (b.p) (void)hipFree(b.p);
    if (c->d_table.p) (void)hipFree(c->d_table.p);
    if (c->d_upsample.p) (void)hipFree(c->d_upsample.p);
    if (c->d_bs.p) (void)hipFree(c->d_bs.p);
    if (c->d_grid.p) (void)hipFree(c->d_grid.p);
    for (PinnedBuf &b : c->stage) if (b.p) (void)hipHostFree(b.p);
    if (c->table_stage.p) (void)hipHostFree(c->table_stage.p);
    for (hipEvent_t e : c->staged) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->lane_done) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : { c->stream_norm, c->stream_long, c->up_stream }) if (st) store_remove_stream(*c->store, st);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    for (auto &e : c->dl_ring) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : { c->stream_norm, c->stream_long }) if (st) { ohevc_mc_forget_stream(st); (void)hipStreamDestroy(st); }
    if (c->switch_ev) (void)hipEventDestroy(c->switch_ev);
    delete c;
}

extern "C" int ohevc_ctx_set_concurrent(ohevc_ctx *c, int on)
{
    OHEVC_REQUIRE(c != nullptr, "null context");
    c->concurrent = on != 0;
    return OHEVC_OK;
}

extern "C" int ohevc_debug_set_level_launch(int mode) { return g_level_launch.exchange(mode, std::memory_order_relaxed); }
// per-context choices (a decoder instance sets them on the contexts it makes; the process-wide debug setters only supply the defaults)
extern "C" int ohevc_ctx_set_option(ohevc_ctx *c, int option, int value)
{
    OHEVC_REQUIRE(c != nullptr && (option == OHEVC_OPT_LEVEL_LAUNCH || option == OHEVC_OPT_FILTERS_ON_DEVICE || option == OHEVC_OPT_PARK_FRAMES), "unknown option");
    OHEVC_REQUIRE(option != OHEVC_OPT_LEVEL_LAUNCH || value <= 3, "level-launch mode 0..3");
    c->opt[option] = value < 0 ? -1 : value;
    return OHEVC_OK;
}
extern "C" int ohevc_ctx_get_option(const ohevc_ctx *c, int option)
{
    return c && (option == OHEVC_OPT_LEVEL_LAUNCH || option == OHEVC_OPT_FILTERS_ON_DEVICE || option == OHEVC_OPT_PARK_FRAMES) ? c->opt[option] : -1;
}
extern "C" int ohevc_debug_set_intra_chain(int on) { const int prev = g_intra_chain; g_intra_chain = on != 0; return prev; }
extern "C" int ohevc_debug_set_intra_pack(int on) { const int prev = g_intra_pack; g_intra_pack = on != 0; return prev; }
extern "C" int ohevc_debug_set_fuse_intra(int on) { const int prev = g_fuse_intra; g_fuse_intra = on != 0; return prev; }
extern "C" int ohevc_debug_set_record_only(int on) { const int prev = g_record_only; g_record_only = on < 0 ? 0 : on > 2 ? 1 : on; return prev; }

// ---- inspection of record-only contexts (ohevc_debug.h): host-logic tests without a GPU
static ohevc_debug_sink g_sink = nullptr;
static void *g_sink_user = nullptr;
extern "C" void ohevc_debug_set_frame_sink(ohevc_debug_sink fn, void *user) { g_sink = fn; g_sink_user = user; }

// The handle never changes over a context's life: stream_norm.  A picture with a long dependency chain is ISSUED on stream_long (select_stream),
// and its frame end joins stream_norm again (frame_end_impl), so that whatever a caller enqueues on - or waits for through - this handle after
// ohevc_frame_end / ohevc_frame_end_async's issue is ordered behind the picture whichever stream carried it.
extern "C" void *ohevc_ctx_stream(ohevc_ctx *c) { return c ? (void *)c->stream_norm : nullptr; }

extern "C" int ohevc_ctx_sync(ohevc_ctx *c)
{
    OHEVC_REQUIRE(c != nullptr, "ctx");
    if (c->dry) return OHEVC_OK;
    if (c->store->issuer && !c->is_exec) {              // frame ends this context submitted run on the issuer's streams
        async_drain(*c->store);
        OHEVC_HIP_TRY(hipSetDevice(c->device));
        OHEVC_HIP_TRY(store_sync(*c->store));
    }
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    c->staged_pending[0] = c->staged_pending[1] = false;
    return OHEVC_OK;
}

extern "C" int ohevc_pic_alloc(ohevc_ctx *c, int width, int height, int cfi, int bd)
{
    OHEVC_REQUIRE(c != nullptr, "ctx");
    OHEVC_REQUIRE(width > 0 && height > 0 && width <= 65535 && height <= 65535, "picture size");
    OHEVC_REQUIRE(cfi >= 1 && cfi <= 3, "chroma_format_idc must be 1..3");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bd), "bit_depth must be 8..12 or 14");
    if (!c->dry) OHEVC_HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->store->m);
    int slot = -1;
    for (int i = 0; i < c->store->npics; i++) if (!c->store->pics[i].used) { slot = i; break; }
    if (slot < 0) { OHEVC_REQUIRE(c->store->npics < kMaxPics, "too many pictures"); slot = c->store->npics++; }
    Picture &np = c->store->pics[slot];
    np = Picture();
    int rc = alloc_picture(np, width, height, cfi, bd, c->dry, c->store.get(), c->stream);
    if (rc != OHEVC_OK) return rc;
    if (!c->dry && np.owned) {            // (a piece of a batch was zeroed with its batch)
        // zeroed like the reference's frame pool (av_buffer_allocz, libavcodec/utils.c): a sample nobody ever wrote -- a stream that
        // predicts from a picture it never sent -- is at least the same sample on every run.  On this context's stream and drained
        // before the slot is handed out: a memset on the null stream would not be ordered against the (non-blocking) streams
        // that reconstruct into the picture.
        for (const ohevc_plane &pl : np.planes) OHEVC_HIP_TRY(hipMemsetAsync(pl.data, 0, (size_t)pl.stride * pl.height, c->stream));
        OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->store->version++;
    return slot;
}

extern "C" int ohevc_pic_adopt(ohevc_ctx *c, const ohevc_plane planes[3], int width, int height, int cfi, int bd)
{
    OHEVC_REQUIRE(c != nullptr && planes != nullptr, "null argument");
    OHEVC_REQUIRE(width > 0 && height > 0 && cfi >= 1 && cfi <= 3 && OHEVC_BIT_DEPTH_OK(bd), "bad picture description");
    std::lock_guard<std::mutex> g(c->store->m);
    int slot = -1;
    for (int i = 0; i < c->store->npics; i++) if (!c->store->pics[i].used) { slot = i; break; }
    if (slot < 0) { OHEVC_REQUIRE(c->store->npics < kMaxPics, "too many pictures"); slot = c->store->npics++; }
    Picture &p = c->store->pics[slot];
    p = Picture();
    p.w = width; p.h = height; p.cfi = cfi; p.bd = bd; p.owned = false; p.single = false; p.used = true;
    for (int i = 0; i < 3; i++) {
        OHEVC_REQUIRE(planes[i].data != nullptr && (planes[i].stride & 15) == 0 && (reinterpret_cast<uintptr_t>(planes[i].data) & 15) == 0,
                      "adopted planes must be 16-byte aligned with a 16-byte multiple stride");
        p.planes[i] = planes[i];
    }
    c->store->version++;
    return slot;
}

static Picture *get_pic(ohevc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= c->store->npics || !c->store->pics[slot].used) return nullptr;
    return &c->store->pics[slot];
}

extern "C" int ohevc_pic_release(ohevc_ctx *c, int slot)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr, "bad picture slot");
    // other contexts of the store may still have kernels in flight that read this picture
    if (!c->dry && !c->is_exec) async_drain(*c->store);
    if (!c->dry) OHEVC_HIP_TRY(c->store.use_count() > 1 ? store_sync(*c->store) : hipStreamSynchronize(c->stream));
    if (c->cur == slot) c->cur = -1;
    std::lock_guard<std::mutex> g(c->store->m);
    c->store->version++;
    p->readers.clear();
    return free_picture(*p, c->dry, c->store.get());
}

extern "C" int ohevc_pic_upload(ohevc_ctx *c, int slot, int plane, const void *host, ptrdiff_t host_stride)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && plane >= 0 && plane < 3 && host != nullptr, "bad argument");
    if (c->dry) return OHEVC_OK;
    async_drain(*c->store);              // queued frame ends may still have to read what lives in this slot
    {   // frames of other contexts may still read (or write) the picture that lived in this slot's memory
        std::lock_guard<std::mutex> g(c->store->m);
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
        for (hipEvent_t e : p->readers) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
        p->readers.clear();
        p->written = nullptr;
        p->failed = false;
    }
    const ohevc_plane &pl = p->planes[plane];
    OHEVC_HIP_TRY(hipMemcpy2DAsync(pl.data, pl.stride, host, host_stride, (size_t)pl.width * (p->bd > 8 ? 2 : 1), pl.height,
                                   hipMemcpyHostToDevice, c->stream));
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));      // pageable source: do not return before it has been read
    return OHEVC_OK;
}

// ---- page-locked application memory.  The decoder's frame buffers (alloc_frame, hevc_refs.c:75-114) are pageable: a copy-back into them
// goes through the runtime's staging buffers and a CPU copy - 99.5 MB per 8K Main10 picture.  Registered, the same copy is one DMA at
// the bus rate.  The application names the ALLOCATIONS (for the decoder: AVFrame.buf[i]->data / ->size, the buffers its pool recycles),
// so ranges of live buffers never overlap; a range overlapping an earlier, different registration means that memory was freed and
// allocated again, and replaces it.  Failure to register is not an error of the decoder: the copies stay pageable.
static void unpin_locked(PicStore &st, size_t i)
{
    const hipError_t e = hipHostUnregister(reinterpret_cast<void *>(st.pinned[i].first));
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: unpin %p + %zu: %s\n", (void *)st.pinned[i].first, st.pinned[i].second, hipGetErrorString(e));
    if (e != hipSuccess) (void)hipGetLastError();
    st.pinned[i] = st.pinned.back();
    st.pinned.pop_back();
}

extern "C" int ohevc_host_pin(ohevc_ctx *c, void *ptr, size_t bytes)
{
    OHEVC_REQUIRE(c != nullptr && ptr != nullptr && bytes > 0, "bad argument");
    if (c->dry) { if (ohevc::config().trace_pin) fprintf(stderr, "pin: (record-only) pin %p + %zu\n", ptr, bytes); return OHEVC_OK; }
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    {   // the common case - a buffer of the decoder's pool seen again - takes the shared lock only
        std::shared_lock<std::shared_mutex> g(c->store->pin_m);
        for (const auto &r : c->store->pinned) if (r.first == a && r.second == bytes) return OHEVC_OK;
    }
    std::unique_lock<std::shared_mutex> g(c->store->pin_m);
    auto &v = c->store->pinned;
    for (size_t i = 0; i < v.size(); i++) if (v[i].first == a && v[i].second == bytes) return OHEVC_OK;
    bool drained = false;
    for (size_t i = 0; i < v.size();) {
        if (v[i].first < a + bytes && a < v[i].first + v[i].second) {
            if (!drained) { OHEVC_HIP_TRY(hipSetDevice(c->device)); (void)store_sync(*c->store); drained = true; }   // a copy into the old range may be in flight
            unpin_locked(*c->store, i);
        } else {
            i++;
        }
    }
    OHEVC_HIP_TRY(hipSetDevice(c->device));
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostRegister(%p, %zu) failed: %s (copies into it stay pageable)", ptr, bytes, hipGetErrorString(e));
        return OHEVC_ERR_HIP;
    }
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: pin %p + %zu (%zu ranges)\n", ptr, bytes, v.size() + 1);
    v.emplace_back(a, bytes);
    return OHEVC_OK;
}

extern "C" int ohevc_host_unpin_all(ohevc_ctx *c)
{
    OHEVC_REQUIRE(c != nullptr, "null context");
    if (c->dry) return OHEVC_OK;
    async_drain(*c->store);                             // queued copy-backs name this memory
    std::unique_lock<std::shared_mutex> g(c->store->pin_m);
    if (c->store->pinned.empty()) return OHEVC_OK;
    OHEVC_HIP_TRY(hipSetDevice(c->device));
    (void)store_sync(*c->store);
    while (!c->store->pinned.empty()) unpin_locked(*c->store, c->store->pinned.size() - 1);
    return OHEVC_OK;
}

// Drop the page locks of ONE allocation (every registered range that overlaps [ptr, ptr + bytes)): the decoder gave the memory back to the
// allocator.  The caller knows no copy into THAT range is pending (the decoder recycles a buffer only after the application let go of the
// picture); copies into other ranges go on undisturbed - they hold the shared lock, and nothing but this range is touched.
extern "C" int ohevc_host_unpin(ohevc_ctx *c, void *ptr, size_t bytes)
{
    OHEVC_REQUIRE(c != nullptr && ptr != nullptr && bytes > 0, "bad argument");
    if (c->dry) return OHEVC_OK;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    // Copy-backs issued by the library's issuer threads (ohevc_frame_end_async) hold no lock while they are queued or in flight: with an
    // issuer, first let it issue what it holds and wait for the device - a page lock must not go while a DMA may still target the range.
    if (c->store->issuer) {
        async_drain(*c->store);
        OHEVC_HIP_TRY(hipSetDevice(c->device));
        (void)store_sync(*c->store);
    }
    std::unique_lock<std::shared_mutex> g(c->store->pin_m);        // (waits for the synchronous copy-backs in flight: they hold the shared lock)
    auto &v = c->store->pinned;
    for (size_t i = 0; i < v.size();) {
        if (v[i].first < a + bytes && a < v[i].first + v[i].second) unpin_locked(*c->store, i);
        else i++;
    }
    return OHEVC_OK;
}

// Page-locked memory of the library's own (ohevc_ctx.h).  A 64-byte header in front of the block says how it was made: a record-only context
// (no device) hands out plain memory, and ohevc_host_free has no context to ask.
namespace { struct HostBlockHeader { uint64_t magic; uint32_t pinned; uint32_t pad; void *base; char fill[40]; }; static_assert(sizeof(HostBlockHeader) == 64, "header"); }
static constexpr uint64_t kHostBlockMagic = 0x6f6865766368626bull;
extern "C" int ohevc_host_alloc(ohevc_ctx *c, size_t bytes, void **out)
{
    OHEVC_REQUIRE(c != nullptr && out != nullptr && bytes > 0, "bad argument");
    *out = nullptr;
    void *base = nullptr;
    const bool pinned = !c->dry;
    if (pinned) {
        OHEVC_HIP_TRY(hipSetDevice(c->device));
        const hipError_t e = hipHostMalloc(&base, bytes + sizeof(HostBlockHeader), hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return OHEVC_ERR_HIP; }
    } else if (posix_memalign(&base, 64, bytes + sizeof(HostBlockHeader)) != 0) {
        set_error("out of memory (%zu bytes of host memory)", bytes);
        return OHEVC_ERR_ARG;
    }
    HostBlockHeader *h = static_cast<HostBlockHeader *>(base);
    h->magic = kHostBlockMagic; h->pinned = pinned; h->pad = 0; h->base = base;
    *out = h + 1;
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: host block %p + %zu (%s)\n", *out, bytes, pinned ? "page-locked" : "plain");
    return OHEVC_OK;
}
extern "C" int ohevc_host_alloc_pins(const ohevc_ctx *c) { return c && !c->dry; }
extern "C" int ohevc_host_block_pinned(const void *ptr)      // 1 page-locked, 0 plain memory (a record-only context made it), -1 not a block
{
    if (!ptr) return -1;
    const HostBlockHeader *h = static_cast<const HostBlockHeader *>(ptr) - 1;
    return h->magic == kHostBlockMagic && h->base == h ? (int)h->pinned : -1;
}
extern "C" int ohevc_host_free(void *ptr)
{
    if (!ptr) return OHEVC_OK;
    HostBlockHeader *h = static_cast<HostBlockHeader *>(ptr) - 1;
    OHEVC_REQUIRE(h->magic == kHostBlockMagic && h->base == h, "not a block of ohevc_host_alloc");
    h->magic = 0;
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: host block %p freed\n", ptr);
    if (h->pinned) { const hipError_t e = hipHostFree(h); if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostFree failed: %s", hipGetErrorString(e)); return OHEVC_ERR_HIP; } }
    else free(h);
    return OHEVC_OK;
}

// the three planes of a picture with ONE wait at the end (ohevc_pic_download waits per plane)
// Wait (lk = the store's mutex, held) until the frame end of picture p has been ISSUED.  With parked frames in the store (ohevc_frame_end_deferred)
// the waiting thread helps: it issues whatever parked frame has become ready - the picture it waits for may be one of them, or hang behind one.
static bool wait_end_issued(ohevc_ctx *c, Picture &p, std::unique_lock<std::mutex> &lk)
{
    if (p.end_issued) return true;
    PicStore &st = *c->store;
    const double deadline = now_s() + g_ref_wait_s;
    while (!p.end_issued) {
        if (get_issuer(st) && !c->is_exec) {
            lk.unlock();
            issuer_help(st);
            lk.lock();
            if (p.end_issued) break;
            st.cv.wait_for(lk, std::chrono::milliseconds(1));
        } else {
            st.cv.wait_for(lk, std::chrono::milliseconds(50));
        }
        if (!p.end_issued && now_s() > deadline) return false;
    }
    return true;
}

extern "C" int ohevc_pic_download_planes(ohevc_ctx *c, int slot, void *const host[3], const ptrdiff_t host_stride[3])
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && host != nullptr && host_stride != nullptr, "bad argument");
    if (c->dry) return OHEVC_OK;
    {
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!wait_end_issued(c, *p, lk)) {
            set_error("picture %d was never completed by its decoding thread", slot);
            return OHEVC_ERR_STATE;
        }
        if (p->failed) { set_error("picture %d: its frame failed", slot); return OHEVC_ERR_STATE; }
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
    }
    const double t0 = g_trace_timing ? now_s() : 0;
    std::shared_lock<std::shared_mutex> pins(c->store->pin_m);     // no page lock is dropped between the issue of these copies and their completion
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: copy-back of slot %d (ctx %p) -> %p %p %p strides %td %td %td\n", slot, (void *)c, host[0], host[1], host[2], host_stride[0], host_stride[1], host_stride[2]);
    for (int i = 0; i < 3; i++) {
        if (!host[i]) continue;
        const ohevc_plane &pl = p->planes[i];
        OHEVC_HIP_TRY(hipMemcpy2DAsync(host[i], host_stride[i], pl.data, pl.stride, (size_t)pl.width * (p->bd > 8 ? 2 : 1), pl.height,
                                       hipMemcpyDeviceToHost, c->stream));
    }
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: copy-back of slot %d landed\n", slot);
    if (g_trace_timing) c->t_part[4] += now_s() - t0;          // (the wait covers the picture's device work as well: nothing waited for it before)
    return OHEVC_OK;
}

// The copy-back QUEUED behind the picture's device work on this context's stream, not waited for (ohevc_ctx.h): ohevc_pic_wait_host returns when
// it has landed.  The event of the copy also counts as a reader of the picture: the next picture begun in the slot is ordered behind it.
extern "C" int ohevc_pic_download_queue(ohevc_ctx *c, int slot, void *const host[3], const ptrdiff_t host_stride[3])
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && host != nullptr && host_stride != nullptr, "bad argument");
    if (c->dry) return OHEVC_OK;
    {
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!wait_end_issued(c, *p, lk)) {
            set_error("picture %d was never completed by its decoding thread", slot);
            return OHEVC_ERR_STATE;
        }
        if (p->failed) { set_error("picture %d: its frame failed", slot); return OHEVC_ERR_STATE; }
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
    }
    if (ohevc::config().trace_pin) fprintf(stderr, "pin: copy-back of slot %d queued (ctx %p) -> %p %p %p\n", slot, (void *)c, host[0], host[1], host[2]);
    for (int i = 0; i < 3; i++) {
        if (!host[i]) continue;
        const ohevc_plane &pl = p->planes[i];
        OHEVC_HIP_TRY(hipMemcpy2DAsync(host[i], host_stride[i], pl.data, pl.stride, (size_t)pl.width * (p->bd > 8 ? 2 : 1), pl.height,
                                       hipMemcpyDeviceToHost, c->stream));
    }
    hipEvent_t ev = c->ring[c->ring_next];
    c->ring_next = (c->ring_next + 1) % 16;
    OHEVC_HIP_TRY(hipEventRecord(ev, c->stream));
    {
        std::lock_guard<std::mutex> g(c->store->m);
        p->host_copy = ev;
        p->host_copy_issued = true;
        if (std::find(p->readers.begin(), p->readers.end(), ev) == p->readers.end()) p->readers.push_back(ev);
    }
    return OHEVC_OK;
}

extern "C" int ohevc_pic_download(ohevc_ctx *c, int slot, int plane, void *host, ptrdiff_t host_stride)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && plane >= 0 && plane < 3 && host != nullptr, "bad argument");
    if (c->dry) return OHEVC_OK;
    {   // the picture may be reconstructed by another context of the store (another decoding thread), possibly not even issued yet
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!wait_end_issued(c, *p, lk)) {
            set_error("picture %d was never completed by its decoding thread", slot);
            return OHEVC_ERR_STATE;
        }
        if (p->failed) { set_error("picture %d: its frame failed", slot); return OHEVC_ERR_STATE; }
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
    }
    const ohevc_plane &pl = p->planes[plane];
    std::shared_lock<std::shared_mutex> pins(c->store->pin_m);
    OHEVC_HIP_TRY(hipMemcpy2DAsync(host, host_stride, pl.data, pl.stride, (size_t)pl.width * (p->bd > 8 ? 2 : 1), pl.height,
                                   hipMemcpyDeviceToHost, c->stream));
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}

// Frame-parallel decoding over several GPUs (one process each): a finished picture leaves its owner through ohevc_pic_export and
// enters every other process's picture store through ohevc_pic_import; what carries the bytes in between (an RCCL broadcast over
// xGMI, openhevc_amd/dist.py) is the application's.  Both work on DEVICE buffers holding the plane exactly as the store lays it out
// (stride x height bytes, ohevc_pic_planes), take part in the store's cross-context ordering like upload / download do, and return
// when the copy is done: the buffer can go straight into a collective / be reused.
extern "C" int ohevc_pic_export(ohevc_ctx *c, int slot, int plane, void *device_dst, size_t bytes)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && plane >= 0 && plane < 3 && device_dst != nullptr, "bad argument");
    const ohevc_plane &pl = p->planes[plane];
    OHEVC_REQUIRE(bytes == (size_t)pl.stride * pl.height, "size must be stride x height of the plane (ohevc_pic_planes)");
    if (c->dry) return OHEVC_OK;
    {
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!wait_end_issued(c, *p, lk)) {
            set_error("picture %d was never completed by its decoding thread", slot);
            return OHEVC_ERR_STATE;
        }
        if (p->failed) { set_error("picture %d: its frame failed", slot); return OHEVC_ERR_STATE; }
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
    }
    OHEVC_HIP_TRY(hipMemcpyAsync(device_dst, pl.data, bytes, hipMemcpyDeviceToDevice, c->stream));
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}

extern "C" int ohevc_pic_import(ohevc_ctx *c, int slot, int plane, const void *device_src, size_t bytes)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && plane >= 0 && plane < 3 && device_src != nullptr, "bad argument");
    const ohevc_plane &pl = p->planes[plane];
    OHEVC_REQUIRE(bytes == (size_t)pl.stride * pl.height, "size must be stride x height of the plane (ohevc_pic_planes)");
    if (c->dry) return OHEVC_OK;
    {   // like ohevc_pic_upload: frames of other contexts may still read (or write) what lived in this slot's memory
        std::lock_guard<std::mutex> g(c->store->m);
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
        for (hipEvent_t e : p->readers) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
        p->readers.clear();
        p->written = nullptr;
        p->failed = false;
    }
    OHEVC_HIP_TRY(hipMemcpyAsync(pl.data, device_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}

// Row ranges of the two calls above (band-chunked exchange, include/ohevc_frames.h): rows [row0, row0 + rows) of the plane, the buffer
// laid out like the whole plane (the band sits at row0 * stride).  The first export of a picture is the one that waits for its device
// work; the first import of a picture (first != 0) is the one that orders the slot's memory against its earlier users.
static int export_rows_impl(ohevc_ctx *c, int slot, int plane, int row0, int rows, void *device_plane_base, bool wait)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && plane >= 0 && plane < 3 && device_plane_base != nullptr, "bad argument");
    const ohevc_plane &pl = p->planes[plane];
    OHEVC_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= pl.height, "row range outside the plane");
    if (c->dry || rows == 0) return OHEVC_OK;
    {
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!wait_end_issued(c, *p, lk)) {
            set_error("picture %d was never completed by its decoding thread", slot);
            return OHEVC_ERR_STATE;
        }
        if (p->failed) { set_error("picture %d: its frame failed", slot); return OHEVC_ERR_STATE; }
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
    }
    const size_t off = (size_t)row0 * pl.stride;
    OHEVC_HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(device_plane_base) + off, static_cast<const unsigned char *>(pl.data) + off, (size_t)rows * pl.stride,
                                 hipMemcpyDeviceToDevice, c->stream));
    if (wait) OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}
extern "C" int ohevc_pic_export_rows(ohevc_ctx *c, int slot, int plane, int row0, int rows, void *device_plane_base)
{
    return export_rows_impl(c, slot, plane, row0, rows, device_plane_base, true);
}
// one band = the three planes' row ranges, ONE wait for the host (the per-plane calls cost a decoding thread up to 24 stalls per exchanged picture)
extern "C" int ohevc_pic_export_band(ohevc_ctx *c, int slot, const int row0[3], const int rows[3], void *const device_plane_base[3])
{
    OHEVC_REQUIRE(c != nullptr && row0 && rows && device_plane_base, "bad argument");
    bool any = false;
    for (int pl = 0; pl < 3; pl++) {
        if (!device_plane_base[pl] || rows[pl] <= 0) continue;
        const int rc = export_rows_impl(c, slot, pl, row0[pl], rows[pl], device_plane_base[pl], false);
        if (rc != OHEVC_OK) return rc;
        any = true;
    }
    if (any && !c->dry) OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}

static int import_rows_impl(ohevc_ctx *c, int slot, int plane, int row0, int rows, const void *device_plane_base, int first, bool wait)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && plane >= 0 && plane < 3 && device_plane_base != nullptr, "bad argument");
    const ohevc_plane &pl = p->planes[plane];
    OHEVC_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= pl.height, "row range outside the plane");
    if (c->dry) return OHEVC_OK;
    if (first) {
        std::lock_guard<std::mutex> g(c->store->m);
        if (p->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, p->written, 0));
        for (hipEvent_t e : p->readers) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
        p->readers.clear();
        p->written = nullptr;
        p->failed = false;
    }
    if (rows == 0) return OHEVC_OK;
    const size_t off = (size_t)row0 * pl.stride;
    OHEVC_HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(pl.data) + off, static_cast<const unsigned char *>(device_plane_base) + off, (size_t)rows * pl.stride,
                                 hipMemcpyDeviceToDevice, c->stream));
    if (wait) OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}
extern "C" int ohevc_pic_import_rows(ohevc_ctx *c, int slot, int plane, int row0, int rows, const void *device_plane_base, int first)
{
    return import_rows_impl(c, slot, plane, row0, rows, device_plane_base, first, true);
}
extern "C" int ohevc_pic_import_band(ohevc_ctx *c, int slot, const int row0[3], const int rows[3], const void *const device_plane_base[3], int first)
{
    OHEVC_REQUIRE(c != nullptr && row0 && rows && device_plane_base, "bad argument");
    bool any = false;
    for (int pl = 0; pl < 3; pl++) {
        if (!device_plane_base[pl]) continue;
        const int rc = import_rows_impl(c, slot, pl, row0[pl], rows[pl] < 0 ? 0 : rows[pl], device_plane_base[pl], first, false);
        if (rc != OHEVC_OK) return rc;
        any = any || rows[pl] > 0;
    }
    if (any && !c->dry) OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}

// The deepest luma row of reference picture `slot` that the motion compensation recorded for the open frame reads (-1: none of it).
extern "C" int ohevc_frame_ref_reach(ohevc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot > OHEVC_MAX_PICTURES) return -1;
    int reach = c->reach[slot];
    if (!c->side.empty()) {
        std::lock_guard<std::mutex> g(c->side_m);
        for (auto &sd : c->side) reach = std::max(reach, (int)sd.second->reach[slot]);
    }
    return reach;
}

extern "C" int ohevc_pic_planes(ohevc_ctx *c, int slot, ohevc_plane out[3])
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr && out != nullptr, "bad argument");
    for (int i = 0; i < 3; i++) out[i] = p->planes[i];
    return OHEVC_OK;
}

extern "C" int ohevc_pic_info(ohevc_ctx *c, int slot, int *width, int *height, int *cfi, int *bd)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr, "bad picture slot");
    if (width) *width = p->w;
    if (height) *height = p->h;
    if (cfi) *cfi = p->cfi;
    if (bd) *bd = p->bd;
    return OHEVC_OK;
}

// SHVC: resample picture src_slot (base layer) into picture dst_slot (the enhancement layer's inter-layer reference picture)
// -- hevc_frame_start / ff_upsample_block, hevc.c:3240-3242, hevc_filter.c:1370-1395.  Ordered like a tiny frame of its own: waits
// for whoever reconstructs src and for earlier users of dst's memory, publishes dst when done.
extern "C" int ohevc_pic_upsample(ohevc_ctx *c, int dst_slot, int src_slot, const ohevc_upsample_params *prm)
{
    Picture *d = get_pic(c, dst_slot), *sp = get_pic(c, src_slot);
    OHEVC_REQUIRE(d != nullptr && sp != nullptr && prm != nullptr && dst_slot != src_slot, "bad picture slots");
    OHEVC_REQUIRE(d->cfi == 1 && sp->cfi == 1 && d->bd == sp->bd, "inter-layer up-sampling is defined for 4:2:0 pictures of one bit depth");
    OHEVC_REQUIRE(prm->el_width == d->w && prm->el_height == d->h && prm->bl_width <= sp->w && prm->bl_height <= sp->h, "parameters do not match the pictures");
    if (c->dry) return OHEVC_OK;
    {
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!wait_end_issued(c, *sp, lk)) {
            set_error("base-layer picture %d was never completed by its decoding thread", src_slot);
            return OHEVC_ERR_STATE;
        }
        if (sp->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, sp->written, 0));
        if (d->written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, d->written, 0));
        for (hipEvent_t e : d->readers) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
        d->readers.clear();
        d->end_issued = false;
    }
    // maps of the three planes, one upload per geometry (the parameters of a layer pair do not change inside a stream)
    if (!c->up_valid || memcmp(&c->up_prm, prm, sizeof(*prm)) != 0) {
        std::vector<unsigned char> host;
        c->up_valid = false;
        for (int pl = 0; pl < 3; pl++) {
            const int w = d->planes[pl].width, h = d->planes[pl].height;
            auto put = [&](size_t bytes) { size_t o = (host.size() + 15) & ~(size_t)15; host.resize(o + bytes); return o; };
            c->up_off_cols[pl] = put((size_t)w * sizeof(ohevc_upsample_tap));
            c->up_off_colof[pl] = put((size_t)w * sizeof(int16_t));
            c->up_off_rows[pl] = put((size_t)h * sizeof(ohevc_upsample_tap));
        }
        for (int pl = 0; pl < 3; pl++) {
            int rc = ohevc_upsample_make_maps(prm, pl, reinterpret_cast<ohevc_upsample_tap *>(host.data() + c->up_off_cols[pl]),
                                              reinterpret_cast<int16_t *>(host.data() + c->up_off_colof[pl]),
                                              reinterpret_cast<ohevc_upsample_tap *>(host.data() + c->up_off_rows[pl]), &c->up_src_cols[pl], &c->up_src_rows[pl]);
            if (rc != OHEVC_OK) return rc;
        }
        OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));   // launches of the previous geometry may still read the old maps
        if (host.size() > c->d_upsample.cap) {
            int rc = c->d_upsample.reserve(host.size());
            if (rc != OHEVC_OK) return rc;
        }
        OHEVC_HIP_TRY(hipMemcpyAsync(c->d_upsample.p, host.data(), host.size(), hipMemcpyHostToDevice, c->stream));
        OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));   // `host` (pageable) must outlive the copy; once per geometry
        c->up_prm = *prm;
        c->up_valid = true;
    }
    unsigned char *base = static_cast<unsigned char *>(c->d_upsample.p);
    {
        const ohevc_upsample_tap *cols[3], *rows[3];
        const int16_t *col_of[3];
        for (int pl = 0; pl < 3; pl++) {
            cols[pl] = reinterpret_cast<const ohevc_upsample_tap *>(base + c->up_off_cols[pl]);
            col_of[pl] = reinterpret_cast<const int16_t *>(base + c->up_off_colof[pl]);
            rows[pl] = reinterpret_cast<const ohevc_upsample_tap *>(base + c->up_off_rows[pl]);
        }
        int rc = ohevc_dev_upsample_picture(d->planes, sp->planes, d->bd, cols, col_of, rows, c->up_src_cols, c->up_src_rows, c->stream);      // one launch
        if (rc != OHEVC_OK) return rc;
    }
    hipEvent_t ev = c->ring[c->ring_next];
    c->ring_next = (c->ring_next + 1) % 16;
    OHEVC_HIP_TRY(hipEventRecord(ev, c->stream));         // (no host wait: the parsing thread goes on, the picture's own launches queue behind)
    {
        std::lock_guard<std::mutex> g(c->store->m);
        d->written = ev;
        if (std::find(sp->readers.begin(), sp->readers.end(), ev) == sp->readers.end()) sp->readers.push_back(ev);
        d->end_issued = true;
    }
    c->store->cv.notify_all();
    return OHEVC_OK;
}

// the recorder the calling thread writes to
static inline Rec &pick(ohevc_ctx *c)
{
    if (!c->concurrent) return *c;
    struct Cache { uint64_t gen = 0, epoch = 0; Rec *r = nullptr; };
    static thread_local Cache cache;
    const uint64_t epoch = c->epoch.load(std::memory_order_acquire);
    if (cache.gen == c->gen && cache.epoch == epoch) return *cache.r;
    Rec *r = c;
    if (std::this_thread::get_id() != c->owner) {
        std::lock_guard<std::mutex> g(c->side_m);
        r = nullptr;
        for (auto &sd : c->side) if (sd.first == std::this_thread::get_id()) r = sd.second.get();
        if (!r) { c->side.emplace_back(std::this_thread::get_id(), std::unique_ptr<Rec>(new Rec())); r = c->side.back().second.get(); }
    }
    cache.gen = c->gen; cache.epoch = epoch; cache.r = r;
    return *r;
}

static inline LevelBins &level_bins(Rec &r, int level)
{
    if (level >= (int)r.levels.size()) r.levels.resize((size_t)level + 16);
    if (level > r.max_level) r.max_level = level;
    return r.levels[level];
}

static void clear_rec(Rec &r)
{
    r.mc.clear(); r.mc_small.clear(); r.coeffs.clear(); r.cips.clear(); r.expand.clear(); r.dense = 0;
    r.ctb_intra.clear(); r.ctb_tu.clear(); r.ctb_ops.clear();
    for (int l = 0; l <= r.max_level; l++) {
        LevelBins &lb = r.levels[l];
        for (uint64_t m = lb.touched; m; m &= m - 1) { const int b = __builtin_ctzll(m); lb.tu[b >> 4][b & 15].clear(); }
        lb.touched = 0;
        lb.intra.clear();
        lb.intra_res.clear();
    }
    r.max_level = -1;
    r.last_intra.level = -1;
    for (int16_t &v : r.reach) v = -1;
}

// Fold what the other threads recorded into the context's own recorder (called by the thread that runs the frame, after
// the workers are done: the reference joins its slice threads before the frame can end).  Jobs keep their dependency levels;
// arena offsets and constrained-intra side-record indices are rebased.
static void merge_side(ohevc_ctx *c)
{
    if (c->side.empty()) return;
    std::lock_guard<std::mutex> g(c->side_m);
    for (auto &sd : c->side) {
        Rec &r = *sd.second;
        c->mc.insert(c->mc.end(), r.mc.begin(), r.mc.end());
        c->mc_small.insert(c->mc_small.end(), r.mc_small.begin(), r.mc_small.end());
        const uint32_t cbase = c->dense, sbase = (uint32_t)c->coeffs.size(), ibase = (uint32_t)c->cips.size();     // dense-arena base of this recorder's blocks; base of its compact stream
        c->coeffs.insert(c->coeffs.end(), r.coeffs.begin(), r.coeffs.end());
        for (ohevc_expand_rec e : r.expand) { e.src += sbase; e.dst += cbase; c->expand.push_back(e); }
        c->dense += r.dense;
        c->cips.insert(c->cips.end(), r.cips.begin(), r.cips.end());
        {   // CTB-ordered intra work: a CTB is decoded by one thread, so its operations stay contiguous and in order
            const uint32_t jbase = (uint32_t)c->ctb_intra.size(), tbase = (uint32_t)c->ctb_tu.size();
            for (ohevc_intra_job j : r.ctb_intra) {
                if (j.flags2 & OHEVC_INTRA2_CIP) j.cip_index += ibase;
                c->ctb_intra.push_back(j);
            }
            c->ctb_tu.insert(c->ctb_tu.end(), r.ctb_tu.begin(), r.ctb_tu.end());      // arena offsets are rebased through the op words below
            for (auto op : r.ctb_ops) {
                uint32_t w = op.second;
                if (w >> 31) {
                    const int kind = (int)((w >> 25) & 15u);
                    ohevc_tu_job &j = c->ctb_tu[tbase + (w & 0x1ffffffu)];
                    if (kind != OHEVC_TU_DC) j.coeff_off += cbase;
                    if (kind == OHEVC_TU_CROSS) j.reserved1 += cbase;
                    w += tbase;
                } else {
                    w += jbase;
                }
                c->ctb_ops.emplace_back(op.first, w);
            }
        }
        for (int l = 0; l <= r.max_level; l++) {
            LevelBins &src = r.levels[l];
            if (!src.touched && src.intra.empty()) continue;
            LevelBins &dst = level_bins(*c, l);
            for (ohevc_intra_job j : src.intra) {
                if (j.flags2 & OHEVC_INTRA2_CIP) j.cip_index += ibase;
                dst.intra.push_back(j);
            }
            for (ohevc_tu_job j : src.intra_res) {            // (empty when residuals are not paired: parallel to intra otherwise)
                if (j.reserved0 && j.reserved0 - 1 != OHEVC_TU_DC) j.coeff_off += cbase;
                dst.intra_res.push_back(j);
            }
            for (uint64_t m = src.touched; m; m &= m - 1) {
                const int b = __builtin_ctzll(m), kind = b & 15;
                auto &dv = dst.tu[b >> 4][kind];
                for (ohevc_tu_job j : src.tu[b >> 4][kind]) {
                    if (kind != OHEVC_TU_DC) j.coeff_off += cbase;
                    if (kind == OHEVC_TU_CROSS) j.reserved1 += cbase;
                    dv.push_back(j);
                }
                dst.touched |= 1ull << b;
            }
        }
        c->dbk_v.insert(c->dbk_v.end(), r.dbk_v.begin(), r.dbk_v.end());
        c->dbk_h.insert(c->dbk_h.end(), r.dbk_h.begin(), r.dbk_h.end());
        c->bs_calls.insert(c->bs_calls.end(), r.bs_calls.begin(), r.bs_calls.end());
        r.bs_calls.clear();
        c->sao.insert(c->sao.end(), r.sao.begin(), r.sao.end());
        c->sao_lagged |= r.sao_lagged;
        for (int k = 0; k < 5; k++) { c->nstat[k] += r.nstat[k]; r.nstat[k] = 0; }
        c->alg += r.alg; r.alg = 0;
        clear_rec(r);
        r.dbk_v.clear(); r.dbk_h.clear(); r.sao.clear(); r.sao_lagged = false;
    }
}

static void clear_recorded(ohevc_ctx *c)
{
    clear_rec(*c);
    for (int i = 0; i < 3; i++) std::fill(c->level_map[i].begin(), c->level_map[i].end(), 0);
}

extern "C" int ohevc_frame_begin(ohevc_ctx *c, int slot)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr, "bad picture slot");
    if (!c->dry) settle_slot(c, slot);                  // parked frames that still read / write this slot's memory go first
    c->cur = slot;
    if (g_trace_order) fprintf(stderr, "order: ctx %p begins target %d\n", (void *)c, slot);
    {
        std::lock_guard<std::mutex> g(c->store->m);
        p->end_issued = false;
        p->failed = false;
        c->my_gen = ++p->gen;
    }
    if (!c->dry && c->stream != c->stream_norm && c->stream_norm) {   // the previous picture ran on the long-chain stream
        int rc = select_stream(c, false);
        if (rc != OHEVC_OK) return rc;
    }
    c->ref_slots.clear();
    c->target_guarded = false;
    c->frame_mode = c->opt[OHEVC_OPT_LEVEL_LAUNCH] >= 0 ? c->opt[OHEVC_OPT_LEVEL_LAUNCH] : (int)g_level_launch;
    c->flushed_intra = 0; c->flush_closed = false;
    c->log2_ctb = 0;
    for (int i = 0; i < 3; i++) {
        c->lm_w[i] = (p->planes[i].width + 3) >> 2;
        c->lm_h[i] = (p->planes[i].height + 3) >> 2;
        c->level_map[i].assign((size_t)c->lm_w[i] * c->lm_h[i], 0);
    }
    clear_recorded(c);
    c->dbk_v.clear(); c->dbk_h.clear(); c->dbk_blob.clear(); c->sao.clear(); c->bypass.clear();
    c->bs_calls.clear(); c->have_bs = false;
    c->keep_motion_l2 = 0; c->grid_zeroed = false;
    c->stats = ohevc_frame_stats{};
    for (int &v : c->nstat) v = 0;
    c->alg = 0;
    c->owner = std::this_thread::get_id();
    c->epoch.fetch_add(1, std::memory_order_release);      // per-thread recorder caches of the previous picture are void
    {
        std::lock_guard<std::mutex> g(c->side_m);
        for (auto &sd : c->side) { clear_rec(*sd.second); sd.second->dbk_v.clear(); sd.second->dbk_h.clear(); sd.second->bs_calls.clear(); sd.second->sao.clear(); for (int &v : sd.second->nstat) v = 0; sd.second->alg = 0; }
    }
    return OHEVC_OK;
}

// CTB executor (frame_mode 2): the raster index of the CTB that holds sample (x, y) of `plane`
static inline uint32_t ctb_index(const ohevc_ctx *c, const Picture *p, int plane, int x, int y, int log2_ctb)
{
    const int hs = plane ? (p->cfi == 1 || p->cfi == 2) : 0, vs = plane ? (p->cfi == 1) : 0;
    const int ctb_w = (p->w + (1 << log2_ctb) - 1) >> log2_ctb;
    (void)c;
    return (uint32_t)(((y << vs) >> log2_ctb) * ctb_w + ((x << hs) >> log2_ctb));
}
// is intra work recorded in CTB order (modes 2 and 3, once the picture's intra jobs have named a CTB size) / in dependency levels?
static inline bool rec_ctb(const ohevc_ctx *c) { return c->frame_mode >= 2 && __atomic_load_n(&c->log2_ctb, __ATOMIC_RELAXED) > 0; }
static inline bool rec_levels(const ohevc_ctx *c) { return c->frame_mode != 3 || __atomic_load_n(&c->log2_ctb, __ATOMIC_RELAXED) <= 0; }

// one N x N block into the recorder's arena; returns its offset in the DENSE arena.  cols / rows: the rectangle that can hold non-zero
// coefficients (multiples of 4; N x N = everything)
static inline uint32_t arena_put(Rec &r, const int16_t *coeffs, int log2, int cols, int rows, bool groups = false)
{
    const int n = 1 << log2;
    const uint32_t dst = r.dense, src = (uint32_t)r.coeffs.size();
    r.dense += (uint32_t)(n * n);
    if (log2 >= 3 && groups) {
        // the sub-block form (ohevc_hip.h): the 4x4 groups of the rectangle that hold a non-zero coefficient, 16 elements each, and one bit per group
        const int gpr = n >> 2, gcols = cols >> 2, parts = (log2 == 5 && rows > 16) ? 2 : 1, grows_all = rows >> 2;
        for (int part = 0; part < parts; part++) {
            const int gy0 = part * 4, gy1 = std::min(grows_all, log2 == 5 ? gy0 + 4 : gpr);
            const uint32_t at = (uint32_t)r.coeffs.size();
            r.coeffs.resize((size_t)at + (size_t)(gy1 - gy0) * gcols * 16);
            int16_t *d = r.coeffs.data() + at;
            uint32_t mask = 0;
            for (int gy = gy0; gy < gy1; gy++)
                for (int gx = 0; gx < gcols; gx++) {
                    const int16_t *g4 = coeffs + (size_t)(gy * 4) * n + gx * 4;
                    uint64_t q[4];
                    for (int k = 0; k < 4; k++) memcpy(&q[k], g4 + (size_t)k * n, 8);
                    if (!(q[0] | q[1] | q[2] | q[3])) continue;
                    memcpy(d, q, 32);
                    d += 16;
                    mask |= 1u << ((gy - gy0) * gpr + gx);
                }
            r.coeffs.resize((size_t)(d - r.coeffs.data()));
            const uint32_t code = log2 != 5 ? 0u : parts == 2 ? (uint32_t)part : 2u;
            r.expand.push_back(ohevc_expand_rec{ at, dst + (uint32_t)part * 512u, mask, 0x100u | (uint32_t)log2 | (code << 9) });
        }
        return dst;
    }
    if (log2 >= 3 && (cols < n || rows < n)) {
        r.coeffs.resize((size_t)src + (size_t)cols * rows);
        int16_t *d = r.coeffs.data() + src;
        for (int y = 0; y < rows; y++) memcpy(d + (size_t)y * cols, coeffs + (size_t)y * n, (size_t)cols * sizeof(int16_t));
        r.expand.push_back(ohevc_expand_rec{ src, dst, (uint32_t)cols | ((uint32_t)rows << 8), (uint32_t)log2 });
        return dst;
    }
    r.coeffs.insert(r.coeffs.end(), coeffs, coeffs + n * n);     // whole: runs of whole blocks share a record (at most 1024 elements: one wavefront's work)
    if (!r.expand.empty()) {
        ohevc_expand_rec &e = r.expand.back();
        if (e.kind == 0 && e.src + e.dims == src && e.dst + e.dims == dst && e.dims + (uint32_t)(n * n) <= 1024u) { e.dims += (uint32_t)(n * n); return dst; }
    }
    r.expand.push_back(ohevc_expand_rec{ src, dst, (uint32_t)(n * n), 0u });
    return dst;
}

static int rec_tu_impl(ohevc_ctx *c, int plane, int x, int y, int log2, int kind, const int16_t *coeffs, int intra, int cols, int rows);

extern "C" int ohevc_rec_tu(ohevc_ctx *c, int plane, int x, int y, int log2, int kind, const int16_t *coeffs, int intra)
{
    return rec_tu_impl(c, plane, x, y, log2, kind, coeffs, intra, 64, 64);
}

// ohevc_rec_tu with the caller's promise that every coefficient outside the top-left cols x rows rectangle is zero (inverse-DCT blocks: what
// the reference passes to its idct slot as col_limit bounds them, hevc_cabac.c:1923-1934: cols = min(col_limit, N), rows = min(col_limit + 4, N))
extern "C" int ohevc_rec_tu_limited(ohevc_ctx *c, int plane, int x, int y, int log2, int kind, const int16_t *coeffs, int intra, int cols, int rows)
{
    OHEVC_REQUIRE(cols >= 1 && rows >= 1, "empty coefficient rectangle");
    return rec_tu_impl(c, plane, x, y, log2, kind, coeffs, intra, cols, rows);
}

static int rec_tu_impl(ohevc_ctx *c, int plane, int x, int y, int log2, int kind, const int16_t *coeffs, int intra, int cols, int rows)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    Rec &r = pick(c);
    OHEVC_REQUIRE(plane >= 0 && plane < 3 && log2 >= 2 && log2 <= 5 && kind >= 0 && kind < OHEVC_TU_NKINDS, "bad TU");
    OHEVC_REQUIRE(kind != OHEVC_TU_DST4 || log2 == 2, "DST is 4x4 only");
    const int n = 1 << log2;
    OHEVC_REQUIRE(x >= 0 && y >= 0 && x + n <= p->planes[plane].width && y + n <= p->planes[plane].height && coeffs != nullptr, "TU outside plane");
    ohevc_tu_job j = {};
    j.x = (uint16_t)x; j.y = (uint16_t)y; j.plane = (uint8_t)plane;
    r.alg += (kind == OHEVC_TU_DC ? 2 : 2 * n * n) + (kind == OHEVC_TU_PCM ? 1 : 2) * (p->bd > 8 ? 2 : 1) * n * n;
    if (kind == OHEVC_TU_DC) {
        j.dc = coeffs[0];
    } else {
        // (the caller's buffer is reused by the next TU: copied now).  Only the plain inverse DCT has a known-zero remainder.
        const bool limited = kind == OHEVC_TU_IDCT && g_compact_coeffs;
        j.coeff_off = arena_put(r, coeffs, log2, limited ? std::min(n, (cols + 3) & ~3) : n, limited ? std::min(n, (rows + 3) & ~3) : n, limited && g_compact_coeffs == 2);
    }
    // `intra`: the block MAY have been predicted by an intra job of this picture (the table slots cannot tell and always say so): the
    // level map knows -- 0 = no intra job covered it: the residual of an inter block (or PCM samples), level 0
    const int level = intra ? c->level_map[plane][(size_t)(y >> 2) * c->lm_w[plane] + (x >> 2)] : 0;
    if (level > 0 && rec_ctb(c)) {                            // follows its block's prediction inside the CTB's task
        r.ctb_ops.emplace_back(ctb_index(c, p, plane, x, y, c->log2_ctb), 0x80000000u | ((uint32_t)(log2 - 2) << 29) | ((uint32_t)kind << 25) | (uint32_t)r.ctb_tu.size());
        r.ctb_tu.push_back(j);
        if (!rec_levels(c)) { r.nstat[0]++; return OHEVC_OK; }
    }
    if (trace_hit(plane, x, y, n, n))
        fprintf(stderr, "trace: target %d tu plane %d x %d y %d log2 %d kind %d level %d c0 %d\n", c->cur, plane, x, y, log2, kind, level, coeffs[0]);
    // the residual of the block that was just predicted (hls_transform_unit predicts a block and adds its residual back to back,
    // hevc.c:1214-1215, 1260-1290) rides with its prediction job: one launch per dependency level instead of two
    auto &li = r.last_intra;
    if (g_fuse_intra && !c->dry && c->frame_mode != 1 && level > 0 && li.level == level && li.plane == plane && li.x == x && li.y == y && li.log2 == log2) {
        j.reserved0 = (uint8_t)(kind + 1);
        r.levels[level].intra_res[li.index] = j;
        li.level = -1;
        r.nstat[0]++;
        return OHEVC_OK;
    }
    LevelBins &lb = level_bins(r, level);
    lb.tu[log2 - 2][kind].push_back(j);
    lb.touched |= 1ull << ((log2 - 2) * 16 + kind);
    r.nstat[0]++;
    return OHEVC_OK;
}

extern "C" int ohevc_rec_tu_cross(ohevc_ctx *c, int plane, int x, int y, int log2, int kind_c, const int16_t *coeffs_c, int kind_y,
                                  const int16_t *coeffs_y, int res_scale_val, int intra)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    Rec &r = pick(c);
    OHEVC_REQUIRE(plane >= 1 && plane < 3 && log2 >= 2 && log2 <= 5, "cross-component prediction applies to chroma blocks");
    OHEVC_REQUIRE(kind_y >= 0 && kind_y < OHEVC_TU_PCM && kind_c >= -1 && kind_c < OHEVC_TU_PCM && coeffs_y != nullptr && (kind_c < 0 || coeffs_c != nullptr),
                  "bad residual kinds");
    OHEVC_REQUIRE((kind_y != OHEVC_TU_DST4 && kind_c != OHEVC_TU_DST4) || log2 == 2, "DST is 4x4 only");
    OHEVC_REQUIRE(res_scale_val >= -8 && res_scale_val <= 8, "res_scale_val out of range");
    const int n = 1 << log2;
    OHEVC_REQUIRE(x >= 0 && y >= 0 && x + n <= p->planes[plane].width && y + n <= p->planes[plane].height, "TU outside plane");
    ohevc_tu_job j = {};
    j.x = (uint16_t)x; j.y = (uint16_t)y; j.plane = (uint8_t)plane;
    j.reserved0 = (uint8_t)((kind_c < 0 ? 15 : kind_c) | (kind_y << 4));
    j.dc = (int16_t)res_scale_val;
    j.reserved1 = arena_put(r, coeffs_y, log2, n, n);
    r.alg += (kind_c >= 0 ? 4 : 2) * n * n + 2 * (p->bd > 8 ? 2 : 1) * n * n;
    if (kind_c >= 0) j.coeff_off = arena_put(r, coeffs_c, log2, n, n);
    const int level = intra ? c->level_map[plane][(size_t)(y >> 2) * c->lm_w[plane] + (x >> 2)] : 0;
    if (level > 0 && rec_ctb(c)) {
        r.ctb_ops.emplace_back(ctb_index(c, p, plane, x, y, c->log2_ctb), 0x80000000u | ((uint32_t)(log2 - 2) << 29) | ((uint32_t)OHEVC_TU_CROSS << 25) | (uint32_t)r.ctb_tu.size());
        r.ctb_tu.push_back(j);
        if (!rec_levels(c)) { r.nstat[0]++; return OHEVC_OK; }
    }
    LevelBins &lb = level_bins(r, level);
    lb.tu[log2 - 2][OHEVC_TU_CROSS].push_back(j);
    lb.touched |= 1ull << ((log2 - 2) * 16 + OHEVC_TU_CROSS);
    r.nstat[0]++;
    return OHEVC_OK;
}

extern "C" int ohevc_rec_mc(ohevc_ctx *c, const ohevc_mc_job *job)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr && job != nullptr, "no frame begun");
    Rec &r = pick(c);
    OHEVC_REQUIRE(job->plane < 3 && job->w >= 2 && job->w <= 64 && job->h >= 2 && job->h <= 64, "bad MC block");
    OHEVC_REQUIRE(get_pic(c, job->ref0) != nullptr && (!(job->flags & OHEVC_MC_BI) || get_pic(c, job->ref1) != nullptr), "bad reference slot");
    if (trace_hit(job->plane, job->x, job->y, job->w, job->h))
        fprintf(stderr, "trace: target %d mc plane %d x %d y %d w %d h %d flags %d ref0 %d (%d,%d)+(%d,%d) ref1 %d (%d,%d)+(%d,%d) denom %d w %d %d o %d %d\n",
                c->cur, job->plane, job->x, job->y, job->w, job->h, job->flags, job->ref0, job->sx0, job->sy0, job->mx0, job->my0, job->ref1,
                job->sx1, job->sy1, job->mx1, job->my1, job->denom, job->wx0, job->wx1, job->ox0, job->ox1);
    // Prediction blocks are cut into tiles of at most 16x16 samples (every tile is an independent job: same references,
    // positions shifted by the tile offset), so a 64x64 PU spreads over 16 wavefronts; tiles of at most 8x8 go to the
    // packed small-block kernel (four per wavefront).
    for (int ty = 0; ty < job->h; ty += 16)
        for (int tx = 0; tx < job->w; tx += 16) {
            ohevc_mc_job t = *job;
            t.x = (uint16_t)(job->x + tx); t.y = (uint16_t)(job->y + ty);
            t.w = (uint8_t)std::min(16, job->w - tx); t.h = (uint8_t)std::min(16, job->h - ty);
            t.sx0 = (int16_t)(job->sx0 + tx); t.sy0 = (int16_t)(job->sy0 + ty);
            t.sx1 = (int16_t)(job->sx1 + tx); t.sy1 = (int16_t)(job->sy1 + ty);
            ((t.w <= 8 && t.h <= 8) ? r.mc_small : r.mc).push_back(t);
        }
    {
        const int P = p->bd > 8 ? 2 : 1, T = job->plane ? 4 : 8;
        r.alg += (int64_t)P * (job->w + T - 1) * (job->h + T - 1) * ((job->flags & OHEVC_MC_BI) ? 2 : 1) + (int64_t)P * job->w * job->h;
        // the deepest reference row the block's filter taps touch (luma: 4 rows below the block, chroma: 2), in luma rows; rows beyond the
        // picture are the clamped last row.  What a frame-parallel subscriber has to have received before this picture launches
        // (hevc_await_progress waits for y0 + (mv.y >> 2) + nPbH + 9, hevc.c:1951-1958).
        const int vs = (job->plane && p->planes[0].height > p->planes[job->plane].height) ? 1 : 0;
        const int below = job->plane ? 2 : 4;
        auto note = [&](int slot, int sy) {
            int row = ((sy + job->h + below) << vs) + vs;
            row = row < 0 ? 0 : row > 32767 ? 32767 : row;
            if ((unsigned)slot <= (unsigned)OHEVC_MAX_PICTURES && row > r.reach[slot]) r.reach[slot] = (int16_t)row;
        };
        note(job->ref0, job->sy0);
        if (job->flags & OHEVC_MC_BI) note(job->ref1, job->sy1);
    }
    r.nstat[1]++;
    return OHEVC_OK;
}

static int rec_intra_impl(ohevc_ctx *c, const ohevc_intra_job *job);

extern "C" int ohevc_rec_intra_cip(ohevc_ctx *c, const ohevc_intra_job *job, const ohevc_intra_cip *cip)
{
    OHEVC_REQUIRE(c != nullptr && job != nullptr, "null argument");
    Rec &r = pick(c);
    ohevc_intra_job j = *job;
    if (j.flags2 & OHEVC_INTRA2_CIP) {
        OHEVC_REQUIRE(cip != nullptr, "CIP job without side record");
        j.cip_index = (uint32_t)r.cips.size();
        r.cips.push_back(*cip);
    }
    return rec_intra_impl(c, &j);
}

extern "C" int ohevc_rec_intra(ohevc_ctx *c, const ohevc_intra_job *job)
{
    OHEVC_REQUIRE(job == nullptr || !(job->flags2 & OHEVC_INTRA2_CIP), "constrained-intra jobs go through ohevc_rec_intra_cip");
    return rec_intra_impl(c, job);
}

static int rec_intra_impl(ohevc_ctx *c, const ohevc_intra_job *job)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr && job != nullptr, "no frame begun");
    Rec &r = pick(c);
    OHEVC_REQUIRE(job->plane < 3 && job->log2_size >= 2 && job->log2_size <= 5 && job->mode <= 34, "bad intra job");
    const int pl = job->plane, n = 1 << job->log2_size, W = c->lm_w[pl], H = c->lm_h[pl];
    OHEVC_REQUIRE(job->x + n <= p->planes[pl].width && job->y + n <= p->planes[pl].height, "intra block outside plane");
    r.alg += (p->bd > 8 ? 2 : 1) * (4 * n + 1 + n * n);
    if (c->frame_mode >= 2) {
        // the CTB executor needs the CTB size; the picture's first intra job decides (jobs built without it: dependency levels)
        int l2 = __atomic_load_n(&c->log2_ctb, __ATOMIC_RELAXED);
        if (l2 == 0) {
            l2 = job->log2_ctb_size >= 4 && job->log2_ctb_size <= 6 && n <= (1 << job->log2_ctb_size) ? job->log2_ctb_size : -1;
            __atomic_store_n(&c->log2_ctb, l2, __ATOMIC_RELAXED);
        }
        if (l2 > 0) {
            OHEVC_REQUIRE(job->log2_ctb_size == l2, "the intra jobs of one picture must name one CTB size");
            r.ctb_ops.emplace_back(ctb_index(c, p, pl, job->x, job->y, l2), (uint32_t)r.ctb_intra.size());
            r.ctb_intra.push_back(*job);
            if (c->frame_mode == 3) {                         // no levels needed: just mark the block's cells as intra-predicted
                if (trace_hit(pl, job->x, job->y, n, n))
                    fprintf(stderr, "trace: target %d intra plane %d x %d y %d log2 %d mode %d flags 0x%x flags2 0x%x bl %d tr %d (ctb task)\n", c->cur, pl, job->x,
                            job->y, job->log2_size, job->mode, job->flags, job->flags2, job->bottom_left_size, job->top_right_size);
                uint16_t *lmp = c->level_map[pl].data();
                for (int cy = job->y >> 2; cy < (job->y + n) >> 2; cy++)
                    for (int cx = job->x >> 2; cx < (job->x + n) >> 2; cx++) __atomic_store_n(&lmp[(size_t)cy * W + cx], (uint16_t)1, __ATOMIC_RELAXED);
                r.nstat[2]++;
                return OHEVC_OK;
            }
        }
    }
    // dependency level = 1 + the highest level among the 4x4 cells this block may read (row above incl. corner and
    // above-right, column to the left incl. below-left): hevcpred_template.c:164-183
    // With slice threads (ohevc_ctx_set_concurrent) the cells of a neighbouring tile / WPP row are written by another thread while
    // this one looks at them.  Cells of blocks this block really reads were written before (the reference's own row / tile
    // synchronisation orders them); the others belong to unavailable neighbours, whose samples the kernel never touches, so any
    // value read there only makes the level higher than necessary.  Relaxed atomics keep those accesses well defined.
    uint16_t *lm = c->level_map[pl].data();
    auto ld = [&](size_t i) { return (int)__atomic_load_n(&lm[i], __ATOMIC_RELAXED); };
    // Which of the five neighbour groups - in the reference's scan order below-left, left, corner, above, above-right - can reach the
    // prediction?  (1) what the predictor of this mode reads (hevcpred_template.c:359-537; the whole set whenever the [1 2 1] / strong
    // smoothing of :289-327 applies, for the negative angles and for constrained intra prediction); (2) an unavailable group is filled
    // from the group before it in scan order (already resolved), the below-left one from the first available group after it (:251-286).
    // A block that predicts from the row above only does not wait for its left neighbour: shorter dependency chains, fewer launches.
    enum { G_BL = 1, G_L = 2, G_UL = 4, G_U = 8, G_UR = 16, G_ALL = 31 };
    unsigned need;
    {
        const int mode = job->mode, log2 = job->log2_size;
        const bool luma_edge = (job->flags & OHEVC_INTRA_LUMA_EDGE) && n < 32;
        bool smooth = false;
        if (!(job->flags & OHEVC_INTRA_NO_SMOOTHING) && mode != 1 && n != 4) {
            const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
            smooth = (dv < dh ? dv : dh) > (log2 == 3 ? 7 : log2 == 4 ? 1 : 0);
        }
        // smoothed reference samples: filtered[k] reads k - 1 .. k + 1 of the same array (k = 0: the corner); the strong (bilinear) form of
        // 32x32 luma blocks decides on both arrays
        const bool strong = smooth && (job->flags & OHEVC_INTRA_STRONG) && log2 == 5;
        if ((job->flags2 & OHEVC_INTRA2_CIP) || strong) need = G_ALL;
        else if (smooth) need = mode >= 27 ? G_UL | G_U | G_UR : (mode >= 2 && mode <= 9) ? G_UL | G_L | G_BL : G_ALL;
        else if (mode == 0) need = G_BL | G_L | G_U | G_UR;
        else if (mode == 1) need = G_L | G_U;
        else if (mode < 10) need = G_L | G_BL;
        else if (mode == 10) need = G_L | (luma_edge ? G_U | G_UL : 0);
        else if (mode < 26) need = G_ALL;
        else if (mode == 26) need = G_U | (luma_edge ? G_L | G_UL : 0);
        else need = G_U | G_UR;
    }
    unsigned src = 0;                                          // the available groups the needed ones take their samples from
    {
        const unsigned avail = job->flags & 31u;               // OHEVC_INTRA_BOTTOM_LEFT .. OHEVC_INTRA_UP_RIGHT = bits 0..4, scan order
        for (int g = 0; g < 5; g++) {
            if (!(need >> g & 1)) continue;
            int j = g;
            while (j >= 0 && !(avail >> j & 1)) j--;           // the group itself, or the nearest available one before it ...
            if (j < 0) { j = g + 1; while (j < 5 && !(avail >> j & 1)) j++; }      // ... or the first one after it
            if (j < 5) src |= 1u << j;
        }
    }
    int level = 0;
    const int cx0 = (job->x >> 2) - 1, cy0 = (job->y >> 2) - 1, cn = n >> 2;              // cells: column left of / row above the block
    const int cxb = job->x >> 2, cyb = job->y >> 2;
    auto row_cells = [&](int x_first, int x_last) {            // cells [x_first, x_last] of the row above
        if (cy0 < 0) return;
        for (int cx = std::max(x_first, 0); cx <= std::min(x_last, W - 1); cx++) level = std::max(level, ld((size_t)cy0 * W + cx));
    };
    auto col_cells = [&](int y_first, int y_last) {            // cells [y_first, y_last] of the column to the left
        if (cx0 < 0) return;
        for (int cy = std::max(y_first, 0); cy <= std::min(y_last, H - 1); cy++) level = std::max(level, ld((size_t)cy * W + cx0));
    };
    if (src & G_BL) col_cells(cyb + cn, cyb + 2 * cn - 1);
    if (src & G_L) col_cells(cyb, cyb + cn - 1);
    if (src & G_UL) { if (cx0 >= 0) row_cells(cx0, cx0); }
    if (src & G_U) row_cells(cxb, cxb + cn - 1);
    if (src & G_UR) row_cells(cxb + cn, cxb + 2 * cn - 1);
    level += 1;
    OHEVC_REQUIRE(level < 65535, "intra dependency chain too long");
    for (int cy = job->y >> 2; cy < (job->y + n) >> 2; cy++)
        for (int cx = job->x >> 2; cx < (job->x + n) >> 2; cx++) __atomic_store_n(&lm[(size_t)cy * W + cx], (uint16_t)level, __ATOMIC_RELAXED);
    if (trace_hit(pl, job->x, job->y, n, n))
        fprintf(stderr, "trace: target %d intra plane %d x %d y %d log2 %d mode %d flags 0x%x flags2 0x%x bl %d tr %d level %d\n", c->cur, pl, job->x,
                job->y, job->log2_size, job->mode, job->flags, job->flags2, job->bottom_left_size, job->top_right_size, level);
    LevelBins &lbi = level_bins(r, level);
    lbi.intra.push_back(*job);
    if (g_fuse_intra && !c->dry && c->frame_mode != 1) {
        lbi.intra_res.push_back(ohevc_tu_job{});
        r.last_intra.level = level; r.last_intra.index = (int)lbi.intra.size() - 1;
        r.last_intra.plane = pl; r.last_intra.x = job->x; r.last_intra.y = job->y; r.last_intra.log2 = job->log2_size;
    }
    r.nstat[2]++;
    return OHEVC_OK;
}

extern "C" int ohevc_rec_deblock(ohevc_ctx *c, const ohevc_dbk_job *job)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr && job != nullptr, "no frame begun");
    Rec &r = pick(c);
    ((job->flags & OHEVC_DBK_VERTICAL_EDGE) ? r.dbk_v : r.dbk_h).push_back(*job);
    if (g_trace_at_on) trace_dbk(c->cur, *job);
    r.alg += 2 * (c->store->pics[c->cur].bd > 8 ? 2 : 1) * (job->plane ? 32 : 64);      // 8 lines x 4 (chroma: 2) samples either side, read + written
    r.nstat[3]++;
    return OHEVC_OK;
}

extern "C" int ohevc_rec_sao(ohevc_ctx *c, const ohevc_sao_job *job)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr && job != nullptr, "no frame begun");
    Rec &r = pick(c);
    r.sao.push_back(*job);
    if (g_trace_at_on) trace_sao(c->cur, *job);
    if (job->quirks & (OHEVC_SAO_LAG_BELOW | OHEVC_SAO_LAG_ABOVE | OHEVC_SAO_LAG_MID)) r.sao_lagged = true;
    r.alg += (int64_t)(c->store->pics[c->cur].bd > 8 ? 2 : 1) * ((job->w + 2) * (job->h + 2) + job->w * job->h);
    r.nstat[4]++;
    return OHEVC_OK;
}

extern "C" int ohevc_frame_set_bypass_map(ohevc_ctx *c, const uint8_t *map, int stride, int width_pu, int height_pu, int log2_min_pu_size,
                                          int exact_reference)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    if (!map) { c->bypass.clear(); return OHEVC_OK; }
    OHEVC_REQUIRE(log2_min_pu_size >= 2 && log2_min_pu_size <= 6 && width_pu > 0 && height_pu > 0 && stride >= width_pu, "bad map description");
    OHEVC_REQUIRE(((long long)width_pu << log2_min_pu_size) >= p->w && ((long long)height_pu << log2_min_pu_size) >= p->h, "map smaller than the picture");
    bool any = false;
    c->bypass.resize((size_t)width_pu * height_pu);
    for (int y = 0; y < height_pu; y++) {
        memcpy(c->bypass.data() + (size_t)y * width_pu, map + (size_t)y * stride, (size_t)width_pu);
        if (!any) for (int x = 0; x < width_pu; x++) any |= map[(size_t)y * stride + x] != 0;
    }
    if (!any) c->bypass.clear();              // nothing flagged: SAO runs as usual
    c->bypass_w = width_pu; c->bypass_l2 = log2_min_pu_size; c->bypass_exact = exact_reference != 0;
    return OHEVC_OK;
}

extern "C" int ohevc_rec_mc_bulk(ohevc_ctx *c, const ohevc_mc_job *jobs, int n)
{
    for (int i = 0; i < n; i++) { int rc = ohevc_rec_mc(c, jobs + i); if (rc != OHEVC_OK) return rc; }
    return OHEVC_OK;
}
extern "C" int ohevc_rec_intra_bulk(ohevc_ctx *c, const ohevc_intra_job *jobs, int n)
{
    for (int i = 0; i < n; i++) { int rc = ohevc_rec_intra(c, jobs + i); if (rc != OHEVC_OK) return rc; }
    return OHEVC_OK;
}
extern "C" int ohevc_rec_tu_bulk(ohevc_ctx *c, int n, const int32_t *desc, const int16_t *coeffs)
{
    OHEVC_REQUIRE(n == 0 || (desc != nullptr && coeffs != nullptr), "null argument");
    for (int i = 0; i < n; i++) {
        const int32_t *d = desc + 6 * i;
        OHEVC_REQUIRE(d[3] >= 2 && d[3] <= 5, "bad TU size");
        int rc = ohevc_rec_tu(c, d[0], d[1], d[2], d[3], d[4], coeffs, d[5]);
        if (rc != OHEVC_OK) return rc;
        coeffs += 1 << (2 * d[3]);
    }
    return OHEVC_OK;
}
extern "C" int ohevc_rec_deblock_bulk(ohevc_ctx *c, const ohevc_dbk_job *jobs, int n)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr && (n == 0 || jobs != nullptr) && n >= 0, "no frame begun");
    Rec &r = pick(c);
    for (int i = 0; i < n; i++) ((jobs[i].flags & OHEVC_DBK_VERTICAL_EDGE) ? r.dbk_v : r.dbk_h).push_back(jobs[i]);
    if (g_trace_at_on) for (int i = 0; i < n; i++) trace_dbk(c->cur, jobs[i]);
    for (int i = 0; i < n; i++) r.alg += 2 * (c->store->pics[c->cur].bd > 8 ? 2 : 1) * (jobs[i].plane ? 32 : 64);
    r.nstat[3] += n;
    return OHEVC_OK;
}
// The deblocking of the current picture, handed over as the decoder's own maps (ohevc_hip.h, ohevc_dbk_maps): copied here (the
// decoder reuses its arrays for the next picture), uploaded with the frame end's job arrays, derived and filtered on the device.
static int rec_deblock_maps_impl(ohevc_ctx *c, const ohevc_dbk_maps *m, bool with_bs);
extern "C" int ohevc_rec_deblock_maps(ohevc_ctx *c, const ohevc_dbk_maps *m) { return rec_deblock_maps_impl(c, m, true); }
// with_bs false: the two boundary-strength arrays are derived on the device (ohevc_rec_deblock_maps_bs) - they do not travel (they used to, as
// 2 x 133 KB of zeros per 1080p picture: a quarter of an encoder-like picture's upload)
static int rec_deblock_maps_impl(ohevc_ctx *c, const ohevc_dbk_maps *m, bool with_bs)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr && m != nullptr, "no frame begun");
    OHEVC_REQUIRE(!c->dry || c->dry_as_device, "record-only contexts take deblocking as jobs (no device to derive them)");
    OHEVC_REQUIRE(m->width > 0 && m->height > 0 && m->log2_ctb_size >= 4 && m->log2_ctb_size <= 6 && m->log2_min_cb_size >= 3 &&
                  m->chroma_format_idc >= 0 && m->chroma_format_idc <= 3, "picture geometry");
    OHEVC_REQUIRE((!with_bs || (m->horizontal_bs && m->vertical_bs)) && m->bs_width > 0 && m->qp_y_tab && m->min_cb_width > 0 && m->deblock && m->deblock_stride >= 2,
                  "deblocking maps");
    OHEVC_REQUIRE(!m->is_pcm || (m->min_pu_width > 0 && m->min_pu_height > 0 && m->log2_min_pu_size >= 2), "pcm map");
    const int hs = m->chroma_format_idc == 1 || m->chroma_format_idc == 2, vs = m->chroma_format_idc == 1;
    const int ctb = 1 << m->log2_ctb_size, ctb_w = (m->width + ctb - 1) >> m->log2_ctb_size, ctb_h = (m->height + ctb - 1) >> m->log2_ctb_size;
    const size_t bs_h = (size_t)(m->height >> 2);
    const size_t n_v = with_bs ? (size_t)m->bs_width * (bs_h + (4u << vs)) : 0, n_h = with_bs ? ((size_t)m->bs_width + (4u << hs)) * bs_h : 0;          // hevc.c:170-171
    const size_t n_qp = (size_t)m->min_cb_width * (size_t)(m->height >> m->log2_min_cb_size);
    const size_t n_db = (size_t)ctb_w * ctb_h * m->deblock_stride, n_pcm = m->is_pcm ? (size_t)m->min_pu_width * m->min_pu_height : 0;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_v = 0, o_h = o_v + up(n_v), o_qp = o_h + up(n_h), o_db = o_qp + up(n_qp), o_pcm = o_db + up(n_db), total = o_pcm + up(n_pcm);
    c->dbk_blob.resize(total);
    if (n_v) memcpy(c->dbk_blob.data() + o_v, m->vertical_bs, n_v);
    if (n_h) memcpy(c->dbk_blob.data() + o_h, m->horizontal_bs, n_h);
    memcpy(c->dbk_blob.data() + o_qp, m->qp_y_tab, n_qp);
    memcpy(c->dbk_blob.data() + o_db, m->deblock, n_db);
    if (n_pcm) memcpy(c->dbk_blob.data() + o_pcm, m->is_pcm, n_pcm);
    c->dbk_maps = *m;
    c->dbk_maps.vertical_bs = reinterpret_cast<const uint8_t *>(o_v); c->dbk_maps.horizontal_bs = reinterpret_cast<const uint8_t *>(o_h);
    c->dbk_maps.qp_y_tab = reinterpret_cast<const int8_t *>(o_qp); c->dbk_maps.deblock = reinterpret_cast<const int8_t *>(o_db);
    c->dbk_maps.is_pcm = n_pcm ? reinterpret_cast<const uint8_t *>(o_pcm) : nullptr;
    {   // the frame bound of deblocking, 2P bytes per sample of every plane
        const Picture &pp = c->store->pics[c->cur];
        for (const ohevc_plane &pl : pp.planes) c->alg += 2ll * (pp.bd > 8 ? 2 : 1) * pl.width * pl.height;
    }
    c->nstat[3]++;
    c->n_map_frames++;
    return OHEVC_OK;
}
// One call of ff_hevc_deblocking_boundary_strengths (hevc.c:1578,1607,2400,2484), recorded instead of executed: ohevc_dev_boundary_strengths
// evaluates the picture's calls at its frame end.
extern "C" int ohevc_rec_bs_call(ohevc_ctx *c, int x0, int y0, int log2_size, int flags)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr, "no frame begun");
    Rec &r = pick(c);
    ohevc_bs_call b = { (uint16_t)x0, (uint16_t)y0, (uint8_t)log2_size, (uint8_t)flags, 0 };
    r.bs_calls.push_back(b);
    return OHEVC_OK;
}

extern "C" int ohevc_rec_bs_calls(ohevc_ctx *c, const ohevc_bs_call *calls, int n)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr && n >= 0 && (n == 0 || calls != nullptr), "no frame begun / null array");
    Rec &r = pick(c);
    r.bs_calls.insert(r.bs_calls.end(), calls, calls + n);
    return OHEVC_OK;
}

// ohevc_rec_deblock_maps with the boundary strengths derived on the device: m->vertical_bs / horizontal_bs are not read; the motion field and
// the cbf_luma map (HOST pointers in *bs) are copied like the other maps.  The calls come through ohevc_rec_bs_call.
extern "C" int ohevc_rec_deblock_maps_bs(ohevc_ctx *c, const ohevc_dbk_maps *m, const ohevc_bs_maps *bs)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr && m != nullptr && bs != nullptr, "no frame begun");
    OHEVC_REQUIRE(!c->dry || c->dry_as_device, "record-only contexts take deblocking as jobs (no device to derive them)");
    // bs->mvf NULL: the frame keeps the motion of its MC jobs on the device (ohevc_frame_keep_motion) - nothing to copy
    OHEVC_REQUIRE((bs->mvf != nullptr ? bs->mvf_stride >= 20 : c->keep_motion_l2 == bs->log2_min_pu_size) && bs->cbf_luma != nullptr && bs->min_pu_width > 0 &&
                  bs->min_pu_height > 0 && bs->min_tb_width > 0 && bs->min_tb_height > 0, "motion field / cbf map");
    // the two boundary-strength arrays are written by the device: nothing of them in the blob (their offsets are never used: frame_end_impl)
    int rc = rec_deblock_maps_impl(c, m, false);
    if (rc != OHEVC_OK) return rc;
    const size_t n_mvf = bs->mvf ? (size_t)bs->min_pu_width * bs->min_pu_height * (size_t)bs->mvf_stride : 0, n_cbf = (size_t)bs->min_tb_width * bs->min_tb_height;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_mvf = up(c->dbk_blob.size()), o_cbf = o_mvf + up(n_mvf);
    c->dbk_blob.resize(o_cbf + up(n_cbf));
    if (n_mvf) memcpy(c->dbk_blob.data() + o_mvf, bs->mvf, n_mvf);
    memcpy(c->dbk_blob.data() + o_cbf, bs->cbf_luma, n_cbf);
    c->bs_maps = *bs;
    c->bs_maps.mvf = bs->mvf ? reinterpret_cast<const uint8_t *>(o_mvf) : nullptr;
    c->bs_maps.cbf_luma = reinterpret_cast<const uint8_t *>(o_cbf);
    c->have_bs = true;
    return OHEVC_OK;
}
// The frame's boundary strengths will be derived from the motion of its own MC jobs (ohevc_dev_motion_grid): call after ohevc_frame_begin,
// before the first ohevc_frame_reconstruct.  log2_unit = sps->log2_min_pu_size, the granularity ohevc_bs_maps indexes the field with.
extern "C" int ohevc_frame_keep_motion(ohevc_ctx *c, int log2_unit)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr, "no frame begun");
    OHEVC_REQUIRE(!c->dry || c->dry_as_device, "record-only contexts have no device to keep it on");
    OHEVC_REQUIRE(log2_unit >= 2 && log2_unit <= 5, "log2_unit");
    OHEVC_REQUIRE(c->keep_motion_l2 == 0 || c->keep_motion_l2 == log2_unit, "the frame already keeps its motion at another granularity");
    c->keep_motion_l2 = log2_unit;
    return OHEVC_OK;
}
// the grid of the frame in flight, cleared once (units no MC job covers read as intra-predicted: pred_flag 0)
static int motion_grid_ready(ohevc_ctx *c, const Picture *p, int &gw, int &gh)
{
    const int u = 1 << c->keep_motion_l2;
    gw = (p->w + u - 1) >> c->keep_motion_l2; gh = (p->h + u - 1) >> c->keep_motion_l2;
    if (c->grid_zeroed) return OHEVC_OK;
    // behind the grid, room for the two boundary-strength arrays the frame end fills (they want zeros too, hevc.c:3207-3208): one memset for both
    const size_t grid_bytes = ((size_t)gw * gh * OHEVC_MOTION_GRID_ENTRY + 255) & ~(size_t)255;
    const size_t bs_bytes = 2 * ((((size_t)(p->w >> 2) + 8) * ((size_t)(p->h >> 2) + 8) + 255) & ~(size_t)255);
    if (grid_bytes + bs_bytes > c->d_grid.cap) {
        OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
        int rc = c->d_grid.reserve(grid_bytes + bs_bytes);
        if (rc != OHEVC_OK) return rc;
    }
    int rc = ohevc_dev_zero(c->d_grid.p, grid_bytes + bs_bytes, c->stream);     // (a launch, not hipMemsetAsync: see ohevc_dev_zero)
    if (rc != OHEVC_OK) return rc;
    c->grid_zeroed = true;
    c->grid_bs_off = grid_bytes; c->grid_bs_cap = bs_bytes;
    return OHEVC_OK;
}
extern "C" int ohevc_ctx_has_device(const ohevc_ctx *c) { return c && (!c->dry || c->dry_as_device); }
extern "C" int ohevc_rec_sao_bulk(ohevc_ctx *c, const ohevc_sao_job *jobs, int n)
{
    for (int i = 0; i < n; i++) { int rc = ohevc_rec_sao(c, jobs + i); if (rc != OHEVC_OK) return rc; }
    return OHEVC_OK;
}

// copy `bytes` of host data into the staging buffer at a 256-byte aligned offset; returns that offset
static size_t stage_put(std::vector<std::pair<const void *, size_t>> &parts, size_t &total, const void *src, size_t bytes)
{
    size_t off = total;
    parts.emplace_back(src, bytes);
    total += (bytes + 255) & ~(size_t)255;
    return off;
}

static int wait_staging_free(ohevc_ctx *c, int lane)
{
    if (c->staged_pending[lane]) {
        const double t0 = g_trace_timing ? now_s() : 0;
        OHEVC_HIP_TRY(hipEventSynchronize(c->staged[lane]));
        if (g_trace_timing) c->t_part[3] += now_s() - t0;
        c->staged_pending[lane] = false;
    }
    return OHEVC_OK;
}

static int upload_table(ohevc_ctx *c)
{
    std::vector<ohevc_plane> t;
    {
        std::lock_guard<std::mutex> g(c->store->m);
        if (c->table_version == c->store->version) return OHEVC_OK;
        t.resize((size_t)kMaxPics * 3);
        for (int s = 0; s < c->store->npics; s++)
            for (int i = 0; i < 3; i++) t[3 * s + i] = c->store->pics[s].used ? c->store->pics[s].planes[i] : ohevc_plane{};
        c->table_version = c->store->version;
    }
    int rc = c->d_table.reserve(t.size() * sizeof(ohevc_plane));
    if (rc != OHEVC_OK) return rc;
    // through page-locked memory of our own: a pageable source makes the runtime look the address up among the registered host ranges
    // (ohevc_host_pin: the decoder's frame buffers, registered and recycled by other decoding threads at this very moment) - seen once as
    // "invalid argument" out of this copy on a frame-threaded stream (profiles/r03end_pytest_gpu_flake.log)
    if ((rc = c->table_stage.reserve(t.size() * sizeof(ohevc_plane), 16384)) != OHEVC_OK) return rc;
    memcpy(c->table_stage.p, t.data(), t.size() * sizeof(ohevc_plane));      // (the previous copy out of it was waited for below)
    OHEVC_HIP_TRY(hipMemcpyAsync(c->d_table.p, c->table_stage.p, t.size() * sizeof(ohevc_plane), hipMemcpyHostToDevice, c->stream));
    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
    return OHEVC_OK;
}

// Make this context's stream wait for whatever other contexts of the store still do with the pictures this frame
// touches: the frame that reconstructs a reference picture (possibly not even issued yet by its decoding thread), and
// earlier readers / the earlier writer of the target picture's memory.
static int guard_pictures(ohevc_ctx *c, int target)
{
    std::vector<int> fresh;
    for (const auto *v : {&c->mc, &c->mc_small})
        for (const ohevc_mc_job &j : *v) {
            const int refs[2] = {j.ref0, (j.flags & OHEVC_MC_BI) ? j.ref1 : -1};
            for (int r : refs)
                if (r >= 0 && r != target && std::find(c->ref_slots.begin(), c->ref_slots.end(), r) == c->ref_slots.end()) {
                    c->ref_slots.push_back(r);
                    fresh.push_back(r);
                }
        }
    if (fresh.empty() && c->target_guarded) return OHEVC_OK;
    const double t0 = g_trace_timing ? now_s() : 0;
    struct Acc { ohevc_ctx *c; double t0; ~Acc() { if (g_trace_timing) c->t_wait_refs += now_s() - t0; } } acc{c, t0};
    std::unique_lock<std::mutex> lk(c->store->m);
    for (int r : fresh) {
        Picture &rp = c->store->pics[r];
        if (g_trace_order) fprintf(stderr, "order: ctx %p target %d needs ref %d (issued %d, event %p)\n", (void *)c, target, r, (int)rp.end_issued, (void *)rp.written);
        // (an executor context was taken from the issuer's queue because the versions of its references had been issued; `end_issued` may
        // already speak of a NEWER picture begun in the slot, whose work the issuer holds back until this reader has been issued)
        if (!c->is_exec && !wait_end_issued(c, rp, lk)) {
            set_error("reference picture %d was never completed by its decoding thread", r);
            return OHEVC_ERR_STATE;
        }
        if (rp.failed && !c->is_exec) { set_error("reference picture %d: its frame failed", r); return OHEVC_ERR_STATE; }
        if (rp.written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, rp.written, 0));
    }
    if (!c->target_guarded) {
        Picture &tp = c->store->pics[target];
        if (tp.written) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, tp.written, 0));
        for (hipEvent_t e : tp.readers) OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, e, 0));
        tp.readers.clear();
        c->target_guarded = true;
    }
    return OHEVC_OK;
}

// upload a set of job arrays in one H2D copy; fills offs[i] with the device offset of parts[i]
static int upload_jobs(ohevc_ctx *c, std::vector<std::pair<const void *, size_t>> &parts, size_t total, int lane)
{
    if (total == 0) return OHEVC_OK;
    int rc = wait_staging_free(c, lane);
    if (rc != OHEVC_OK) return rc;
    if ((rc = c->stage[lane].reserve(total)) != OHEVC_OK) return rc;
    if (total > c->d_jobs[lane].cap) {
        OHEVC_HIP_TRY(hipStreamSynchronize(c->up_stream));  // an upload may still write the old buffer ...
        OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));     // ... and in-flight kernels may still read it
        if ((rc = c->d_jobs[lane].reserve(total)) != OHEVC_OK) return rc;
    }
    size_t off = 0;
    const double t_copy = g_trace_timing ? now_s() : 0;
    for (auto &pr : parts) {
        memcpy(c->stage[lane].p + off, pr.first, pr.second);
        off += (pr.second + 255) & ~(size_t)255;
    }
    if (g_trace_timing) c->t_part[0] += now_s() - t_copy;
    // on the upload stream (see ohevc_ctx::up_stream): behind the earlier readers of this lane's device buffer, in front of this call's kernels
    if (c->lane_done_pending[lane]) OHEVC_HIP_TRY(hipStreamWaitEvent(c->up_stream, c->lane_done[lane], 0));
    OHEVC_HIP_TRY(hipMemcpyAsync(c->d_jobs[lane].p, c->stage[lane].p, total, hipMemcpyHostToDevice, c->up_stream));
    OHEVC_HIP_TRY(hipEventRecord(c->staged[lane], c->up_stream));
    OHEVC_HIP_TRY(hipStreamWaitEvent(c->stream, c->staged[lane], 0));
    c->staged_pending[lane] = true;
    c->stats.upload_bytes += (int64_t)total;
    return OHEVC_OK;
}

// CTB executor: sort the recorded intra operations by CTB (stable: decoding order inside a CTB is kept) and cut them into tasks, one per
// CTB, in raster order.  A task waits for the task of a neighbouring CTB (left, above-left, above, above-right: hevc.c:2779) only if
// one of its blocks really reads samples of that CTB: a prediction block reads the row above / the column left of itself as far as its
// availability flags say (hevcpred_template.c:164-183), so only blocks on the CTB's top row / left column reach into a neighbour.
// Returns the length of the longest chain of dependent tasks in estimated microseconds (what the launch will take at least).
static double build_ctb_tasks(ohevc_ctx *c, const Picture *p)
{
    c->ctb_tasks.clear(); c->ctb_opwords.clear();
    if (c->ctb_ops.empty()) return 0;
    const int l2 = c->log2_ctb, ctb_w = (p->w + (1 << l2) - 1) >> l2, ctb_h = (p->h + (1 << l2) - 1) >> l2;
    std::stable_sort(c->ctb_ops.begin(), c->ctb_ops.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first < b.first; });
    c->ctb_task_of.assign((size_t)ctb_w * ctb_h, -1);
    c->ctb_opwords.reserve(c->ctb_ops.size());
    static thread_local std::vector<double> cost;
    cost.clear();
    double longest = 0;
    const size_t nops = c->ctb_ops.size();
    for (size_t i = 0; i < nops;) {
        const uint32_t ctb = c->ctb_ops[i].first;
        ohevc_ctb_task t = {};
        t.cx = (uint16_t)(ctb % ctb_w); t.cy = (uint16_t)(ctb / ctb_w);
        t.first_op = (uint32_t)i;
        unsigned need = 0;                                  // bit 0 left, 1 above-left, 2 above, 3 above-right
        size_t k = i;
        for (; k < nops && c->ctb_ops[k].first == ctb; k++) {
            const uint32_t w = c->ctb_ops[k].second;
            c->ctb_opwords.push_back(w);
            if (w >> 31) continue;
            const ohevc_intra_job &j = c->ctb_intra[w & 0x1ffffffu];
            const int hs = j.plane ? (p->cfi == 1 || p->cfi == 2) : 0, vs = j.plane ? (p->cfi == 1) : 0;
            const int cw = (1 << l2) >> hs, ch = (1 << l2) >> vs, n = 1 << j.log2_size;
            const bool top = j.y == t.cy * ch, left = j.x == t.cx * cw;
            if (top && (j.flags & OHEVC_INTRA_UP)) need |= 4;
            if (top && (j.flags & OHEVC_INTRA_UP_RIGHT)) need |= j.x + n >= (t.cx + 1) * cw ? 8 : 4;
            if (j.flags & OHEVC_INTRA_UP_LEFT) need |= top && left ? 2 : top ? 4 : left ? 1 : 0;
            if (left && (j.flags & (OHEVC_INTRA_LEFT | OHEVC_INTRA_BOTTOM_LEFT))) need |= 1;
        }
        t.nops = (uint32_t)(k - i);
        const int nb[4][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
        double before = 0;
        for (int d = 0; d < 4; d++) {
            const int x = t.cx + nb[d][0], y = t.cy + nb[d][1];
            t.dep[d] = ((need >> d) & 1) && x >= 0 && y >= 0 && x < ctb_w ? c->ctb_task_of[(size_t)y * ctb_w + x] : -1;
            if (t.dep[d] >= 0) before = std::max(before, cost[(size_t)t.dep[d]]);
        }
        // measured on MI355X (profiles/r02s / r02u): ~6 us to pick a task up, load and store its tiles, ~2.3 us per operation (one wave, latency-bound)
        const double mine = before + 6.0 + 2.3 * t.nops;
        cost.push_back(mine);
        longest = std::max(longest, mine);
        c->ctb_task_of[ctb] = (int32_t)c->ctb_tasks.size();
        c->ctb_tasks.push_back(t);
        i = k;
    }
    return longest;
}

// The 32x32 inverse-DCT kernel takes 8 consecutive jobs as one tile and touches the picture in whole row segments of those 8 blocks
// (tu_idct32_tile1_kernel).  The decoder emits transform blocks CTB by CTB in z-scan, so 8 consecutive jobs are two CTBs - 2 x 2 blocks each,
// 128-byte row pieces at 8 bit - where 8 horizontal neighbours would be one 256-byte segment: measured 6 % slower on the headline batch
// (bench.py "zscan": 0.819 against 0.770 ms per 2^20 blocks).  Jobs of a bin are independent and find their coefficients through
// coeff_off, so their order is free: a stable counting sort by (plane, block row) - inside a block row the CTB order already is the x order -
// restores raster order for ~2 ns per job.
static void sort_tile_bin(std::vector<ohevc_tu_job> &v)
{
    const size_t n = v.size();
    if (n < 16) return;
    static thread_local std::vector<uint32_t> count;
    static thread_local std::vector<ohevc_tu_job> tmp;
    constexpr int kRows = 2048;                                // y < 65536: block rows of 32 samples
    count.assign((size_t)3 * kRows + 1, 0u);
    bool sorted = true;
    uint32_t prev = 0;
    for (const ohevc_tu_job &j : v) {
        const uint32_t key = (uint32_t)(j.plane % 3) * kRows + (j.y >> 5);
        sorted = sorted && key >= prev;
        prev = key;
        count[key + 1]++;
    }
    if (sorted) return;
    for (size_t k = 1; k < count.size(); k++) count[k] += count[k - 1];
    tmp.resize(n);
    for (const ohevc_tu_job &j : v) tmp[count[(uint32_t)(j.plane % 3) * kRows + (j.y >> 5)]++] = j;
    v.swap(tmp);
}

extern "C" int ohevc_frame_reconstruct(ohevc_ctx *c)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    merge_side(c);
    const double ctb_us = build_ctb_tasks(c, p);
    if (c->frame_mode == 2 && !c->ctb_tasks.empty()) {
        // both forms were recorded: keep the cheaper one.  The level form costs a prediction launch and a residual launch per level
        // (~10.5 us per level on the device and about as much launch work on the host: profiles/r02q, r02t)
        const double level_us = (g_fuse_intra ? 5.5 : 10.5) * std::max(c->max_level, 0);      // one launch per level when the residuals ride with their prediction
        c->stats.chose_ctbs = ctb_us < level_us;
    } else {
        c->stats.chose_ctbs = !c->ctb_tasks.empty();
    }
    if (c->stats.chose_ctbs) {                               // drop the level form of the intra work (level 0 = residuals of inter blocks stays)
        for (int l = 1; l <= c->max_level; l++) {
            LevelBins &lb = c->levels[l];
            for (uint64_t m = lb.touched; m; m &= m - 1) { const int b = __builtin_ctzll(m); lb.tu[b >> 4][b & 15].clear(); }
            lb.touched = 0;
            lb.intra.clear();
            lb.intra_res.clear();
        }
        c->max_level = std::min(c->max_level, c->levels.empty() ? -1 : 0);
    } else {
        c->ctb_tasks.clear(); c->ctb_opwords.clear();
    }
    if (ohevc::config().trace_ctb) {
        size_t l0 = 0;
        if (c->max_level >= 0) for (uint64_t m = c->levels[0].touched; m; m &= m - 1) { const int b = __builtin_ctzll(m); l0 += c->levels[0].tu[b >> 4][b & 15].size(); }
        unsigned long long h = 1469598103934665603ull;
        for (uint32_t w : c->ctb_opwords) h = (h ^ w) * 1099511628211ull;
        for (auto &t : c->ctb_tasks) for (int d = 0; d < 4; d++) h = (h ^ (unsigned)t.dep[d]) * 1099511628211ull;
        fprintf(stderr, "ctb trace: mode %d chose %d tasks %zu ops %zu intra %zu tu %zu max_level %d level0_tu %zu coeffs %zu hash %llx\n", c->frame_mode, c->stats.chose_ctbs,
                c->ctb_tasks.size(), c->ctb_opwords.size(), c->ctb_intra.size(), c->ctb_tu.size(), c->max_level, l0, c->coeffs.size(), h);
    }
    if (ohevc::config().trace_levels) {       // diagnosis: how wide the dependency levels are, in wavefronts of the packed intra kernel
        fprintf(stderr, "levels: target %d max_level %d waves:", c->cur, c->max_level);
        for (int l = 1; l <= c->max_level; l++) {
            int cnt[4] = {0, 0, 0, 0};
            for (const ohevc_intra_job &j : c->levels[l].intra) cnt[j.log2_size - 2]++;
            fprintf(stderr, " %d", (cnt[0] + 15) / 16 + (cnt[1] + 7) / 8 + (cnt[2] + 3) / 4 + (cnt[3] + 1) / 2);
        }
        fprintf(stderr, "\n");
    }
    if (ohevc::config().trace_upload) {       // what this hand-over puts on the bus, by kind (bytes; every array is padded to 256 in the staging buffer)
        size_t intra = 0, intra_res = 0, tu = 0;
        for (int l = 0; l <= c->max_level; l++) {
            intra += c->levels[l].intra.size() * sizeof(ohevc_intra_job); intra_res += c->levels[l].intra_res.size() * sizeof(ohevc_tu_job);
            for (uint64_t m = c->levels[l].touched; m; m &= m - 1) { const int b = __builtin_ctzll(m); tu += c->levels[l].tu[b >> 4][b & 15].size() * sizeof(ohevc_tu_job); }
        }
        fprintf(stderr, "upload: target %d mc %zu mc_small %zu intra %zu intra_res %zu tu %zu coeffs %zu (dense %zu) expand %zu cips %zu ctb %zu levels %d\n", c->cur,
                c->mc.size() * sizeof(ohevc_mc_job), c->mc_small.size() * sizeof(ohevc_mc_job), intra, intra_res, tu, c->coeffs.size() * 2, (size_t)c->dense * 2,
                c->expand.size() * sizeof(ohevc_expand_rec), c->cips.size() * sizeof(ohevc_intra_cip),
                c->ctb_tasks.size() * sizeof(ohevc_ctb_task) + c->ctb_opwords.size() * 4 + c->ctb_intra.size() * sizeof(ohevc_intra_job) + c->ctb_tu.size() * sizeof(ohevc_tu_job), c->max_level);
    }
    if (c->dry) {
        if (g_sink) g_sink(g_sink_user, c, 0);
        clear_recorded(c);
        return OHEVC_OK;
    }
    OHEVC_HIP_TRY(hipSetDevice(c->device));
    if (c->mc.empty() && c->mc_small.empty() && c->max_level < 0 && c->ctb_tasks.empty()) return OHEVC_OK;
    int rc;
    if (!c->is_exec && g_long_chain_levels > 0 && c->max_level >= g_long_chain_levels && (rc = select_stream(c, true)) != OHEVC_OK) return rc;
    if ((rc = upload_table(c)) != OHEVC_OK) return rc;
    if ((rc = guard_pictures(c, c->cur)) != OHEVC_OK) return rc;

    // ---- stage every job array + the coefficient arena, one H2D copy
    std::vector<std::pair<const void *, size_t>> parts;
    size_t total = 0;
    const size_t off_mc = c->mc.empty() ? 0 : stage_put(parts, total, c->mc.data(), c->mc.size() * sizeof(ohevc_mc_job));
    const size_t off_mcs = c->mc_small.empty() ? 0 : stage_put(parts, total, c->mc_small.data(), c->mc_small.size() * sizeof(ohevc_mc_job));
    // per level: the intra jobs, then every touched (size, kind) bin back to back (one segmented launch per level)
    struct LevelOff { size_t intra = 0, intra_res = 0, tu_first = 0; int32_t count[4] = {0, 0, 0, 0}; bool packed = false; };
    std::vector<LevelOff> loff((size_t)(c->max_level + 1));
    for (int l = 0; l <= c->max_level; l++) {
        LevelBins &lb = c->levels[l];
        // OHEVC_REVERSE_LEVELS=1 (tests over the emulated device code, whose workgroups run one after the other in launch order): the
        // jobs of a level are independent, so their order must not matter - a dependency the level computation missed shows up
        const bool reverse_levels = g_reverse_levels;          // ohevc_debug_set_reverse_levels
        if (reverse_levels) {
            std::reverse(lb.intra.begin(), lb.intra.end());
            std::reverse(lb.intra_res.begin(), lb.intra_res.end());
            for (uint64_t m = lb.touched; m; m &= m - 1) { const int b = __builtin_ctzll(m); std::reverse(lb.tu[b >> 4][b & 15].begin(), lb.tu[b >> 4][b & 15].end()); }
        }
        if (!lb.intra.empty() && c->cips.empty() && g_intra_pack) {
            // the packed kernel (ohevc_dev_intra_recon_sorted) wants the level's blocks by size - N lanes serve an N x N block, so the blocks of
            // a wavefront must be of one size (ohevc_intra_sort_level, host_jobs.hip: stable, the residual records ride with their jobs)
            const bool paired = lb.intra_res.size() == lb.intra.size();
            int32_t cnt[4];
            const int src = ohevc_intra_sort_level(lb.intra.data(), paired ? lb.intra_res.data() : nullptr, (int)lb.intra.size(), cnt);
            if (src != OHEVC_OK) return src;
            for (int k = 0; k < 4; k++) loff[l].count[k] = cnt[k];
            loff[l].packed = true;
        }
        if (!lb.intra.empty()) loff[l].intra = stage_put(parts, total, lb.intra.data(), lb.intra.size() * sizeof(ohevc_intra_job));
        if (!lb.intra_res.empty()) loff[l].intra_res = stage_put(parts, total, lb.intra_res.data(), lb.intra_res.size() * sizeof(ohevc_tu_job));
        bool first = true;
        for (uint64_t m = lb.touched; m; m &= m - 1) {
            const int b = __builtin_ctzll(m);
            auto &v = lb.tu[b >> 4][b & 15];
            if (b == 3 * 16 + OHEVC_TU_IDCT) sort_tile_bin(v);
            const size_t o = stage_put(parts, total, v.data(), v.size() * sizeof(ohevc_tu_job));
            if (first) { loff[l].tu_first = o; first = false; }
        }
    }
    // runs of consecutive NARROW levels (at most g_intra_chain_waves wavefronts of the packed kernel each): one ohevc_dev_intra_chain launch per run.  A
    // level with residual bins of its own (blocks whose residual does not ride with the prediction) can only END a run: its bins launch
    // behind it and in front of the next level.
    std::vector<ohevc_intra_chain_level> &chain = c->chain_tab;
    chain.clear();
    c->chain_first.assign((size_t)c->max_level + 2, 0);
    c->chain_len.assign((size_t)c->max_level + 2, 0);
    if (g_intra_chain && c->frame_mode != 1) {
        auto waves_of = [&](int k) { return (loff[k].count[0] + 15) / 16 + (loff[k].count[1] + 7) / 8 + (loff[k].count[2] + 3) / 4 + (loff[k].count[3] + 1) / 2; };
        const int max_waves = std::min(g_intra_chain_waves, ohevc_intra_chain_max_waves());
        auto narrow = [&](int k) { return loff[k].packed && waves_of(k) > 0 && waves_of(k) <= max_waves; };
        for (int l = 1; l <= c->max_level;) {
            if (!narrow(l)) { l++; continue; }
            int e = l;
            while (e + 1 <= c->max_level && c->levels[e].touched == 0 && narrow(e + 1) && e - l + 1 < ohevc_intra_chain_max_levels()) e++;
            if (e - l + 1 >= g_intra_chain_min_run) {       // (a run costs two launches - the transforms, then the chain: short ones go level by level)
                c->chain_first[l] = (int)chain.size();
                c->chain_len[l] = e - l + 1;
                for (int k = l; k <= e; k++) {
                    ohevc_intra_chain_level cl = {};
                    for (int q = 0; q < 4; q++) {
                        cl.njobs[q] = loff[k].count[q];
                        cl.first_wave[q + 1] = cl.first_wave[q] + (loff[k].count[q] + (16 >> q) - 1) / (16 >> q);
                    }
                    cl.jobs_off16 = (uint32_t)(loff[k].intra / 16);
                    cl.res_off16 = c->levels[k].intra_res.size() == c->levels[k].intra.size() ? (uint32_t)(loff[k].intra_res / 16) : 0xffffffffu;
                    chain.push_back(cl);
                }
            }
            l = e + 1;
        }
    }
    const size_t off_chain = chain.empty() ? 0 : stage_put(parts, total, chain.data(), chain.size() * sizeof(ohevc_intra_chain_level));
    // levels >= 1 run as ONE launch (ohevc_dev_levels): phases in execution order, job offsets relative to the first
    // staged intra / residual array of level 1 (arrays are 256-byte = 16-job aligned, so offsets are whole jobs)
    std::vector<ohevc_level_phase> &phases = c->phases;
    std::vector<uint32_t> &need = c->need;
    phases.clear(); need.clear();
    size_t intra_base = 0, tu_base = 0;
    bool have_intra_base = false, have_tu_base = false;
    int total_wgs = 0;
    if (c->frame_mode == 1) {
        for (int l = 1; l <= c->max_level; l++) {
            LevelBins &lb = c->levels[l];
            if (!lb.intra.empty()) {
                if (!have_intra_base) { intra_base = loff[l].intra; have_intra_base = true; }
                ohevc_level_phase ph = {};
                ph.first_wg = total_wgs; ph.step = (int32_t)need.size(); ph.type = 0;
                ph.first_job = (int32_t)((loff[l].intra - intra_base) / sizeof(ohevc_intra_job)); ph.njobs = (int32_t)lb.intra.size();
                const int w = ohevc_level_phase_workgroups(0, 0, 0, ph.njobs);
                total_wgs += w; need.push_back((uint32_t)w); phases.push_back(ph);
            }
            if (lb.touched) {
                if (!have_tu_base) { tu_base = loff[l].tu_first; have_tu_base = true; }
                size_t o = loff[l].tu_first;
                uint32_t wsum = 0;
                for (uint64_t m = lb.touched; m; m &= m - 1) {
                    const int b = __builtin_ctzll(m);
                    const auto &v = lb.tu[b >> 4][b & 15];
                    ohevc_level_phase ph = {};
                    ph.first_wg = total_wgs; ph.step = (int32_t)need.size(); ph.type = 1;
                    ph.first_job = (int32_t)((o - tu_base) / sizeof(ohevc_tu_job)); ph.njobs = (int32_t)v.size();
                    ph.log2_size = (b >> 4) + 2; ph.kind = b & 15;
                    const int w = ohevc_level_phase_workgroups(1, ph.log2_size, ph.kind, ph.njobs);
                    total_wgs += w; wsum += (uint32_t)w; phases.push_back(ph);
                    o += (v.size() * sizeof(ohevc_tu_job) + 255) & ~(size_t)255;
                }
                need.push_back(wsum);
            }
        }
    }
    c->sync_zero.assign(need.size() + 2, 0u);
    const size_t off_phases = phases.empty() ? 0 : stage_put(parts, total, phases.data(), phases.size() * sizeof(ohevc_level_phase));
    const size_t off_need = phases.empty() ? 0 : stage_put(parts, total, need.data(), need.size() * sizeof(uint32_t));
    const size_t off_sync = phases.empty() ? 0 : stage_put(parts, total, c->sync_zero.data(), c->sync_zero.size() * sizeof(uint32_t));
    const size_t off_coeffs = c->coeffs.empty() ? 0 : stage_put(parts, total, c->coeffs.data(), c->coeffs.size() * sizeof(int16_t));
    const size_t off_expand = c->expand.empty() ? 0 : stage_put(parts, total, c->expand.data(), c->expand.size() * sizeof(ohevc_expand_rec));
    const size_t off_cips = c->cips.empty() ? 0 : stage_put(parts, total, c->cips.data(), c->cips.size() * sizeof(ohevc_intra_cip));
    // CTB executor: tasks, operation words, the jobs they index, zeroed sync words (home XCD, ticket, one done flag per task)
    const bool ctbs = !c->ctb_tasks.empty();
    c->ctb_sync_zero.assign(ctbs ? 2 * c->ctb_tasks.size() + 2 : 0, 0u);
    const size_t off_ct = ctbs ? stage_put(parts, total, c->ctb_tasks.data(), c->ctb_tasks.size() * sizeof(ohevc_ctb_task)) : 0;
    const size_t off_co = ctbs ? stage_put(parts, total, c->ctb_opwords.data(), c->ctb_opwords.size() * sizeof(uint32_t)) : 0;
    const size_t off_ci = ctbs && !c->ctb_intra.empty() ? stage_put(parts, total, c->ctb_intra.data(), c->ctb_intra.size() * sizeof(ohevc_intra_job)) : 0;
    const size_t off_cu = ctbs && !c->ctb_tu.empty() ? stage_put(parts, total, c->ctb_tu.data(), c->ctb_tu.size() * sizeof(ohevc_tu_job)) : 0;
    const size_t off_cs = ctbs ? stage_put(parts, total, c->ctb_sync_zero.data(), c->ctb_sync_zero.size() * sizeof(uint32_t)) : 0;
    if (!c->tail_parts.empty()) {                       // the frame end's filter maps ride along (frame_end_impl)
        c->tail_base = total;
        parts.insert(parts.end(), c->tail_parts.begin(), c->tail_parts.end());
        total += c->tail_total;
        c->tail_parts.clear();
    }
    // Early flushes of one picture (ohevc_frame_flush_intra) alternate between the two staging / device buffer pairs: with one pair the
    // parsing thread stood still in every flush until the device had finished the chain of the flush before (the arena that chain reads
    // is what this upload overwrites, and the staging copy waits for the upload in front of it).  (Giving the second pair to the filter maps
    // of the frame end instead tied in two rounds of A/B runs and is gone.)
    const int rlane = c->recon_lane;
    c->recon_lane ^= 1;
    c->last_recon_lane = rlane;
    if ((rc = upload_jobs(c, parts, total, rlane)) != OHEVC_OK) return rc;
    struct CallTime { ohevc_ctx *c; int k; double t0; ~CallTime() { if (g_trace_timing) c->t_part[k] += now_s() - t0; } } call_time{c, 1, g_trace_timing ? now_s() : 0};
    unsigned char *base = static_cast<unsigned char *>(c->d_jobs[rlane].p);
    // the dense arena the kernels index, rebuilt on the device from the compact stream that crossed the bus (one buffer per upload lane, like
    // the job arrays: an early flush's chain may still read the other one)
    const int16_t *d_coeffs = nullptr;
    if (!c->expand.empty()) {
        DevBuf &dn = c->d_dense[rlane];
        const size_t need_bytes = (size_t)c->dense * sizeof(int16_t);
        if (need_bytes > dn.cap) {
            OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));     // launches of an earlier frame may still read the old buffer
            if ((rc = dn.reserve(need_bytes)) != OHEVC_OK) return rc;
        }
        rc = ohevc_dev_expand_coeffs(reinterpret_cast<const int16_t *>(base + off_coeffs), reinterpret_cast<const ohevc_expand_rec *>(base + off_expand),
                                     (int)c->expand.size(), static_cast<int16_t *>(dn.p), c->stream);
        if (rc != OHEVC_OK) return rc;
        c->stats.launches++;
        d_coeffs = static_cast<const int16_t *>(dn.p);
    }

    // ---- phase 1: inter prediction (reads other pictures only) -- hevc.c:2430-2464
    if (!c->mc.empty()) {
        rc = ohevc_dev_mc_batch_bounded(p->planes, static_cast<const ohevc_plane *>(c->d_table.p), kMaxPics, p->bd,        // the recorder cuts into tiles
                                        reinterpret_cast<const ohevc_mc_job *>(base + off_mc), (int)c->mc.size(), 16, 16, c->stream);
        if (rc != OHEVC_OK) return rc;
        c->stats.launches++;
    }
    if (!c->mc_small.empty()) {
        // (the slots in use, not the table's capacity: the kernel keeps the plane records of that many pictures in LDS)
        rc = ohevc_dev_mc_batch_small(p->planes, static_cast<const ohevc_plane *>(c->d_table.p), std::max(1, std::min(kMaxPics, (int)c->store->npics)), p->bd,
                                      reinterpret_cast<const ohevc_mc_job *>(base + off_mcs), (int)c->mc_small.size(), c->stream);
        if (rc != OHEVC_OK) return rc;
        c->stats.launches++;
    }
    if (c->keep_motion_l2 && (!c->mc.empty() || !c->mc_small.empty())) {      // what the boundary strengths will need of these jobs (ohevc_dev_motion_grid)
        int gw, gh;
        if ((rc = motion_grid_ready(c, p, gw, gh)) != OHEVC_OK) return rc;
        rc = ohevc_dev_motion_grid2(reinterpret_cast<const ohevc_mc_job *>(base + off_mc), (int)c->mc.size(), reinterpret_cast<const ohevc_mc_job *>(base + off_mcs),
                                    (int)c->mc_small.size(), static_cast<uint8_t *>(c->d_grid.p), gw, gh, c->keep_motion_l2, c->stream);
        if (rc != OHEVC_OK) return rc;
        c->stats.launches++;
    }
    // ---- phase 2..: level 0 = residuals of inter blocks; level L >= 1 = intra prediction of level L, then its residuals
    const int max_level = c->max_level;
    const int last_separate = phases.empty() ? max_level : 0;      // level 0 (residuals of inter blocks) keeps its own wide launch
    if (!phases.empty()) {
        // (issued after level 0 below; prepared here to keep the offsets together)
    }
    const bool trace_launches = ohevc::config().trace_launches;
    int n_lv_intra = 0, n_lv_tu = 0;
    int chained_until = -1;                          // levels up to here had their intra blocks done by a chain launch
    for (int level = 0; level <= last_separate; level++) {
        LevelBins &lb = c->levels[level];
        if (level < (int)c->chain_len.size() && c->chain_len[level] > 0) {
            rc = ohevc_dev_intra_chain(p->planes, p->bd, base, reinterpret_cast<const ohevc_intra_chain_level *>(base + off_chain) + c->chain_first[level],
                                       c->chain_len[level], d_coeffs, c->stream);
            if (rc != OHEVC_OK) return rc;
            c->stats.launches++;
            chained_until = level + c->chain_len[level] - 1;
        }
        if (!lb.intra.empty() && level > chained_until) {
            if (loff[level].packed)
                rc = ohevc_dev_intra_recon_sorted(p->planes, p->bd, reinterpret_cast<const ohevc_intra_job *>(base + loff[level].intra),
                                                  lb.intra_res.size() == lb.intra.size() ? reinterpret_cast<const ohevc_tu_job *>(base + loff[level].intra_res) : nullptr,
                                                  loff[level].count, d_coeffs, c->stream);
            else if (lb.intra_res.size() == lb.intra.size())
                rc = ohevc_dev_intra_recon_batch(p->planes, p->bd, reinterpret_cast<const ohevc_intra_job *>(base + loff[level].intra),
                                                 reinterpret_cast<const ohevc_tu_job *>(base + loff[level].intra_res), (int)lb.intra.size(),
                                                 c->cips.empty() ? nullptr : reinterpret_cast<const ohevc_intra_cip *>(base + off_cips), d_coeffs, c->stream);
            else
                rc = ohevc_dev_intra_batch_cip(p->planes, p->bd, reinterpret_cast<const ohevc_intra_job *>(base + loff[level].intra),
                                               (int)lb.intra.size(),
                                               c->cips.empty() ? nullptr : reinterpret_cast<const ohevc_intra_cip *>(base + off_cips), c->stream);
            if (rc != OHEVC_OK) return rc;
            c->stats.launches++;
        }
        // every (size, kind) bin of this level in ONE launch (bins are staged back to back, 256-byte = 16-job aligned)
        ohevc_tu_segment segs[40];
        int nsegs = 0;
        size_t job_off = 0;                          // in jobs, relative to the level's first bin
        auto flush_segs = [&]() -> int {
            if (!nsegs) return OHEVC_OK;
            int r = ohevc_dev_tu_multi(p->planes, p->bd, segs, nsegs, reinterpret_cast<const ohevc_tu_job *>(base + loff[level].tu_first), d_coeffs, c->stream);
            c->stats.launches++;
            nsegs = 0;
            return r;
        };
        for (uint64_t m = lb.touched; m; m &= m - 1) {
            const int b = __builtin_ctzll(m);
            const auto &v = lb.tu[b >> 4][b & 15];
            if (nsegs == 40 && (rc = flush_segs()) != OHEVC_OK) return rc;      // 4 sizes x 11 kinds can exceed one table
            ohevc_tu_segment &sg = segs[nsegs++];
            sg.log2_size = (b >> 4) + 2; sg.kind = b & 15;
            sg.first_job = (int32_t)job_off;
            sg.njobs = (int32_t)v.size();
            job_off += ((v.size() * sizeof(ohevc_tu_job) + 255) & ~(size_t)255) / sizeof(ohevc_tu_job);
        }
        if (trace_launches && level > 0) { n_lv_intra += !lb.intra.empty(); n_lv_tu += lb.touched != 0; }
        if ((rc = flush_segs()) != OHEVC_OK) return rc;
    }
    if (trace_launches)
        fprintf(stderr, "launches: target %d levels %d: %d prediction(+residual) launches, %d residual launches of unpaired blocks; mc %d+%d jobs, level-0 residual bins %d\n",
                c->cur, max_level, n_lv_intra, n_lv_tu, (int)c->mc.size(), (int)c->mc_small.size(), max_level >= 0 ? __builtin_popcountll(c->levels[0].touched) : 0);
    if (!phases.empty()) {
        rc = ohevc_dev_levels(p->planes, p->bd, reinterpret_cast<const ohevc_level_phase *>(base + off_phases), (int)phases.size(), total_wgs,
                              reinterpret_cast<uint32_t *>(base + off_sync), reinterpret_cast<const uint32_t *>(base + off_need),
                              reinterpret_cast<const ohevc_intra_job *>(base + intra_base),
                              c->cips.empty() ? nullptr : reinterpret_cast<const ohevc_intra_cip *>(base + off_cips),
                              reinterpret_cast<const ohevc_tu_job *>(base + tu_base), d_coeffs, c->stream);
        if (rc != OHEVC_OK) return rc;
        c->stats.launches++;
    }
    if (ctbs) {      // every intra-coded block of the picture: one launch, behind inter prediction and the residuals of inter blocks
        rc = ohevc_dev_ctbs(p->planes, p->bd, p->cfi, c->log2_ctb, reinterpret_cast<const ohevc_ctb_task *>(base + off_ct), (int)c->ctb_tasks.size(),
                            reinterpret_cast<const uint32_t *>(base + off_co), reinterpret_cast<const ohevc_intra_job *>(base + off_ci),
                            c->cips.empty() ? nullptr : reinterpret_cast<const ohevc_intra_cip *>(base + off_cips),
                            reinterpret_cast<const ohevc_tu_job *>(base + off_cu), d_coeffs, reinterpret_cast<uint32_t *>(base + off_cs), c->stream);
        if (rc != OHEVC_OK) return rc;
        c->stats.launches++;
        static const bool ctb_debug = ohevc::config().ctb_debug;
        if (ctb_debug) {       // diagnosis: wait (bounded) for the launch, then look at the sync words: home / ticket / flags / progress
            const double t0 = now_s();
            hipError_t q;
            while ((q = hipStreamQuery(c->stream)) == hipErrorNotReady && now_s() - t0 < 5.0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            std::vector<uint32_t> sw(2 * c->ctb_tasks.size() + 2);
            hipStream_t side;
            (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
            (void)hipMemcpyAsync(sw.data(), base + off_cs, sw.size() * 4, hipMemcpyDeviceToHost, side);
            (void)hipStreamSynchronize(side);
            (void)hipStreamDestroy(side);
            const size_t nt = c->ctb_tasks.size();
            size_t done = 0;
            for (size_t i = 0; i < nt; i++) done += sw[2 + i] != 0;
            if (q == hipErrorNotReady || (sw[0] & 0x80000000u) || done != nt) {
                fprintf(stderr, "ctb debug: launch %s after %.1f s: home 0x%x ticket %u done %zu / %zu tasks\n", q == hipErrorNotReady ? "STILL RUNNING" : "finished",
                        now_s() - t0, sw[0], sw[1], done, nt);
                for (size_t i = 0; i < nt && i < 200; i++)
                    if (!sw[2 + i])
                        fprintf(stderr, "  task %zu ctb (%u,%u) ops %u deps %d %d %d %d state 0x%x\n", i, c->ctb_tasks[i].cx, c->ctb_tasks[i].cy, c->ctb_tasks[i].nops,
                                c->ctb_tasks[i].dep[0], c->ctb_tasks[i].dep[1], c->ctb_tasks[i].dep[2], c->ctb_tasks[i].dep[3], sw[2 + nt + i]);
                if (q == hipErrorNotReady) { fflush(stderr); abort(); }
            }
        }
    }
    c->stats.intra_levels = std::max(c->stats.intra_levels, std::max(max_level, 0));
    clear_recorded(c);
    // (the frame end's filter kernels read the same lane: it records the event again behind them)
    OHEVC_HIP_TRY(hipEventRecord(c->lane_done[rlane], c->stream));
    c->lane_done_pending[rlane] = true;
    return OHEVC_OK;
}

// An intra-coded picture is one long dependency chain on the device (a 1080p picture: ~1000 levels, milliseconds) and, in a random-access
// stream, what every other picture of its GOP waits for.  Its blocks do not have to wait for the picture's last CTU to be parsed: whatever
// has been recorded can run while the host parses on (ohevc_frame_reconstruct may be called any number of times per frame; the levels of a
// later call start behind the earlier call's in the stream).  The front end calls this at the end of every CTU row; it hands the recorded
// work over when the frame has no inter prediction so far (a frame with references would have to wait here, on the parsing thread, for
// their frame ends to be issued - that wait belongs at the frame end) and at least min_pending_kib KiB of records and coefficients are
// waiting.  The price of a flush: the blocks of a band form a chain of their own - a picture's dependency levels run along diagonals
// through ALL of its CTU rows (level ~ x / 4 + 2 y / 4), a band of h rows still has W / 4 + 2 h / 4 of them - so cutting a 1080p intra
// picture into its 17 CTU rows makes ~8700 levels out of ~1000 and the device falls behind the parser instead of keeping up with it
// (measured: profiles/r4n_*; 1414 against 1576 fps with 16 frame threads on the encoder-like stream).  It pays where a picture's parsing
// takes much longer than its chain: dense residuals, 4K / 8K pictures - hence a threshold in bytes, not in rows.
extern "C" int ohevc_frame_flush_intra(ohevc_ctx *c, int min_pending_kib)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    if (c->dry || c->concurrent || c->flush_closed) return OHEVC_OK;
    if (!c->mc.empty() || !c->mc_small.empty()) { c->flush_closed = true; return OHEVC_OK; }
    // what an upload of the recorded work would carry: the coefficient arena and ~32 bytes of records per intra block
    // (the coefficients counted as they will lie in the dense arena, not as the compact stream that crosses the bus: the threshold stands for
    // "enough device work to start on", and was tuned - 2 or 3 hand-overs per intra picture - on dense bytes)
    const size_t pending = (size_t)c->dense * sizeof(int16_t) + (size_t)(c->nstat[2] - c->flushed_intra) * 32;
    if (pending < (size_t)min_pending_kib * 1024) return OHEVC_OK;
    c->flushed_intra = c->nstat[2];
    return ohevc_frame_reconstruct(c);
}

// A frame that cannot be completed must still be PUBLISHED: other decoding threads block (for up to 20 s each) until the frame_end of
// every picture they reference has been issued.  Marks the picture complete-and-failed; dependents return OHEVC_ERR_STATE at once.
extern "C" int ohevc_frame_abort(ohevc_ctx *c)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    clear_recorded(c);
    c->dbk_v.clear(); c->dbk_h.clear(); c->dbk_blob.clear(); c->sao.clear(); c->sao_lagged = false; c->bypass.clear();
    {
        std::lock_guard<std::mutex> g(c->store->m);
        if (p->gen == c->my_gen) { p->failed = true; p->end_issued = true; }
        if ((int32_t)(c->my_gen - p->issued_gen) > 0) p->issued_gen = c->my_gen;
    }
    c->store->cv.notify_all();
    return OHEVC_OK;
}

static int frame_end_impl(ohevc_ctx *c);
extern "C" int ohevc_frame_end(ohevc_ctx *c)
{
    OHEVC_REQUIRE(get_pic(c, c ? c->cur : -1) != nullptr, "no frame begun");
    const int rc = frame_end_impl(c);
    if (rc != OHEVC_OK) {
        char keep[512];
        snprintf(keep, sizeof(keep), "%s", ohevc_last_error());
        ohevc_frame_abort(c);
        set_error("%s", keep);
    }
    return rc;
}

static int frame_end_impl(ohevc_ctx *c)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    const double t_begin = g_trace_timing ? now_s() : 0;
    struct Acc { ohevc_ctx *c; double t0; ~Acc() { if (g_trace_timing) { c->t_issue += now_s() - t0; c->n_frames++; } } } acc{c, t_begin};
    // The filter maps and records are staged FIRST and handed to ohevc_frame_reconstruct, which puts them behind its job arrays in ONE host-to-
    // device copy (one staging pass, one copy, one event less per picture).  (It does not shorten the frame end: about a dozen launches into a
    // picture some call blocks until the device has caught up - whichever call it is, with or without a second copy in front of it - so the
    // calls of a frame end take as long as the device needs for its work, OHEVC_TRACE=timing, profiles/r04o_* - r04q_*.)
    merge_side(c);                                      // (slice threads: their recorders hold filter records too; the arrays must not move after this)
    if (c->sao.empty()) c->bypass.clear();
    const bool filters = !c->dry && (!c->dbk_v.empty() || !c->dbk_h.empty() || !c->sao.empty() || !c->dbk_blob.empty());
    std::vector<std::pair<const void *, size_t>> parts;
    size_t total = 0, off_m = 0, off_bsc = 0, off_v = 0, off_h = 0, off_s = 0, off_b = 0;
    bool dev_bs = false;
    int n_sao_wide = 0;
    if (filters) {
        off_m = c->dbk_blob.empty() ? 0 : stage_put(parts, total, c->dbk_blob.data(), c->dbk_blob.size());
        dev_bs = c->have_bs && !c->dbk_blob.empty();
        off_bsc = dev_bs && !c->bs_calls.empty() ? stage_put(parts, total, c->bs_calls.data(), c->bs_calls.size() * sizeof(ohevc_bs_call)) : 0;
        off_v = c->dbk_v.empty() ? 0 : stage_put(parts, total, c->dbk_v.data(), c->dbk_v.size() * sizeof(ohevc_dbk_job));
        off_h = c->dbk_h.empty() ? 0 : stage_put(parts, total, c->dbk_h.data(), c->dbk_h.size() * sizeof(ohevc_dbk_job));
        // the blocks the wide SAO kernel takes first (ohevc_dev_sao_batch_sorted); SAO blocks of a picture are independent of each other.
        // (The deblocked copy they read is allocated like the picture: same alignment, same pitch.)
        n_sao_wide = (int)(std::stable_partition(c->sao.begin(), c->sao.end(), [&](const ohevc_sao_job &j) {
                               return ohevc_sao_job_is_wide(&j, p->planes, p->planes, p->bd) != 0; }) - c->sao.begin());
        off_s = c->sao.empty() ? 0 : stage_put(parts, total, c->sao.data(), c->sao.size() * sizeof(ohevc_sao_job));
        off_b = c->bypass.empty() ? 0 : stage_put(parts, total, c->bypass.data(), c->bypass.size());
        c->tail_parts = parts; c->tail_total = total;
    }
    if (ohevc::config().trace_upload)
        fprintf(stderr, "upload: target %d filter maps %zu bs_calls %zu dbk jobs %zu sao %zu bypass %zu\n", c->cur, c->dbk_blob.size(), c->bs_calls.size() * sizeof(ohevc_bs_call),
                (c->dbk_v.size() + c->dbk_h.size()) * sizeof(ohevc_dbk_job), c->sao.size() * sizeof(ohevc_sao_job), c->bypass.size());
    c->tail_base = SIZE_MAX;
    int rc = ohevc_frame_reconstruct(c);
    c->tail_parts.clear();
    if (rc != OHEVC_OK) return rc;
    if (c->dry) {
        if (g_sink) g_sink(g_sink_user, c, 1);
        c->dbk_v.clear(); c->dbk_h.clear(); c->sao.clear(); c->sao_lagged = false;
        c->dbk_blob.clear(); c->bs_calls.clear(); c->have_bs = false; c->bypass.clear();      // (ohevc_debug_set_record_only(2): the device forms, dropped)
    }
    if (filters) {
        int lane = c->last_recon_lane;                  // (the maps rode with the job arrays of the reconstruction above)
        size_t tail = c->tail_base;
        if (tail == SIZE_MAX) {                           // nothing was reconstructed: an upload of their own
            lane = c->recon_lane;
            c->recon_lane ^= 1;
            if ((rc = upload_jobs(c, parts, total, lane)) != OHEVC_OK) return rc;
            tail = 0;
        }
        c->tail_base = SIZE_MAX;
        struct CallTime { ohevc_ctx *c; int k; double t0; ~CallTime() { if (g_trace_timing) c->t_part[k] += now_s() - t0; } } call_time{c, 2, g_trace_timing ? now_s() : 0};
        double t_lap = g_trace_timing ? now_s() : 0;
        auto lap = [&](int k) { if (g_trace_timing) { const double t = now_s(); c->t_f[k] += t - t_lap; t_lap = t; } };
        unsigned char *base = static_cast<unsigned char *>(c->d_jobs[lane].p) + tail;
        ohevc_dbk_maps dm = c->dbk_maps;                  // offsets -> device addresses
        if (!c->dbk_blob.empty()) {
            dm.vertical_bs = base + off_m + reinterpret_cast<uintptr_t>(c->dbk_maps.vertical_bs);
            dm.horizontal_bs = base + off_m + reinterpret_cast<uintptr_t>(c->dbk_maps.horizontal_bs);
            dm.qp_y_tab = reinterpret_cast<const int8_t *>(base + off_m + reinterpret_cast<uintptr_t>(c->dbk_maps.qp_y_tab));
            dm.deblock = reinterpret_cast<const int8_t *>(base + off_m + reinterpret_cast<uintptr_t>(c->dbk_maps.deblock));
            dm.is_pcm = c->dbk_maps.is_pcm ? base + off_m + reinterpret_cast<uintptr_t>(c->dbk_maps.is_pcm) : nullptr;      // its offset is never 0
        }
        if (dev_bs) {          // boundary strengths from the motion field, on the device (hevc_filter.c:805-941)
            const size_t bs_h = (size_t)(dm.height >> 2), n_v = ((size_t)dm.bs_width * (bs_h + 8) + 255) & ~(size_t)255, n_h = (((size_t)dm.bs_width + 8) * bs_h + 255) & ~(size_t)255;
            uint8_t *vbs = nullptr;
            if (!c->bs_maps.mvf) {                            // the grid's clearing covered the room behind it (motion_grid_ready)
                int gw0, gh0;
                if ((rc = motion_grid_ready(c, p, gw0, gh0)) != OHEVC_OK) return rc;
                if (n_v + n_h <= c->grid_bs_cap) vbs = static_cast<uint8_t *>(c->d_grid.p) + c->grid_bs_off;
            }
            if (!vbs) {
                if (n_v + n_h > c->d_bs.cap) {
                    OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
                    if ((rc = c->d_bs.reserve(n_v + n_h)) != OHEVC_OK) return rc;
                }
                if ((rc = ohevc_dev_zero(c->d_bs.p, n_v + n_h, c->stream)) != OHEVC_OK) return rc;
                vbs = static_cast<uint8_t *>(c->d_bs.p);
            }
            ohevc_bs_maps bm = c->bs_maps;
            if (c->bs_maps.mvf) {
                bm.mvf = base + off_m + reinterpret_cast<uintptr_t>(c->bs_maps.mvf);
            } else {                                          // rebuilt from the MC jobs by ohevc_frame_reconstruct
                int gw, gh;
                OHEVC_REQUIRE(c->keep_motion_l2 == bm.log2_min_pu_size, "ohevc_frame_keep_motion was not called for this frame");
                if ((rc = motion_grid_ready(c, p, gw, gh)) != OHEVC_OK) return rc;      // (a picture without inter blocks: cleared here)
                OHEVC_REQUIRE(gw >= bm.min_pu_width && gh >= bm.min_pu_height, "motion grid smaller than the picture's min_pu map");
                bm.mvf = static_cast<const uint8_t *>(c->d_grid.p); bm.min_pu_width = gw;
                bm.mvf_stride = OHEVC_MOTION_GRID_ENTRY; bm.off_mv = 0; bm.off_poc = 8; bm.off_pred_flag = 16; bm.pred_flag_bytes = 4;
            }
            bm.cbf_luma = base + off_m + reinterpret_cast<uintptr_t>(c->bs_maps.cbf_luma);
            uint8_t *hbs = vbs + n_v;
            if ((rc = ohevc_dev_boundary_strengths(&bm, reinterpret_cast<const ohevc_bs_call *>(base + off_bsc), (int)c->bs_calls.size(), vbs, hbs, c->stream)) != OHEVC_OK) return rc;
            if (!c->bs_calls.empty()) c->stats.launches++;
            dm.vertical_bs = vbs; dm.horizontal_bs = hbs;
        }
        lap(0);
        // all vertical edges, then all horizontal edges: deblocking_filter_CTB, hevc_filter.c:385-580
        if (!c->dbk_blob.empty()) {
            if ((rc = ohevc_dev_deblock_maps(p->planes, p->bd, &dm, 1, c->stream)) != OHEVC_OK) return rc;
            c->stats.launches++;
        }
        if (!c->dbk_v.empty()) {
            if ((rc = ohevc_dev_deblock_batch(p->planes, p->bd, reinterpret_cast<const ohevc_dbk_job *>(base + off_v), (int)c->dbk_v.size(), c->stream)) != OHEVC_OK) return rc;
            c->stats.launches++;
        }
        lap(1);
        const bool lagged = c->sao_lagged && !c->sao.empty() && !c->dbk_h.empty();
        auto ensure_like = [&](Picture &q) -> int {
            if (q.used && q.w == p->w && q.h == p->h && q.cfi == p->cfi && q.bd == p->bd) return OHEVC_OK;
            int r;
            if (q.used) {                 // another geometry: launches that read the old copy may still be in flight
                OHEVC_HIP_TRY(hipStreamSynchronize(c->stream));
                if ((r = free_picture(q, false, c->store.get())) != OHEVC_OK) return r;
            }
            return alloc_picture(q, p->w, p->h, p->cfi, p->bd, false, c->store.get(), c->stream);
        };
        if (lagged) {          // the state the reference's early copy saw (ohevc_hip.h, OHEVC_SAO_LAG_*): chroma only
            if ((rc = ensure_like(c->lag)) != OHEVC_OK) return rc;
            const size_t chroma = (size_t)p->planes[1].stride * p->planes[1].height + (size_t)p->planes[2].stride * p->planes[2].height;
            if (p->single && c->lag.single) {
                if ((rc = ohevc_dev_copy(c->lag.planes[1].data, p->planes[1].data, chroma, c->stream)) != OHEVC_OK) return rc;
            } else {
                for (int i = 1; i < 3; i++)
                    OHEVC_HIP_TRY(hipMemcpy2DAsync(c->lag.planes[i].data, (size_t)c->lag.planes[i].stride, p->planes[i].data, (size_t)p->planes[i].stride,
                                                   (size_t)p->planes[i].width * (p->bd > 8 ? 2 : 1), (size_t)p->planes[i].height, hipMemcpyDeviceToDevice, c->stream));
            }
        }
        if (!c->dbk_blob.empty()) {
            if ((rc = ohevc_dev_deblock_maps(p->planes, p->bd, &dm, 0, c->stream)) != OHEVC_OK) return rc;
            c->stats.launches++;
        }
        if (!c->dbk_h.empty()) {
            if ((rc = ohevc_dev_deblock_batch(p->planes, p->bd, reinterpret_cast<const ohevc_dbk_job *>(base + off_h), (int)c->dbk_h.size(), c->stream)) != OHEVC_OK) return rc;
            c->stats.launches++;
        }
        lap(2);
        if (!c->sao.empty()) {
            // SAO reads a deblocked copy and writes the picture: sao_filter_CTB, hevc_filter.c:269-315
            if ((rc = ensure_like(c->twin)) != OHEVC_OK) return rc;
            if (p->single && c->twin.single) {              // (same geometry: ensure_like)
                size_t all = 0;
                for (int i = 0; i < 3; i++) all += (size_t)p->planes[i].stride * p->planes[i].height;
                if ((rc = ohevc_dev_copy(c->twin.planes[0].data, p->planes[0].data, all, c->stream)) != OHEVC_OK) return rc;
            } else {
                // (an adopted picture - ohevc_pic_adopt - keeps its owner's pitch, the copy has the store's: row by row then)
                for (int i = 0; i < 3; i++)
                    OHEVC_HIP_TRY(hipMemcpy2DAsync(c->twin.planes[i].data, (size_t)c->twin.planes[i].stride, p->planes[i].data, (size_t)p->planes[i].stride,
                                                   (size_t)p->planes[i].width * (p->bd > 8 ? 2 : 1), (size_t)p->planes[i].height, hipMemcpyDeviceToDevice, c->stream));
            }
            lap(3);
            ohevc_plane lagp[3] = {c->twin.planes[0], lagged ? c->lag.planes[1] : c->twin.planes[1], lagged ? c->lag.planes[2] : c->twin.planes[2]};
            ohevc_sao_bypass bp = {};                     // restore_tqb_pixels, hevc_filter.c:163-193
            if (!c->bypass.empty()) {
                bp.map = base + off_b; bp.stride = c->bypass_w; bp.log2_min_pu_size = c->bypass_l2;
                bp.chroma_hshift = p->cfi == 1 || p->cfi == 2; bp.chroma_vshift = p->cfi == 1; bp.exact_reference = c->bypass_exact;
            }
            if ((rc = ohevc_dev_sao_batch_sorted(p->planes, c->twin.planes, lagp, p->bd, reinterpret_cast<const ohevc_sao_job *>(base + off_s), n_sao_wide, (int)c->sao.size() - n_sao_wide, &bp, c->stream)) != OHEVC_OK) return rc;
            c->stats.launches += (n_sao_wide > 0) + (n_sao_wide < (int)c->sao.size());
            lap(4);
        }
        c->dbk_v.clear(); c->dbk_h.clear(); c->dbk_blob.clear(); c->sao.clear(); c->sao_lagged = false; c->bypass.clear();
        c->bs_calls.clear(); c->have_bs = false;
    }
    if (!c->dry) {
        // publish: this picture is reconstructed once `ev` fires; the references were read until then
        if (!c->target_guarded && (rc = guard_pictures(c, c->cur)) != OHEVC_OK) return rc;     // filter-only frames
        hipEvent_t ev = c->ring[c->ring_next];
        if (g_trace_order) fprintf(stderr, "order: ctx %p ends target %d event %p\n", (void *)c, c->cur, (void *)ev);
        c->ring_next = (c->ring_next + 1) % 16;
        OHEVC_HIP_TRY(hipEventRecord(ev, c->stream));
        // the filter kernels above read the maps out of one of the two upload lanes (whichever carried them): no upload into either before they are done
        for (int k = 0; k < 2; k++) {
            OHEVC_HIP_TRY(hipEventRecord(c->lane_done[k], c->stream));
            c->lane_done_pending[k] = true;
        }
        {
            std::lock_guard<std::mutex> g(c->store->m);
            p->written = ev;
            for (int r : c->ref_slots) {
                auto &rd = c->store->pics[r].readers;
                if (std::find(rd.begin(), rd.end(), ev) == rd.end()) rd.push_back(ev);
            }
        }
        // a long-chain picture joins the context's public stream again HERE (not at the next frame_begin): a handle cached from
        // ohevc_ctx_stream stays ordered behind every picture, and the copy-back that follows goes out on the public stream
        if (c->stream != c->stream_norm && c->stream_norm && (rc = select_stream(c, false)) != OHEVC_OK) return rc;
    }
    {
        std::lock_guard<std::mutex> g(c->store->m);
        if (p->gen == c->my_gen) p->end_issued = true;      // (else a newer picture has been begun in this slot meanwhile)
        if ((int32_t)(c->my_gen - p->issued_gen) > 0) p->issued_gen = c->my_gen;
    }
    c->store->cv.notify_all();
    c->stats.alg_bytes = c->alg;
    c->stats.n_tu = c->nstat[0]; c->stats.n_mc = c->nstat[1]; c->stats.n_intra = c->nstat[2]; c->stats.n_dbk = c->nstat[3]; c->stats.n_sao = c->nstat[4];
    { std::lock_guard<std::mutex> g(c->stats_m); c->last_stats = c->stats; }
    return OHEVC_OK;
}

// ------------------------------------------------------------------ asynchronous frame ends
// With the reference's frame threads every decoding thread ends its picture itself: stage, upload, ~30-75 launches, copy-back.  Measured
// with 8-16 threads: a frame end that takes 0.6-0.9 ms alone takes 3-7 ms, 2.3 ms of it blocked until the threads decoding
// its REFERENCE pictures have issued theirs (their completion events must exist before this picture's work can be ordered behind them) and
// the rest issuing against the other threads' HIP calls.  The reference's own frame threads never block like that: they wait row by row
// (pthread_frame.c:479-513) and only where a motion vector really points.  The remedy here does not cut pictures into bands - it takes the
// issue out of the decoding threads: ohevc_frame_end_async hands the recorded frame (a swap of vectors) to an executor context of the
// store's ISSUER thread and returns; the decoding thread goes on parsing.  The issuer takes queued frames in an order in which every
// reference picture's frame end has been issued before (never blocks on one: it takes another frame), issues them one after the other -
// no lock contention inside the HIP runtime - on a ring of executor streams, and queues the copy-back into the application's (page-locked)
// planes behind each.  The host meets the device again only where the application takes the picture out: ohevc_pic_wait_host.
static void swap_frame_state(ohevc_ctx &a, ohevc_ctx &b)
{
    std::swap(static_cast<Rec &>(a), static_cast<Rec &>(b));
    a.dbk_blob.swap(b.dbk_blob);
    std::swap(a.dbk_maps, b.dbk_maps);
    a.bypass.swap(b.bypass);
    std::swap(a.bypass_w, b.bypass_w); std::swap(a.bypass_l2, b.bypass_l2); std::swap(a.bypass_exact, b.bypass_exact);
    std::swap(a.cur, b.cur); std::swap(a.frame_mode, b.frame_mode); std::swap(a.log2_ctb, b.log2_ctb);
    std::swap(a.stats, b.stats);
    std::swap(a.my_gen, b.my_gen);
    std::swap(a.bs_maps, b.bs_maps); std::swap(a.have_bs, b.have_bs);
    std::swap(a.keep_motion_l2, b.keep_motion_l2); std::swap(a.grid_zeroed, b.grid_zeroed);      // (d_grid stays: it is only touched on its owner's stream)
}

// the first queued frame whose reference pictures have all had their frame ends issued (or failed) - taken off the queue - or nullptr (is->m held)
static ohevc_ctx *issuer_take_ready_locked(Issuer *is)
{
    PicStore &st = *is->store;
    ohevc_ctx *e = nullptr;
    std::lock_guard<std::mutex> g(st.m);
    for (size_t i = 0; i < is->queue.size() && !e; i++) {
        const ohevc_ctx *q = is->queue[i];
        bool ready = true;
        for (const auto &r : q->async_refs) ready = ready && (int32_t)(st.pics[r.first].issued_gen - r.second) >= 0;
        // ... and no frame submitted earlier still has to read (or write) the memory this one overwrites
        // (a frame submitted earlier may also read THIS frame's picture - its thread finished parsing first: that one waits for us)
        auto blocks = [&](const ohevc_ctx *k) {
            if (k->cur == q->cur) return true;
            for (const auto &r : k->async_refs) if (r.first == q->cur && (int32_t)(r.second - q->my_gen) < 0) return true;
            return false;
        };
        for (size_t k = 0; k < i && ready; k++) ready = !blocks(is->queue[k]);
        for (size_t k = 0; k < is->executing.size() && ready; k++) ready = !blocks(is->executing[k]);      // (other issuing threads)
        if (ready) { e = is->queue[i]; is->queue.erase(is->queue.begin() + (long)i); }
    }
    return e;
}

// issue the frame end of executor context e (taken off the queue; the caller has put it into is->executing and counted it in in_flight)
static void issuer_issue(Issuer *is, ohevc_ctx *e)
{
    PicStore &st = *is->store;
    const double t0 = now_s();
    e->ref_slots.clear();
    e->target_guarded = false;
    int rc = ohevc_frame_end(e);                   // (aborts and publishes the picture as failed on error)
    Picture *p = get_pic(e, e->cur);
    if (p && e->async_host[0]) {
        hipEvent_t ev = nullptr;
        if (rc == OHEVC_OK) {
            ev = e->dl_ring[e->dl_next];
            e->dl_next = (e->dl_next + 1) % 8;
            for (int i = 0; i < 3 && rc == OHEVC_OK; i++) {
                if (!e->async_host[i]) continue;
                const ohevc_plane &pl = p->planes[i];
                if (hipMemcpy2DAsync(e->async_host[i], e->async_stride[i], pl.data, pl.stride, (size_t)pl.width * (p->bd > 8 ? 2 : 1), pl.height,
                                     hipMemcpyDeviceToHost, e->stream) != hipSuccess) { set_error("asynchronous copy-back failed: %s", hipGetErrorString(hipGetLastError())); rc = OHEVC_ERR_HIP; }
            }
            if (rc == OHEVC_OK && hipEventRecord(ev, e->stream) != hipSuccess) rc = OHEVC_ERR_HIP;
        }
        std::lock_guard<std::mutex> g(st.m);
        p->host_copy = rc == OHEVC_OK ? ev : nullptr;
        if (rc != OHEVC_OK) p->failed = true;
        p->host_copy_issued = true;
    }
    st.cv.notify_all();
    {
        std::lock_guard<std::mutex> lk(is->m);
        if (rc != OHEVC_OK && is->error == OHEVC_OK) { is->error = rc; snprintf(is->error_text, sizeof(is->error_text), "%s", ohevc_last_error()); }
        if (e->async_from) {
            ohevc_frame_stats done;
            { std::lock_guard<std::mutex> g(e->stats_m); done = e->last_stats; }
            std::lock_guard<std::mutex> g(e->async_from->stats_m);
            if (e->parked) {                       // a parked frame: its numbers are ADDED to what the recording context reports next (ohevc_frame_get_stats)
                ohevc_frame_stats &a = e->async_from->parked_stats;
                a.launches += done.launches; a.upload_bytes += done.upload_bytes; a.n_tu += done.n_tu; a.n_mc += done.n_mc; a.n_intra += done.n_intra;
                a.n_dbk += done.n_dbk; a.n_sao += done.n_sao; a.alg_bytes += done.alg_bytes; a.intra_levels = std::max(a.intra_levels, done.intra_levels);
            } else {
                e->async_from->last_stats = done;
            }
        }
        e->exec_busy = false;
        is->executing.erase(std::find(is->executing.begin(), is->executing.end(), e));
        is->in_flight--;
        is->busy_s += now_s() - t0;
        is->frames++;
    }
    is->cv.notify_all();
}

static void issuer_run(Issuer *is)
{
    (void)hipSetDevice(is->device);
    for (;;) {
        ohevc_ctx *e = nullptr;
        {
            std::unique_lock<std::mutex> lk(is->m);
            for (;;) {
                if (is->stop && is->queue.empty()) return;
                if ((e = issuer_take_ready_locked(is)) != nullptr) break;
                if (is->queue.empty()) { is->cv.wait(lk); continue; }
                // frames are queued but none is ready: a reference is still being parsed by its thread.  Its submission wakes us; a thread
                // that died would leave us here for ever, so the oldest frame gives up after the reference wait limit
                if (is->cv.wait_for(lk, std::chrono::seconds(g_ref_wait_s)) == std::cv_status::timeout && !is->queue.empty()) {
                    e = is->queue.front(); is->queue.pop_front();
                    e->async_refs.clear();             // frame_end_impl's own wait will fail it with the proper message
                    break;
                }
            }
            is->in_flight++;
            is->executing.push_back(e);
        }
        issuer_issue(is, e);
    }
}

// A thread that has just issued a frame end (or parked one, or waits for one) issues every parked frame that has become ready: the store needs no
// issuer threads of its own for frames parked by ohevc_frame_end_deferred - whoever unblocks a frame runs it.
static void issuer_help(PicStore &st)
{
    Issuer *is = get_issuer(st);
    if (!is) return;
    if (!is->th.empty()) { is->cv.notify_all(); return; }      // the store has issuer threads of its own: they take what has become ready
    for (;;) {
        ohevc_ctx *e;
        {
            std::lock_guard<std::mutex> lk(is->m);
            if (is->queue.empty() || is->stop) return;
            if (!(e = issuer_take_ready_locked(is))) return;
            is->in_flight++;
            is->executing.push_back(e);
        }
        (void)hipSetDevice(is->device);
        issuer_issue(is, e);
    }
}

// wait until every submitted frame end has been issued (not: executed)
static void async_drain(PicStore &st)
{
    Issuer *is = get_issuer(st);
    if (!is) return;
    for (;;) {
        issuer_help(st);                                   // (parked frames have no issuer thread of their own)
        std::unique_lock<std::mutex> lk(is->m);
        if (is->queue.empty() && is->in_flight == 0) return;
        is->cv.wait_for(lk, std::chrono::milliseconds(is->th.empty() ? 1 : 20));
    }
}

static void issuer_shutdown(PicStore &st)
{
    Issuer *is = st.issuer;
    if (!is) return;
    { std::lock_guard<std::mutex> lk(is->m); is->stop = true; }
    is->cv.notify_all();
    for (std::thread &t : is->th) if (t.joinable()) t.join();
    if (g_trace_timing && is->frames)
        fprintf(stderr, "timing: issuer of store %p: %ld frame ends, %.3f ms each\n", (void *)&st, is->frames, 1e3 * is->busy_s / is->frames);
    st.issuer = nullptr;
    std::vector<ohevc_ctx *> execs;
    execs.swap(is->execs);
    delete is;
    for (ohevc_ctx *e : execs) ohevc_ctx_destroy(e);
}

// the reference pictures (slot, version) of the frame recorded in c (the decoder still holds them: the slots name the right versions)
static void frame_refs(const ohevc_ctx *c, const PicStore &st, std::vector<std::pair<int, uint32_t>> &out)
{
    out.clear();
    for (const auto *v : {&c->mc, &c->mc_small})
        for (const ohevc_mc_job &j : *v) {
            const int refs[2] = {j.ref0, (j.flags & OHEVC_MC_BI) ? j.ref1 : -1};
            for (int r : refs) {
                if (r < 0 || r == c->cur) continue;
                bool seen = false;
                for (const auto &a : out) seen = seen || a.first == r;
                if (!seen) out.emplace_back(r, st.pics[r].gen);
            }
        }
}

// hand the frame recorded in c to an executor context on the store's queue.  threads: dedicated issuer threads to start with the store's first
// submission (0: none - parked frames are issued by the threads that unblock them, issuer_help)
static int submit_frame(ohevc_ctx *c, void *const host[3], const ptrdiff_t host_stride[3], bool parked, int threads)
{
    Picture *p = get_pic(c, c->cur);
    PicStore &st = *c->store;
    Issuer *is;
    {
        std::lock_guard<std::mutex> g(st.m);
        if (!st.issuer) {
            Issuer *n = new Issuer();
            n->device = c->device;
            n->store = &st;
            for (int k = 0; k < threads; k++) n->th.emplace_back(issuer_run, n);
            __atomic_store_n(&st.issuer, n, __ATOMIC_RELEASE);
        }
        is = st.issuer;
    }
    // a free executor context (its vectors keep their capacity from picture to picture), or a new one
    ohevc_ctx *e = nullptr;
    {
        std::lock_guard<std::mutex> lk(is->m);
        for (ohevc_ctx *x : is->execs) if (!x->exec_busy) { e = x; break; }
        if (e) e->exec_busy = true;
    }
    if (!e) {
        int rc = ohevc_ctx_create_shared(&e, c->device, c);
        if (rc != OHEVC_OK) return rc;
        e->is_exec = true; e->exec_busy = true;
        for (auto &ev : e->dl_ring) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("event creation failed"); return OHEVC_ERR_HIP; }
        std::lock_guard<std::mutex> lk(is->m);
        is->execs.push_back(e);
    }
    swap_frame_state(*c, *e);
    e->async_from = c;
    e->parked = parked;
    {
        std::lock_guard<std::mutex> g(st.m);                // the decoder still holds this frame's references: their slots name the right versions
        frame_refs(e, st, e->async_refs);
    }
    for (int i = 0; i < 3; i++) { e->async_host[i] = host ? host[i] : nullptr; e->async_stride[i] = host && host_stride ? host_stride[i] : 0; }
    {
        std::lock_guard<std::mutex> g(st.m);
        p->host_copy_issued = !(host && host[0]);
        p->host_copy = nullptr;
    }
    c->stats = ohevc_frame_stats{};
    if (parked) { std::lock_guard<std::mutex> g(c->stats_m); c->last_stats = ohevc_frame_stats{}; }
    {
        std::lock_guard<std::mutex> lk(is->m);
        is->queue.push_back(e);
    }
    is->cv.notify_all();
    return OHEVC_OK;
}

extern "C" int ohevc_frame_end_async(ohevc_ctx *c, void *const host[3], const ptrdiff_t host_stride[3])
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    OHEVC_REQUIRE(!c->is_exec, "executor contexts do not record");
    if (c->dry) {                                       // record-only contexts have nothing to overlap
        int rc = ohevc_frame_end(c);
        return rc;
    }
    OHEVC_REQUIRE(!c->grid_zeroed, "a frame that keeps its motion (ohevc_frame_keep_motion) and was partly reconstructed ends with ohevc_frame_end");
    merge_side(c);                                      // the slice threads of this picture have been joined: fold their recorders in
    return submit_frame(c, host, host_stride, false, std::max(1, ohevc::config().issuer_threads));
}

// ohevc_frame_end that never makes the calling thread WAIT for other threads' frame ends.  The reference's frame threads block only on row progress
// (pthread_frame.c:479-513); this back end needs no reference rows while it parses, but a picture's device work can only be ORDERED behind its
// reference pictures' once their frame ends have been issued (their completion events must exist) - and in a random-access GOP the references are
// the big pictures, still being parsed when the small ones that predict from them are done: at 16 frame threads a decoding thread spent 1-3 ms per
// picture in that wait (OHEVC_TRACE=timing, profiles/r6f_*).  Here: if every reference has been issued, the frame is issued at once, on this
// thread (the common case, and the only one with one decoding thread); if not, the recorded frame is PARKED - swapped into an executor context on
// the store's queue - and the call returns.  Whoever issues the last missing reference issues the parked frame right behind it (issuer_help):
// the work of the issue moves to the thread that made it possible, nobody waits, no extra threads.  A thread that needs the picture - the
// application taking it out, a picture begun in a slot a parked frame still reads - helps and waits (wait_end_issued, settle_slot).
// OFF by default: measured on the device at 16 frame threads (profiles/r6i_*, r6j_*, r6k_*) parking LOSES - encoder-like stream 3440 -> 2940 fps
// steady, 2360-2690 -> 1670-1810 from a cold decoder, with helpers and with 1 / 2 / 4 issuer threads alike: the wait it removes was idle time of a
// thread that had nothing else to do (the decoder hands it its next packet only in decoding order), while a parked frame costs an executor context
// (streams, staging lanes, device buffers: a pool that has to warm up) and moves the issue onto the thread that parses the GOP's big pictures.
static bool g_park_frames = false;         // ohevc_debug_set_park_frames
static std::atomic<long> g_parked_total{0};
extern "C" long ohevc_debug_parked_total(void) { return g_parked_total.load(std::memory_order_relaxed); }      // frames parked so far, process-wide (tests)
extern "C" int ohevc_debug_set_park_frames(int on) { const int prev = g_park_frames; g_park_frames = on != 0; return prev; }
extern "C" int ohevc_frame_end_deferred(ohevc_ctx *c)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr, "no frame begun");
    const int park = c->opt[OHEVC_OPT_PARK_FRAMES] >= 0 ? c->opt[OHEVC_OPT_PARK_FRAMES] : (int)g_park_frames;
    if (c->dry || c->is_exec || c->grid_zeroed || !park) return ohevc_frame_end(c);
    PicStore &st = *c->store;
    merge_side(c);
    bool ready = true;
    {
        std::lock_guard<std::mutex> g(st.m);
        frame_refs(c, st, c->async_refs);
        for (const auto &r : c->async_refs) ready = ready && st.pics[r.first].end_issued;
    }
    c->async_refs.clear();
    if (ready) {
        const int rc = ohevc_frame_end(c);
        issuer_help(st);                                // this picture may be what parked frames were waiting for
        return rc;
    }
    const int rc = submit_frame(c, nullptr, nullptr, true, ohevc::config().park_threads);
    if (rc != OHEVC_OK) return rc;
    c->n_parked++;
    g_parked_total.fetch_add(1, std::memory_order_relaxed);
    issuer_help(st);                                    // (the missing reference may have been issued between the look above and the push)
    return OHEVC_OK;
}

// A picture is about to be begun in (uploaded into, released from) `slot`: frames still on the store's queue that read or write that slot's memory
// must be issued first - their completion events are what orders the new work behind them (a parked frame has returned from its frame end, so the
// decoder may recycle the pictures it read).
static void settle_slot(ohevc_ctx *c, int slot)
{
    PicStore &st = *c->store;
    Issuer *is = get_issuer(st);
    if (!is || c->is_exec) return;
    const double deadline = now_s() + g_ref_wait_s;
    for (;;) {
        bool busy = false;
        {
            std::lock_guard<std::mutex> lk(is->m);
            auto touches = [&](const ohevc_ctx *q) {
                if (q->cur == slot) return true;
                for (const auto &r : q->async_refs) if (r.first == slot) return true;
                return false;
            };
            for (const ohevc_ctx *q : is->queue) busy = busy || touches(q);
            for (const ohevc_ctx *q : is->executing) busy = busy || touches(q);
        }
        if (!busy || now_s() > deadline) return;
        issuer_help(st);
        std::unique_lock<std::mutex> lk(is->m);
        is->cv.wait_for(lk, std::chrono::microseconds(200));
    }
}

// the application takes the picture out: its samples are in the planes given to ohevc_frame_end_async when this returns OHEVC_OK
extern "C" int ohevc_pic_wait_host(ohevc_ctx *c, int slot)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr, "bad picture slot");
    if (c->dry) return OHEVC_OK;
    hipEvent_t ev;
    {
        std::unique_lock<std::mutex> lk(c->store->m);
        if (!c->store->cv.wait_for(lk, std::chrono::seconds(g_ref_wait_s), [&] { return p->host_copy_issued && p->end_issued; })) {
            set_error("picture %d: its frame end was never issued", slot);
            return OHEVC_ERR_STATE;
        }
        if (p->failed) { set_error("picture %d: its frame failed", slot); return OHEVC_ERR_STATE; }
        ev = p->host_copy;
    }
    if (ev) OHEVC_HIP_TRY(hipEventSynchronize(ev));
    return OHEVC_OK;
}

// first failure of an asynchronous frame end since the last call (OHEVC_OK: none); the text goes to ohevc_last_error()
extern "C" int ohevc_ctx_async_status(ohevc_ctx *c)
{
    OHEVC_REQUIRE(c != nullptr, "null context");
    Issuer *is = c->store->issuer;
    if (!is) return OHEVC_OK;
    std::lock_guard<std::mutex> lk(is->m);
    const int rc = is->error;
    if (rc != OHEVC_OK) set_error("%s", is->error_text);
    is->error = OHEVC_OK;
    return rc;
}

// seconds the store's issuer has spent issuing frame ends, and how many (cumulative; for benches)
extern "C" int ohevc_ctx_async_profile(ohevc_ctx *c, double *busy_s, long long *frames)
{
    OHEVC_REQUIRE(c != nullptr && busy_s != nullptr && frames != nullptr, "null argument");
    *busy_s = 0; *frames = 0;
    Issuer *is = c->store->issuer;
    if (!is) return OHEVC_OK;
    std::lock_guard<std::mutex> lk(is->m);
    *busy_s = is->busy_s; *frames = is->frames;
    return OHEVC_OK;
}

extern "C" int ohevc_debug_target(ohevc_ctx *c, int *slot, int *width, int *height, int *cfi, int *bd)
{
    OHEVC_REQUIRE(c != nullptr && c->cur >= 0, "no frame begun");
    if (slot) *slot = c->cur;
    return ohevc_pic_info(c, c->cur, width, height, cfi, bd);
}
extern "C" int ohevc_debug_mc(ohevc_ctx *c, int small, const ohevc_mc_job **jobs, int *n)
{
    OHEVC_REQUIRE(c != nullptr && jobs != nullptr && n != nullptr, "null argument");
    const auto &v = small ? c->mc_small : c->mc;
    *jobs = v.data(); *n = (int)v.size();
    return OHEVC_OK;
}
extern "C" int ohevc_debug_level_count(ohevc_ctx *c) { return c ? c->max_level + 1 : 0; }
extern "C" int ohevc_debug_level_intra(ohevc_ctx *c, int level, const ohevc_intra_job **jobs, int *n)
{
    OHEVC_REQUIRE(c != nullptr && level >= 0 && level <= c->max_level && jobs != nullptr && n != nullptr, "bad level");
    *jobs = c->levels[level].intra.data(); *n = (int)c->levels[level].intra.size();
    return OHEVC_OK;
}
extern "C" int ohevc_debug_level_tu(ohevc_ctx *c, int level, int log2, int kind, const ohevc_tu_job **jobs, int *n)
{
    OHEVC_REQUIRE(c != nullptr && level >= 0 && level <= c->max_level && log2 >= 2 && log2 <= 5 && kind >= 0 && kind < OHEVC_TU_NKINDS &&
                  jobs != nullptr && n != nullptr, "bad bin");
    const auto &v = c->levels[level].tu[log2 - 2][kind];
    *jobs = v.data(); *n = (int)v.size();
    return OHEVC_OK;
}
extern "C" int ohevc_debug_ctbs(ohevc_ctx *c, const ohevc_ctb_task **tasks, int *ntasks, const uint32_t **ops, const ohevc_intra_job **intra_jobs,
                                const ohevc_tu_job **tu_jobs, int *log2_ctb_size)
{
    OHEVC_REQUIRE(c != nullptr && tasks && ntasks && ops && intra_jobs && tu_jobs, "null argument");
    *tasks = c->ctb_tasks.data(); *ntasks = (int)c->ctb_tasks.size(); *ops = c->ctb_opwords.data();
    *intra_jobs = c->ctb_intra.data(); *tu_jobs = c->ctb_tu.data();
    if (log2_ctb_size) *log2_ctb_size = c->log2_ctb;
    return OHEVC_OK;
}
extern "C" int ohevc_debug_arena(ohevc_ctx *c, const int16_t **coeffs, const ohevc_intra_cip **cips)
{
    OHEVC_REQUIRE(c != nullptr, "null context");
    if (coeffs) {              // the DENSE arena the jobs index, rebuilt on the host from the compact stream (what ohevc_dev_expand_coeffs does on the device)
        c->dense_host.assign((size_t)c->dense, (int16_t)0);
        for (const ohevc_expand_rec &e : c->expand) {
            if (e.kind == 0) { memcpy(c->dense_host.data() + e.dst, c->coeffs.data() + e.src, (size_t)e.dims * sizeof(int16_t)); continue; }
            if (e.kind & 0x100u) {                     // sub-block form: the set groups of the record's region, 16 elements each
                const int n = 1 << (e.kind & 0xff), gpr = n >> 2;
                const int16_t *in = c->coeffs.data() + e.src;
                for (int gi = 0; gi < 32; gi++) {
                    if (!(e.dims >> gi & 1u)) continue;
                    const int gy = gi / gpr, gx = gi % gpr;
                    for (int k = 0; k < 4; k++) memcpy(c->dense_host.data() + e.dst + (size_t)(gy * 4 + k) * n + gx * 4, in + 4 * k, 8);
                    in += 16;
                }
                continue;
            }
            const int n = 1 << e.kind, cols = (int)(e.dims & 0xff), rows = (int)(e.dims >> 8);
            for (int y = 0; y < rows; y++) memcpy(c->dense_host.data() + e.dst + (size_t)y * n, c->coeffs.data() + e.src + (size_t)y * cols, (size_t)cols * sizeof(int16_t));
        }
        *coeffs = c->dense_host.data();
    }
    if (cips) *cips = c->cips.data();
    return OHEVC_OK;
}
extern "C" int ohevc_debug_filters(ohevc_ctx *c, const ohevc_dbk_job **v, int *nv, const ohevc_dbk_job **h, int *nh, const ohevc_sao_job **sao, int *ns,
                                   ohevc_sao_bypass *bypass)
{
    Picture *p = get_pic(c, c ? c->cur : -1);
    OHEVC_REQUIRE(p != nullptr && v && nv && h && nh && sao && ns, "bad argument");
    *v = c->dbk_v.data(); *nv = (int)c->dbk_v.size();
    *h = c->dbk_h.data(); *nh = (int)c->dbk_h.size();
    *sao = c->sao.data(); *ns = (int)c->sao.size();
    if (bypass) {
        *bypass = ohevc_sao_bypass{};
        if (!c->bypass.empty()) {
            bypass->map = c->bypass.data(); bypass->stride = c->bypass_w; bypass->log2_min_pu_size = c->bypass_l2;
            bypass->chroma_hshift = p->cfi == 1 || p->cfi == 2; bypass->chroma_vshift = p->cfi == 1; bypass->exact_reference = c->bypass_exact;
        }
    }
    return OHEVC_OK;
}
extern "C" int ohevc_debug_wait_picture(ohevc_ctx *c, int slot)
{
    Picture *p = get_pic(c, slot);
    OHEVC_REQUIRE(p != nullptr, "bad picture slot");
    std::unique_lock<std::mutex> lk(c->store->m);
    if (!wait_end_issued(c, *p, lk)) {
        set_error("picture %d was never completed by its decoding thread", slot);
        return OHEVC_ERR_STATE;
    }
    return OHEVC_OK;
}

extern "C" int ohevc_frame_get_stats(ohevc_ctx *c, ohevc_frame_stats *out)
{
    OHEVC_REQUIRE(c != nullptr && out != nullptr, "bad argument");
    std::lock_guard<std::mutex> g(c->stats_m);
    *out = c->last_stats;
    // frames this context parked that have been issued since the last call: their numbers are reported with this one (sums over a run stay exact)
    const ohevc_frame_stats &a = c->parked_stats;
    out->launches += a.launches; out->upload_bytes += a.upload_bytes; out->n_tu += a.n_tu; out->n_mc += a.n_mc; out->n_intra += a.n_intra;
    out->n_dbk += a.n_dbk; out->n_sao += a.n_sao; out->alg_bytes += a.alg_bytes; out->intra_levels = std::max(out->intra_levels, a.intra_levels);
    c->parked_stats = ohevc_frame_stats{};
    return OHEVC_OK;
}
