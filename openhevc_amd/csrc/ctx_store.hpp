// ctx_store.hpp -- part of ctx.hip (ONE translation unit: included by it in this order, never compiled alone): the device picture store, the context (streams, staging lanes, device buffers), its options and its life cycle.
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
