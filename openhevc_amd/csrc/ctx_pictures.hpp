// ctx_pictures.hpp -- part of ctx.hip (ONE translation unit: included by it in this order, never compiled alone): pictures of the store: allocation, adoption, upload / copy-back (waited for or queued), page-locked host memory, export / import between processes, SHVC up-sampling.
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
