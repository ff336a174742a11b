// ctx_frame_end.hpp -- part of ctx.hip (ONE translation unit: included by it in this order, never compiled alone): the frame end: in-loop filters and publication of the picture (frame_end_impl), the issuer of asynchronous / parked frame ends, waiting for a picture on the host.
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
