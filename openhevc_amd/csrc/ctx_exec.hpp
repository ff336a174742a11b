// ctx_exec.hpp -- part of ctx.hip (ONE translation unit: included by it in this order, never compiled alone): the executor: staging and upload of a picture's job arrays, ordering against other pictures (guard_pictures), ohevc_frame_reconstruct (levels, chains, CTB tasks), early flushes, abort.
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
