// ctx_record.hpp -- part of ctx.hip (ONE translation unit: included by it in this order, never compiled alone): the recorder: ohevc_frame_begin and the ohevc_rec_* calls (job records, dependency levels of the intra-coded blocks, the compact coefficient arena, filter maps).
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
