// ctx_debug.hpp -- part of ctx.hip (ONE translation unit: included by it in this order, never compiled alone): inspection entry points (ohevc_debug.h) and per-frame statistics.
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
