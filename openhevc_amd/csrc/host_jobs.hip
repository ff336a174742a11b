// host_jobs.hip -- host-side (CPU) job construction helpers of the C ABI; no device code.
#include <algorithm>
#include <string.h>
#include <vector>
#include "common.hpp"

namespace {
// z-scan order of a minimum transform block inside its CTB; -1 for the row/column just outside
// (the reference's CTB-local MinTbAddrZs table, hevc_ps.c:2551-2567)
int zscan_in_ctb(int x, int y, int bits)
{
    if (x < 0 || y < 0) return -1;
    int v = 0;
    for (int i = 0; i < bits; i++)
        v |= (((x >> i) & 1) << (2 * i)) | (((y >> i) & 1) << (2 * i + 1));
    return v;
}
}  // namespace

static int make_job_common(const ohevc_intra_geom *g, int lpu, const uint8_t *pf, ptrdiff_t pf_stride, int intra_value,
                           int x0, int y0, int log2_size, int c_idx, int mode,
                           int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                           ohevc_intra_job *out, ohevc_intra_cip *cip);

extern "C" int ohevc_intra_make_job(const ohevc_intra_geom *g, int x0, int y0, int log2_size, int c_idx, int mode,
                                    int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                                    ohevc_intra_job *out)
{
    using namespace ohevc;
    OHEVC_REQUIRE(g != nullptr, "null argument");
    OHEVC_REQUIRE(!g->constrained_intra_pred, "constrained_intra_pred streams need ohevc_intra_make_job_cip");
    return make_job_common(g, 2, nullptr, 0, 0, x0, y0, log2_size, c_idx, mode, cand_bottom_left, cand_left, cand_up_left, cand_up,
                           cand_up_right, out, nullptr);
}

extern "C" int ohevc_intra_make_job_cip(const ohevc_intra_geom *g, int log2_min_pu_size, const uint8_t *pred_flag,
                                        ptrdiff_t pred_flag_stride, int intra_value, int x0, int y0, int log2_size, int c_idx, int mode,
                                        int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                                        ohevc_intra_job *out, ohevc_intra_cip *cip)
{
    using namespace ohevc;
    OHEVC_REQUIRE(g != nullptr, "null argument");
    OHEVC_REQUIRE(!g->constrained_intra_pred || (pred_flag != nullptr && cip != nullptr && log2_min_pu_size >= 2 && log2_min_pu_size <= 5),
                  "constrained_intra_pred needs the prediction-mode map");
    return make_job_common(g, log2_min_pu_size, pred_flag, pred_flag_stride, intra_value, x0, y0, log2_size, c_idx, mode,
                           cand_bottom_left, cand_left, cand_up_left, cand_up, cand_up_right, out, cip);
}

static int make_job_common(const ohevc_intra_geom *g, int lpu, const uint8_t *pf, ptrdiff_t pf_stride, int intra_value,
                           int x0, int y0, int log2_size, int c_idx, int mode,
                           int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                           ohevc_intra_job *out, ohevc_intra_cip *cip)
{
    using namespace ohevc;
    OHEVC_REQUIRE(g != nullptr && out != nullptr, "null argument");
    OHEVC_REQUIRE(log2_size >= 2 && log2_size <= 5, "log2_size must be 2..5");
    OHEVC_REQUIRE(c_idx >= 0 && c_idx <= 2 && mode >= 0 && mode <= 34, "c_idx / mode");
    const int cfi = g->chroma_format_idc;
    const int hs = c_idx ? (cfi == 1 || cfi == 2) : 0, vs = c_idx ? (cfi == 1) : 0;
    const int n = 1 << log2_size, nlh = n << hs, nlv = n << vs;
    const int bits = g->log2_ctb_size - g->log2_min_tb_size, mask = (1 << bits) - 1;
    const int x_tb = (x0 >> g->log2_min_tb_size) & mask, y_tb = (y0 >> g->log2_min_tb_size) & mask;
    const int cur = zscan_in_ctb(x_tb, y_tb, bits);
    // hevcpred_template.c:105-109: "ahead" neighbours must already be decoded in z-scan order
    bool bl = cand_bottom_left && cur > zscan_in_ctb(x_tb - 1, (y_tb + (nlv >> g->log2_min_tb_size)) & mask, bits);
    bool ur = cand_up_right && cur > zscan_in_ctb((x_tb + (nlh >> g->log2_min_tb_size)) & mask, y_tb - 1, bits);
    const bool cipmode = g->constrained_intra_pred != 0;
    const int pu_w = (g->width + (1 << lpu) - 1) >> lpu, pu_h = (g->height + (1 << lpu) - 1) >> lpu;
    // minimum PU (xp, yp) intra?  Outside the picture the reference reads undefined memory; it counts as "not intra" here.
    auto ispu = [&](int xp, int yp) -> bool {
        return xp >= 0 && yp >= 0 && xp < pu_w && yp < pu_h && pf[((ptrdiff_t)xp + (ptrdiff_t)yp * pu_w) * pf_stride] == (uint8_t)intra_value;
    };
    if (cipmode) {                                   // hevcpred_template.c:116-159: inter-coded neighbours do not count
        int spu_v = nlv >> lpu, spu_h = nlh >> lpu;
        const bool edge_x = !(x0 & ((1 << lpu) - 1)), edge_y = !(y0 & ((1 << lpu) - 1));
        if (!spu_h) spu_h++;
        if (bl && edge_x) {
            const int xl = (x0 - 1) >> lpu, yb = (y0 + nlv) >> lpu, mx = std::min(spu_v, pu_h - yb);
            bl = false;
            for (int i = 0; i < mx; i += 2) bl = bl || ispu(xl, yb + i);
        }
        if (cand_left && edge_x) {
            const int xl = (x0 - 1) >> lpu, yl = y0 >> lpu, mx = std::min(spu_v, pu_h - yl);
            cand_left = 0;
            for (int i = 0; i < mx; i += 2) cand_left |= ispu(xl, yl + i);
        }
        if (cand_up_left) cand_up_left = ispu((x0 - 1) >> lpu, (y0 - 1) >> lpu);
        if (cand_up && edge_y) {
            const int xt = x0 >> lpu, yt = (y0 - 1) >> lpu, mx = std::min(spu_h, pu_w - xt);
            cand_up = 0;
            for (int i = 0; i < mx; i += 2) cand_up |= ispu(xt + i, yt);
        }
        if (ur && edge_y) {
            const int yt = (y0 - 1) >> lpu, xr = (x0 + nlh) >> lpu, mx = std::min(spu_h, pu_w - xr);
            ur = false;
            for (int i = 0; i < mx; i += 2) ur = ur || ispu(xr + i, yt);
        }
    }
    // :111-114: neighbour runs clipped to the picture
    const int y_end = y0 + 2 * nlv < g->height ? y0 + 2 * nlv : g->height;
    const int x_end = x0 + 2 * nlh < g->width ? x0 + 2 * nlh : g->width;
    int bl_size = (y_end - (y0 + nlv)) >> vs, tr_size = (x_end - (x0 + nlh)) >> hs;
    if (bl_size < 0) bl_size = 0;
    if (tr_size < 0) tr_size = 0;
    memset(out, 0, sizeof(*out));
    out->x = (uint16_t)(x0 >> hs);
    out->y = (uint16_t)(y0 >> vs);
    out->plane = (uint8_t)c_idx;
    out->log2_size = (uint8_t)log2_size;
    out->mode = (uint8_t)mode;
    unsigned f = 0;
    if (bl) f |= OHEVC_INTRA_BOTTOM_LEFT;
    if (cand_left) f |= OHEVC_INTRA_LEFT;
    if (cand_up_left) f |= OHEVC_INTRA_UP_LEFT;
    if (cand_up) f |= OHEVC_INTRA_UP;
    if (ur) f |= OHEVC_INTRA_UP_RIGHT;
    // :289: smoothing only for luma, or chroma in 4:4:4, and not when disabled by the RExt flag
    if (g->intra_smoothing_disabled || !(c_idx == 0 || cfi == 3)) f |= OHEVC_INTRA_NO_SMOOTHING;
    if (g->strong_intra_smoothing && c_idx == 0) f |= OHEVC_INTRA_STRONG;
    if (c_idx == 0) f |= OHEVC_INTRA_LUMA_EDGE;
    out->flags = (uint8_t)f;
    out->bottom_left_size = (uint8_t)bl_size;
    out->top_right_size = (uint8_t)tr_size;
    out->log2_ctb_size = (uint8_t)(g->log2_ctb_size >= 4 && g->log2_ctb_size <= 6 ? g->log2_ctb_size : 0);
    if (cipmode) {
        out->flags2 |= OHEVC_INTRA2_CIP;
        memset(cip, 0, sizeof(*cip));
        auto isi = [&](int x, int y) -> bool {           // IS_INTRA(x, y) in block-relative plane samples (:33-40)
            return ispu((x0 + (int)((unsigned)x << hs)) >> lpu, (y0 + (int)((unsigned)y << vs)) >> lpu);
        };
        for (int k = -1; k < 64; k++) {
            if (isi(k, -1)) cip->top_bits[(k + 1) >> 3] |= (uint8_t)(1u << ((k + 1) & 7));
            if (isi(-1, k)) cip->left_bits[(k + 1) >> 3] |= (uint8_t)(1u << ((k + 1) & 7));
        }
        // scan limits, :187-198 (they depend on the re-derived availability)
        int smx = x0 + ((2 * n) << hs) < g->width ? 2 * n : (g->width - x0) >> hs;
        int smy = y0 + ((2 * n) << vs) < g->height ? 2 * n : (g->height - y0) >> vs;
        if (!ur) smx = x0 + (n << hs) < g->width ? n : (g->width - x0) >> hs;
        if (!bl) smy = y0 + (n << vs) < g->height ? n : (g->height - y0) >> vs;
        cip->size_max_x = (uint8_t)smx; cip->size_max_y = (uint8_t)smy;
        cip->x0_nonzero = x0 != 0; cip->y0_nonzero = y0 != 0;
    }
    return OHEVC_OK;
}

// The order the packed intra kernel wants a dependency level's blocks in (intra_pack.hpp: N lanes per N x N block, 16 / 8 / 4 / 2 blocks per
// wavefront): by size - a wavefront's blocks must be of one size - and otherwise as recorded, i.e. in decoding order: neighbours in the picture
// stay neighbours in a wavefront.
// (Round 5 also sorted by prediction mode inside a size, so that the blocks sharing a wavefront take one path through the predictors of
// hevcpred_template.c:359-537.  Measured on 130 000 independent blocks, all 35 modes: 17 % fewer vector instructions - and 2.4x the time for
// 4x4 blocks (0.061 -> 0.147 ms), because blocks of one mode lie all over the picture: L2 misses 1.25 M -> 4.9 M per launch, fabric read requests
// 1.09 M -> 3.44 M, SQ_WAIT_ANY 62 M -> 151 M wave cycles (profiles/r5a_sq_counters_intra_pack_before.txt, r5g_sq_counters_intra_pack_mode_sorted.txt).
// The kernel waits for memory 60 % of its wave cycles; its instruction stream is not what bounds it.  Size only, again.)
// Stable counting sort of the jobs and of the residual records riding with them (residuals may be NULL); count_by_size[k] = blocks of (4 << k).
extern "C" int ohevc_intra_sort_level(ohevc_intra_job *jobs, ohevc_tu_job *residuals, int n, int32_t count_by_size[4])
{
    OHEVC_REQUIRE(n >= 0 && (n == 0 || jobs != nullptr) && count_by_size != nullptr, "bad argument");
    int cnt[4] = {};
    for (int k = 0; k < n; k++) {
        OHEVC_REQUIRE(jobs[k].log2_size >= 2 && jobs[k].log2_size <= 5 && jobs[k].mode <= 34, "bad intra job");
        cnt[jobs[k].log2_size - 2]++;
    }
    for (int sz = 0; sz < 4; sz++) count_by_size[sz] = cnt[sz];
    bool sorted = true;
    for (int k = 1; k < n && sorted; k++) sorted = jobs[k].log2_size >= jobs[k - 1].log2_size;
    if (sorted) return OHEVC_OK;
    int pos[4];
    for (int b = 0, at = 0; b < 4; b++) { pos[b] = at; at += cnt[b]; }
    static thread_local std::vector<ohevc_intra_job> tj;
    static thread_local std::vector<ohevc_tu_job> tr;
    tj.resize((size_t)n);
    if (residuals) tr.resize((size_t)n);
    for (int k = 0; k < n; k++) {
        const int d = pos[jobs[k].log2_size - 2]++;
        tj[(size_t)d] = jobs[k];
        if (residuals) tr[(size_t)d] = residuals[k];
    }
    memcpy(jobs, tj.data(), (size_t)n * sizeof(*jobs));
    if (residuals) memcpy(residuals, tr.data(), (size_t)n * sizeof(*residuals));
    return OHEVC_OK;
}
