// filters_host.hip -- the in-loop filter DRIVERS of the reference, in bulk (host code; SURVEY.md 8f-3, first half).
//
// deblocking_filter_CTB and sao_filter_CTB (hevc_filter.c:197-581) walk every CTB with a one-CTB lag while the picture is being
// parsed, derive tc / beta / pcm flags per 8-sample edge and the SAO edge / border flags per CTB, and make one table call per edge
// and per CTB plane -- about a third of all table calls of a picture, plus host-pixel copies (copy_CTB, the sao_frame ring) that
// nobody reads behind recording tables.  Everything they need is final once the picture is parsed: the boundary-strength maps,
// qp_y_tab, the per-CTB deblocking offsets and SAO parameters, the slice / tile maps.  ohevc_tables_derive_filters reads those
// arrays ONCE at the frame end and records the same job set in one tight loop: no per-edge pointer translation, no calls, and the
// reference's drivers can be skipped altogether.  On a device the maps themselves travel and the kernels derive
// everything (ohevc_dev_deblock_maps); this file's per-edge derivation serves record-only contexts (host-logic tests through the software
// executor: tests/test_stream_cpu.py runs both paths) and the filter-lag replay of 16x16-CTB streams.  It is the kernels' closed form, edge
// by edge - which offsets an edge next to a CTB boundary gets included - because the job set must be the one the drivers would have produced.
#include <algorithm>
#include <unordered_map>
#include <utility>
#include <vector>
#include "common.hpp"
#include "ohevc_tables.h"
#include "ohevc_ctx.h"
#include "ohevc_debug.h"

namespace {

// H.265 table 8-12 as hevc_filter.c:50-60 spells it
const uint8_t kTc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                          5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
const uint8_t kBeta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28,
                            30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
inline int clip(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

struct Ctx {
    const ohevc_filter_maps &m;
    ohevc_ctx *ctx;
    int rc = OHEVC_OK;
    std::vector<ohevc_dbk_job> edges;   // handed to the recorder in one call
    // filter-lag emulation (16x16 CTBs, ohevc_hip.h OHEVC_SAO_LAG_*): the order in which the reference's drivers would have made
    // their calls, kept exactly like the recording slots keep it (tables.hip)
    bool lag = false;
    uint32_t seq = 0;
    std::unordered_map<uint32_t, uint32_t> h_edge_seq;
    std::vector<std::pair<ohevc_sao_job, uint32_t>> held_sao;
    int hs, vs;                    // chroma subsampling shifts
    int qpy(int x, int y) const { return m.qp_y_tab[(x >> m.log2_min_cb_size) + (y >> m.log2_min_cb_size) * m.min_cb_width]; }    // get_qPy, :144-150
    int pcm(int x, int y) const                                    // get_pcm, :325-338
    {
        if (x < 0 || y < 0) return 2;
        const int xp = x >> m.log2_min_pu_size, yp = y >> m.log2_min_pu_size;
        if (xp >= m.min_pu_width || yp >= m.min_pu_height) return 2;
        return m.is_pcm[yp * m.min_pu_width + xp];
    }
    int tc_luma(int qp, int bs, int tc_offset) const { return kTc[clip(qp + 2 * (bs - 1) + (tc_offset >> 1 << 1), 0, 53)]; }             // TC_CALC, :340-343
    int tc_chroma(int qp_y, int c_idx, int tc_offset) const        // chroma_tc, :62-89
    {
        static const int qp_c[] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };
        const int qp_i = clip(qp_y + (c_idx == 1 ? m.cb_qp_offset : m.cr_qp_offset), 0, 57);
        int qp;
        if (m.chroma_format_idc == 1) qp = qp_i < 30 ? qp_i : qp_i > 43 ? qp_i - 6 : qp_c[qp_i - 30];
        else                          qp = clip(qp_i, 0, 51);
        return kTc[clip(qp + 2 + tc_offset, 0, 53)];
    }
    void edge(int plane, int x, int y, bool vertical, int beta, int tc0, int tc1, const int *no_p, const int *no_q)
    {
        ohevc_dbk_job j = {};
        j.x = (uint16_t)x; j.y = (uint16_t)y; j.plane = (uint8_t)plane; j.beta = (uint8_t)beta;
        j.tc[0] = (int16_t)tc0; j.tc[1] = (int16_t)tc1;
        j.flags = (uint8_t)((vertical ? OHEVC_DBK_VERTICAL_EDGE : 0) | (no_p[0] ? OHEVC_DBK_NO_P0 : 0) | (no_p[1] ? OHEVC_DBK_NO_P1 : 0) |
                            (no_q[0] ? OHEVC_DBK_NO_Q0 : 0) | (no_q[1] ? OHEVC_DBK_NO_Q1 : 0));
        edges.push_back(j);
        if (lag && !vertical && plane > 0) h_edge_seq[((uint32_t)plane << 30) | ((uint32_t)y << 15) | (uint32_t)x] = ++seq;
    }

    // One 8-sample edge of the 8x8 luma grid / the (8h)x(8v) chroma grid, its first sample at luma position (x, y): the same closed form as
    // deblock_maps_kernel (filter_kernels.hip) - which neighbour CTB's offsets an edge next to a CTB boundary gets follows from where the
    // reference's loops start and stop (hevc_filter.c:385-579):
    //   vertical edges          beta and tc offset of the CTB the edge lies in
    //   horizontal luma edge    beta offset of the CTB holding x, tc offset of the CTB holding x + 8
    //   horizontal chroma edge  segment 0: tc offset of the CTB holding x, segment 1: of the CTB holding x + 8h   (last column: clamped)
    void edge_at(bool chroma, bool vertical, int x, int y)
    {
        const int h = 1 << hs, v = 1 << vs, step = chroma ? (vertical ? 4 * v : 4 * h) : 4;
        const int sx = vertical ? 0 : step, sy = vertical ? step : 0;              // from segment 0 to segment 1
        const int px = vertical ? x - 1 : x, py = vertical ? y : y - 1;            // the P side
        const uint8_t *bsm = vertical ? m.vertical_bs : m.horizontal_bs;
        const int bs[2] = { bsm[(x + y * m.bs_width) >> 2], bsm[((x + sx) + (y + sy) * m.bs_width) >> 2] };
        if (chroma ? !(bs[0] == 2 || bs[1] == 2) : !(bs[0] || bs[1])) return;
        int no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
        if (m.pcm_or_bypass)
            for (int k = 0; k < 2; k++) { no_p[k] = pcm(px + k * sx, py + k * sy); no_q[k] = pcm(x + k * sx, y + k * sy); }
        const int log2_ctb = m.log2_ctb_size, ctb_w = (m.width + (1 << log2_ctb) - 1) >> log2_ctb;
        auto offset = [&](int xx, int which) {                                   // slice_beta_offset / slice_tc_offset of the CTB holding (xx, y)
            const int cx = std::min(xx >> log2_ctb, ctb_w - 1);
            return (int)m.deblock[(size_t)(cx + (y >> log2_ctb) * ctb_w) * m.deblock_stride + which];
        };
        auto qp_at = [&](int k) { return (qpy(px + k * sx, py + k * sy) + qpy(x + k * sx, y + k * sy) + 1) >> 1; };
        if (!chroma) {                                                            // both segments share segment 0's QP (:398, :497)
            const int qp = qp_at(0), tco = offset(vertical ? x : x + 8, 1);
            edge(0, x, y, vertical, kBeta[clip(qp + offset(x, 0), 0, 51)], bs[0] ? tc_luma(qp, bs[0], tco) : 0, bs[1] ? tc_luma(qp, bs[1], tco) : 0, no_p, no_q);
        } else {                                                                  // bS 2 only, a QP per segment (:440-441, :549-550)
            const int tco[2] = { offset(x, 1), offset(vertical ? x : x + 8 * h, 1) };
            const int qp[2] = { bs[0] == 2 ? qp_at(0) : 0, bs[1] == 2 ? qp_at(1) : 0 };
            for (int c = 1; c <= 2; c++)
                edge(c, x >> hs, y >> vs, vertical, 0, bs[0] == 2 ? tc_chroma(qp[0], c, tco[0]) : 0, bs[1] == 2 ? tc_chroma(qp[1], c, tco[1]) : 0, no_p, no_q);
        }
    }

    // The edges one call of deblocking_filter_CTB (hevc_filter.c:345-581) covers - only the filter-lag replay needs to know which call an
    // edge belongs to, the executor takes a picture's edges in any order: the vertical edges inside the CTB; the horizontal edges of a run that
    // begins 8 (chroma: 8h) samples inside the CTB to the left and ends as far short of the next one (the last column runs to the picture edge).
    void deblock_ctb(int x0, int y0)
    {
        const int ctb_size = 1 << m.log2_ctb_size, x_end = std::min(x0 + ctb_size, m.width), y_end = std::min(y0 + ctb_size, m.height);
        for (int chroma = 0; chroma <= (m.chroma_format_idc ? 1 : 0); chroma++) {
            const int gx = chroma ? 8 << hs : 8, gy = chroma ? 8 << vs : 8;
            for (int y = y0; y < y_end; y += gy)
                for (int x = std::max(x0, gx); x < x_end; x += gx) edge_at(chroma != 0, true, x, y);
        }
        for (int chroma = 0; chroma <= (m.chroma_format_idc ? 1 : 0); chroma++) {
            const int gx = chroma ? 8 << hs : 8, gy = chroma ? 8 << vs : 8;
            const int run_begin = x0 ? x0 - gx : 0, run_end = x_end == m.width ? m.width : x_end - gx;
            for (int y = std::max(y0, gy); y < y_end; y += gy)
                for (int x = run_begin; x < run_end; x += gx) edge_at(chroma != 0, false, x, y);
        }
    }

    // sao_filter_CTB, hevc_filter.c:197-322: what the table calls receive (the host copies around them have no counterpart)
    void sao_ctb(int x, int y)
    {
        const int ctb_size = 1 << m.log2_ctb_size, ctb_w = (m.width + ctb_size - 1) >> m.log2_ctb_size, ctb_h = (m.height + ctb_size - 1) >> m.log2_ctb_size;
        const int x_ctb = x >> m.log2_ctb_size, y_ctb = y >> m.log2_ctb_size, rs = y_ctb * ctb_w + x_ctb;
        const ohevc_SAOParams &sao = m.sao[rs];
        const int ts = m.ctb_addr_rs_to_ts[rs];
        const bool lfase = m.filter_slice_edges[rs] != 0, no_tile_filter = m.tiles_enabled && !m.loop_filter_across_tiles;
        const bool restore = no_tile_filter || !lfase;
        const bool e0 = x_ctb == 0, e1 = y_ctb == 0, e2 = x_ctb == ctb_w - 1, e3 = y_ctb == ctb_h - 1;
        bool ve[2] = {}, he[2] = {}, de[4] = {}, lt = false, rt = false, ut = false, bt = false;
        auto slice = [&](int dx, int dy) { return m.tab_slice_address[rs + dx + dy * ctb_w]; };
        auto tile = [&](int dx, int dy) { return m.tile_id[m.ctb_addr_rs_to_ts[rs + dx + dy * ctb_w]]; };
        if (restore) {
            if (!e0) { lt = no_tile_filter && m.tile_id[ts] != tile(-1, 0); ve[0] = (!lfase && slice(0, 0) != slice(-1, 0)) || lt; }
            if (!e2) { rt = no_tile_filter && m.tile_id[ts] != tile(1, 0);  ve[1] = (!lfase && slice(0, 0) != slice(1, 0)) || rt; }
            if (!e1) { ut = no_tile_filter && m.tile_id[ts] != tile(0, -1); he[0] = (!lfase && slice(0, 0) != slice(0, -1)) || ut; }
            if (!e3) { bt = no_tile_filter && m.tile_id[ts] != tile(0, 1);  he[1] = (!lfase && slice(0, 0) != slice(0, 1)) || bt; }
            if (!e0 && !e1) de[0] = (!lfase && slice(0, 0) != slice(-1, -1)) || lt || ut;
            if (!e1 && !e2) de[1] = (!lfase && slice(0, 0) != slice(1, -1)) || rt || ut;
            if (!e2 && !e3) de[2] = (!lfase && slice(0, 0) != slice(1, 1)) || rt || bt;
            if (!e0 && !e3) de[3] = (!lfase && slice(0, 0) != slice(-1, 1)) || lt || bt;
        }
        for (int c = 0; c < (m.chroma_format_idc ? 3 : 1); c++) {
            const int type = sao.type_idx[c];
            if (type != 1 && type != 2) continue;                  // SAO_BAND = 1, SAO_EDGE = 2 (hevc.h:498-503)
            const int chs = c ? hs : 0, cvs = c ? vs : 0;
            const int xc = x >> chs, yc = y >> cvs;
            ohevc_sao_job j = {};
            j.x = (uint16_t)xc; j.y = (uint16_t)yc; j.plane = (uint8_t)c;
            j.w = (uint16_t)std::min(ctb_size >> chs, (m.width >> chs) - xc);
            j.h = (uint16_t)std::min(ctb_size >> cvs, (m.height >> cvs) - yc);
            j.type = (uint8_t)(type == 1 ? OHEVC_SAO_BAND : OHEVC_SAO_EDGE);
            j.klass = type == 1 ? sao.band_position[c] : sao.eo_class[c];
            j.borders = (uint8_t)((e0 ? 1 : 0) | (e1 ? 2 : 0) | (e2 ? 4 : 0) | (e3 ? 8 : 0));
            if (type == 2) {
                j.restore = restore ? 1 : 0;
                if (restore)
                    j.edges = (uint8_t)((ve[0] ? 1 : 0) | (ve[1] ? 2 : 0) | (he[0] ? 4 : 0) | (he[1] ? 8 : 0) | (de[0] ? 16 : 0) | (de[1] ? 32 : 0) |
                                        (de[2] ? 64 : 0) | (de[3] ? 128 : 0));
            }
            for (int k = 0; k < 5; k++) j.offset_val[k] = sao.offset_val[c][k];
            if (lag && c > 0) { held_sao.emplace_back(j, ++seq); continue; }      // flags depend on calls still to come
            const int r = ohevc_rec_sao(ctx, &j);
            if (r != OHEVC_OK && rc == OHEVC_OK) rc = r;
        }
    }

    // ff_hevc_hls_filter / ff_hevc_hls_filters, hevc_filter.c:1027-1064: which CTBs the decoding of CTB (x, y) releases
    void hls_filter(int x, int y)
    {
        const int ctb_size = 1 << m.log2_ctb_size;
        deblock_ctb(x, y);
        if (!m.sao_enabled) return;
        const bool x_end = x >= m.width - ctb_size, y_end = y >= m.height - ctb_size;
        if (y && x) sao_ctb(x - ctb_size, y - ctb_size);
        if (x && y_end) sao_ctb(x - ctb_size, y);
        if (y && x_end) sao_ctb(x, y - ctb_size);
        if (x_end && y_end) sao_ctb(x, y);
    }
    void hls_filters(int x, int y)
    {
        const int ctb_size = 1 << m.log2_ctb_size;
        const bool x_end = x >= m.width - ctb_size, y_end = y >= m.height - ctb_size;
        if (y && x) hls_filter(x - ctb_size, y - ctb_size);
        if (y && x_end) hls_filter(x, y - ctb_size);
        if (x && y_end) hls_filter(x - ctb_size, y);
    }
    // a held SAO job saw, in the reference, the samples right of its block BEFORE a horizontal edge through them was filtered iff
    // that edge's call came after the SAO call (same rule as ohevc_tables_end_frame, tables.hip)
    void release_held_sao()
    {
        for (auto &held : held_sao) {
            ohevc_sao_job &j = held.first;
            const int xr = j.x + j.w, plane_w = m.width >> hs;
            if (xr < plane_w) {
                auto later = [&](int y) {
                    auto it = h_edge_seq.find(((uint32_t)j.plane << 30) | ((uint32_t)y << 15) | (uint32_t)xr);
                    return it != h_edge_seq.end() && it->second > held.second;
                };
                j.quirks = (uint8_t)((later(j.y + j.h) ? OHEVC_SAO_LAG_BELOW : 0) | (later(j.y) ? OHEVC_SAO_LAG_ABOVE : 0) |
                                     ((j.h > 8 && later(j.y + 8)) ? OHEVC_SAO_LAG_MID : 0));
            }
            const int r = ohevc_rec_sao(ctx, &j);
            if (r != OHEVC_OK && rc == OHEVC_OK) rc = r;
        }
        held_sao.clear();
    }
};

bool g_filters_on_device = true;
// boundary strengths on the device: 0 no (the reference's function runs on the host), 1 from the uploaded motion field, 2 from the motion
// the picture's own MC jobs carry (ohevc_frame_keep_motion: nothing extra travels)
int g_bs_on_device = 2;

}  // namespace

// the context's own choice (ohevc_ctx_set_option) or the process default
static inline bool filters_on_device(const ohevc_ctx *ctx)
{
    const int o = ohevc_ctx_get_option(ctx, OHEVC_OPT_FILTERS_ON_DEVICE);
    return o >= 0 ? o != 0 : g_filters_on_device;
}

extern "C" int ohevc_debug_set_bs_on_device(int mode) { const int prev = g_bs_on_device; g_bs_on_device = mode < 0 ? 0 : mode > 2 ? 2 : mode; return prev; }

extern "C" int ohevc_tables_bs_wanted(ohevc_ctx *ctx, int log2_ctb_size, int sao_enabled, int chroma_format_idc, int emulate_filter_lag)
{
    const bool lag = emulate_filter_lag && log2_ctb_size == 4 && sao_enabled && chroma_format_idc != 0 && chroma_format_idc != 3;
    return g_bs_on_device && filters_on_device(ctx) && ctx != nullptr && ohevc_ctx_has_device(ctx) && !lag ? g_bs_on_device : 0;
}

extern "C" int ohevc_debug_set_filters_on_device(int on) { const int prev = g_filters_on_device; g_filters_on_device = on != 0; return prev; }

extern "C" int ohevc_tables_derive_filters(ohevc_ctx *ctx, const ohevc_filter_maps *m)
{
    using namespace ohevc;
    OHEVC_REQUIRE(ctx != nullptr && m != nullptr, "null argument");
    OHEVC_REQUIRE(m->width > 0 && m->height > 0 && m->log2_ctb_size >= 4 && m->log2_ctb_size <= 6 && m->log2_min_cb_size >= 3 && m->log2_min_pu_size >= 2 &&
                  m->chroma_format_idc >= 0 && m->chroma_format_idc <= 3, "picture geometry");
    OHEVC_REQUIRE(m->horizontal_bs && m->vertical_bs && m->bs_width > 0 && m->qp_y_tab && m->min_cb_width > 0 && m->deblock && m->deblock_stride >= 2,
                  "deblocking maps");
    OHEVC_REQUIRE(!m->sao_enabled || (m->sao && m->filter_slice_edges && m->tab_slice_address && m->ctb_addr_rs_to_ts && m->tile_id), "SAO maps");
    OHEVC_REQUIRE(!m->pcm_or_bypass || (m->is_pcm && m->min_pu_width > 0 && m->min_pu_height > 0), "pcm map");
    Ctx c{*m, ctx};
    c.edges.reserve(32768);
    c.hs = m->chroma_format_idc == 1 || m->chroma_format_idc == 2; c.vs = m->chroma_format_idc == 1;
    const int ctb_size = 1 << m->log2_ctb_size, ctb_w = (m->width + ctb_size - 1) >> m->log2_ctb_size, ctb_h = (m->height + ctb_size - 1) >> m->log2_ctb_size;
    // With 16x16 CTBs the reference's one-CTB filter lag is visible in its output (ohevc_hip.h, OHEVC_SAO_LAG_*): which samples a
    // chroma SAO block saw before their horizontal edge was filtered depends on the ORDER of the driver calls.  To stay bit-identical
    // the drivers' control flow is replayed CTB by CTB in decoding order (hls_decode_entry, hevc.c:2644-2698) and the call order kept.
    c.lag = m->emulate_filter_lag && m->log2_ctb_size == 4 && m->sao_enabled && m->chroma_format_idc != 0 && m->chroma_format_idc != 3;
    if (c.lag) {
        OHEVC_REQUIRE(m->ctb_addr_ts_to_rs != nullptr, "filter-lag emulation needs pps->ctb_addr_ts_to_rs");
        int x = 0, y = 0;
        for (int ts = 0; ts < ctb_w * ctb_h; ts++) {
            const int rs = m->ctb_addr_ts_to_rs[ts];
            x = (rs % ctb_w) << m->log2_ctb_size; y = (rs / ctb_w) << m->log2_ctb_size;
            c.hls_filters(x, y);
        }
        c.hls_filter(x, y);                                    // hevc.c:2693-2695: the last CTB of the picture releases itself
        c.release_held_sao();
    } else if (filters_on_device(ctx) && ohevc_ctx_has_device(ctx)) {
        // the product path: no per-edge host work - the maps travel, ohevc_dev_deblock_maps derives and filters;
        // what stays here is one SAO record per CTB and plane
        ohevc_dbk_maps d = {};
        d.vertical_bs = m->vertical_bs; d.horizontal_bs = m->horizontal_bs; d.bs_width = m->bs_width;
        d.qp_y_tab = m->qp_y_tab; d.min_cb_width = m->min_cb_width; d.deblock = m->deblock; d.deblock_stride = m->deblock_stride;
        d.is_pcm = m->pcm_or_bypass ? m->is_pcm : nullptr; d.min_pu_width = m->min_pu_width; d.min_pu_height = m->min_pu_height;
        d.width = m->width; d.height = m->height; d.log2_ctb_size = m->log2_ctb_size; d.log2_min_cb_size = m->log2_min_cb_size;
        d.log2_min_pu_size = m->log2_min_pu_size; d.chroma_format_idc = m->chroma_format_idc;
        d.cb_qp_offset = m->cb_qp_offset; d.cr_qp_offset = m->cr_qp_offset;
        int r;
        if (m->cbf_luma != nullptr && g_bs_on_device) {         // boundary strengths on the device too: the motion field travels instead of them,
            OHEVC_REQUIRE(m->min_tb_width > 0 && m->min_tb_height > 0 && m->log2_min_tb_size >= 2, "cbf_luma map");      // or nothing (tab_mvf NULL)
            ohevc_bs_maps b = {};
            b.mvf = static_cast<const uint8_t *>(m->tab_mvf); b.mvf_stride = m->mvf_stride; b.off_mv = m->mvf_off_mv; b.off_poc = m->mvf_off_poc;
            b.off_pred_flag = m->mvf_off_pred_flag; b.pred_flag_bytes = m->mvf_pred_flag_bytes;
            b.cbf_luma = m->cbf_luma; b.min_pu_width = m->min_pu_width; b.min_pu_height = m->min_pu_height; b.log2_min_pu_size = m->log2_min_pu_size;
            b.min_tb_width = m->min_tb_width; b.min_tb_height = m->min_tb_height; b.log2_min_tb_size = m->log2_min_tb_size;
            b.log2_ctb_size = m->log2_ctb_size; b.bs_width = m->bs_width; b.width = m->width; b.height = m->height;
            b.loop_filter_across_tiles = m->loop_filter_across_tiles;
            r = ohevc_rec_deblock_maps_bs(ctx, &d, &b);
        } else {
            r = ohevc_rec_deblock_maps(ctx, &d);
        }
        if (r != OHEVC_OK) return r;
        if (m->sao_enabled)
            for (int y = 0; y < m->height; y += ctb_size)
                for (int x = 0; x < m->width; x += ctb_size) c.sao_ctb(x, y);
        return c.rc;
    } else {
        // record-only contexts (host-logic tests through the frame sink) and ohevc_debug_set_filters_on_device(0): the same
        // derivation on the host, one job per edge
        for (int y = 0; y < m->height; y += ctb_size)
            for (int x = 0; x < m->width; x += ctb_size) {
                c.deblock_ctb(x, y);
                if (m->sao_enabled) c.sao_ctb(x, y);
            }
    }
    const int r = ohevc_rec_deblock_bulk(ctx, c.edges.data(), (int)c.edges.size());
    return c.rc != OHEVC_OK ? c.rc : r;
}
