// host_jobs.hip -- host-side (CPU) job construction helpers of the C ABI; no device code.
#include <string.h>
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

extern "C" int ohevc_intra_make_job(const ohevc_intra_geom *g, int x0, int y0, int log2_size, int c_idx, int mode,
                                    int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right,
                                    ohevc_intra_job *out)
{
    using namespace ohevc;
    OHEVC_REQUIRE(g != nullptr && out != nullptr, "null argument");
    OHEVC_REQUIRE(log2_size >= 2 && log2_size <= 5, "log2_size must be 2..5");
    OHEVC_REQUIRE(c_idx >= 0 && c_idx <= 2 && mode >= 0 && mode <= 34, "c_idx / mode");
    OHEVC_REQUIRE(!g->constrained_intra_pred, "constrained_intra_pred streams are not supported yet");
    const int cfi = g->chroma_format_idc;
    const int hs = c_idx ? (cfi == 1 || cfi == 2) : 0, vs = c_idx ? (cfi == 1) : 0;
    const int n = 1 << log2_size, nlh = n << hs, nlv = n << vs;
    const int bits = g->log2_ctb_size - g->log2_min_tb_size, mask = (1 << bits) - 1;
    const int x_tb = (x0 >> g->log2_min_tb_size) & mask, y_tb = (y0 >> g->log2_min_tb_size) & mask;
    const int cur = zscan_in_ctb(x_tb, y_tb, bits);
    // hevcpred_template.c:105-109: "ahead" neighbours must already be decoded in z-scan order
    const bool bl = cand_bottom_left && cur > zscan_in_ctb(x_tb - 1, (y_tb + (nlv >> g->log2_min_tb_size)) & mask, bits);
    const bool ur = cand_up_right && cur > zscan_in_ctb((x_tb + (nlh >> g->log2_min_tb_size)) & mask, y_tb - 1, bits);
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
    return OHEVC_OK;
}
