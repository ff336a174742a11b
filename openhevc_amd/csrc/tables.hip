// tables.hip -- the drop-in back-end: recording implementations of the reference's table slots (include/ohevc_tables.h).
//
// Every function here has the exact signature of a HEVCDSPContext / VideoDSPContext slot.  Instead of touching host
// pixels it translates its pointer arguments back to (picture slot, plane, x, y) through the registry and appends a job
// to the calling thread's ohevc_ctx.  Host code only.
//
// Call-sequence knowledge used (all from the reference's call sites):
//   * residuals: idct*/transform_skip/transform_rdpcm run IN PLACE on lc->tu.coeffs and are followed by
//     transform_add on the same pointer (hevc_cabac.c:1868-1949) -> the in-place calls only note the pending kind, the
//     raw coefficients are copied when transform_add arrives (they are reused by the next TU, hevc.h:1063);
//   * bi-prediction: put_hevc_{qpel,epel}(tmp, MAX_PB_SIZE, ref0...) is followed by put_hevc_*_bi[_w](dst, .., ref1.., tmp, ..)
//     with the same tmp (hevc.c:1761-1773,1933-1948) -> the first call only notes reference 0;
//   * edge emulation: vdsp.emulated_edge_mc(buf, src - offset, ..., src_x, src_y, w, h) precedes the MC call that reads
//     `buf + buf_offset` (hevc.c:1660-1675) -> it notes (picture, src_x, src_y) for that buffer and copies nothing; the
//     MC kernel clamps coordinates instead.
#include <algorithm>
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>
#include <string.h>
#include <vector>
#include "common.hpp"
#include <time.h>
#include "ohevc_tables.h"

namespace {

struct HostPic {
    int slot = -1, bd = 8, ps = 1;
    uint8_t *data[3] = {};
    int linesize[3] = {};
    int w[3] = {}, h[3] = {};
    size_t bytes[3] = {};             // linesize * h
    uint32_t magic[3] = {};           // floor(2^32 / linesize) + 1: row = (offset * magic) >> 32, at most one too high
    void finish()
    {
        for (int c = 0; c < 3; c++) {
            bytes[c] = (data[c] && linesize[c] > 0) ? (size_t)linesize[c] * h[c] : 0;
            magic[c] = linesize[c] > 1 ? (uint32_t)((1ull << 32) / (uint32_t)linesize[c]) + 1 : 0;
        }
    }
};

// A registry entry as other threads see it.  Writers (one at a time: Registry::m) bump `gen` to an odd value, store the fields,
// bump it again; readers copy the fields and keep the copy only if `gen` was even and did not move (a seqlock: the table slots
// look pictures up on every call and must not take a lock).  All accesses are atomic (relaxed: plain moves on x86).
struct Entry {
    std::atomic<uint32_t> gen{0};
    HostPic v;
    template <typename T> static T ld(const T &f) { return __atomic_load_n(&f, __ATOMIC_RELAXED); }
    template <typename T> static void st(T &f, T x) { __atomic_store_n(&f, x, __ATOMIC_RELAXED); }
    int slot() const { return ld(v.slot); }
    bool snapshot(HostPic &out, uint32_t *seen = nullptr) const
    {
        for (int tries = 0; tries < 64; tries++) {
            const uint32_t g = gen.load(std::memory_order_acquire);
            if (seen) *seen = g;
            if (g & 1) continue;                               // being rewritten
            out.slot = ld(v.slot); out.bd = ld(v.bd); out.ps = ld(v.ps);
            for (int c = 0; c < 3; c++) {
                out.data[c] = ld(v.data[c]); out.linesize[c] = ld(v.linesize[c]); out.w[c] = ld(v.w[c]); out.h[c] = ld(v.h[c]);
                out.bytes[c] = ld(v.bytes[c]); out.magic[c] = ld(v.magic[c]);
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (gen.load(std::memory_order_relaxed) == g) return out.slot >= 0;
        }
        return false;
    }
    void publish(const HostPic &src)                           // under Registry::m
    {
        const uint32_t g = gen.load(std::memory_order_relaxed);
        gen.store(g + 1, std::memory_order_relaxed);
        std::atomic_thread_fence(std::memory_order_release);
        st(v.slot, src.slot); st(v.bd, src.bd); st(v.ps, src.ps);
        for (int c = 0; c < 3; c++) {
            st(v.data[c], src.data[c]); st(v.linesize[c], src.linesize[c]); st(v.w[c], src.w[c]); st(v.h[c], src.h[c]);
            st(v.bytes[c], src.bytes[c]); st(v.magic[c], src.magic[c]);
        }
        gen.store(g + 2, std::memory_order_release);
    }
    void retire()                                              // under Registry::m: the picture is gone, the entry free
    {
        const uint32_t g = gen.load(std::memory_order_relaxed);
        gen.store(g + 1, std::memory_order_relaxed);
        std::atomic_thread_fence(std::memory_order_release);
        st(v.slot, -1);
        gen.store(g + 2, std::memory_order_release);
    }
};

// host-buffer registry: common to all contexts sharing one picture store (frame threads resolve MC source pointers into
// pictures other threads registered).  Readers take no lock (Entry); writers hold `m`.
inline uint64_t next_registry_id() { static std::atomic<uint64_t> n{0}; return ++n; }
struct Registry {
    std::mutex m;
    Entry pics[OHEVC_MAX_PICTURES + 1];        // one per picture of the store (ohevc_ctx.h) - the store's own limit is met first
    std::atomic<int> n{0};
    const uint64_t id = next_registry_id();    // never reused (an address may be: kept_snapshot)
};

struct TablesState {
    std::shared_ptr<Registry> reg;
    Entry *pics = nullptr;            // = reg->pics
    int npics() const { return reg->n.load(std::memory_order_acquire); }
    int cur = -1;                     // index into pics
    int status = OHEVC_OK;
    // ohevc_tables_emulate_filter_lag: call-order bookkeeping of the current frame
    int lag_on = 0;
    uint32_t seq = 0;
    std::unordered_map<uint32_t, uint32_t> h_edge_seq;                    // (plane, x, y) of a chroma horizontal edge -> call number
    std::vector<std::pair<ohevc_sao_job, uint32_t>> held_sao;           // chroma SAO jobs held back until the frame ends
    // ohevc_tables_set_concurrent: several threads (the reference's slice threads: WPP rows / tiles of ONE picture) record
    // into this context at the same time; every recorder call then runs under `spin`
    bool concurrent = false;
    std::atomic_flag spin = ATOMIC_FLAG_INIT;
    std::vector<int> upsampled;       // SHVC: enhancement-layer picture slots already resampled (cleared by begin_frame / registration)
    ohevc_HEVCDSPContext saved = {};  // the reference's own C slots (put_pcm is still executed on the host into scratch)
};

std::mutex g_lock;
std::map<ohevc_ctx *, TablesState *> g_states;
void (*g_ref_put_pcm[15])(uint8_t *, ptrdiff_t, int, int, struct GetBitContext *, int) = {};

struct Pending {
    const int16_t *coeffs = nullptr;  // residual kind noted by the in-place transform call
    int kind = -1;
    int col_limit = 0;                // inverse DCT: the bound the reference passed with it (0: none)
    // cross-component prediction (hevc.c:1291-1365): the transform unit's luma block as last handed to transform_add, and the
    // res_scale_val announced for the next chroma block (ohevc_tables_cross_component)
    const int16_t *luma_coeffs = nullptr;
    int luma_kind = -1, luma_log2 = 0, cross_scale = 0;
    int up_bl_slot = -1;              // SHVC: base-layer picture named by the last horizontal-pass slot call
    const int16_t *bi_tmp = nullptr;  // first half of a bi-prediction
    int bi_slot = -1, bi_plane = 0, bi_sx = 0, bi_sy = 0, bi_mx = 0, bi_my = 0;
    struct Emu { const uint8_t *buf = nullptr; ptrdiff_t linesize = 0; int slot = -1, plane = 0, x = 0, y = 0; } emu[4];
    int emu_next = 0;
};

thread_local ohevc_ctx *tl_ctx = nullptr;
thread_local TablesState *tl_state = nullptr;
thread_local Pending tl_pend;

std::atomic<int> g_unbound_calls{0};   // table calls from threads without a bound context: reported by ohevc_tables_status

void fail(int rc)
{
    if (tl_state) { if (tl_state->status == OHEVC_OK) tl_state->status = rc; }
    else g_unbound_calls.fetch_add(1, std::memory_order_relaxed);
}

// protects the TABLE layer's shared bookkeeping (filter-lag call order, up-sampling state) when several threads feed one
// context (ohevc_tables_set_concurrent); the recorder itself gives every thread its own arrays (ohevc_ctx_set_concurrent)
struct Guard {
    TablesState *s;
    explicit Guard(TablesState *st) : s(st && st->concurrent ? st : nullptr)
    {
        if (s) while (s->spin.test_and_set(std::memory_order_acquire)) __builtin_ia32_pause();
    }
    ~Guard() { if (s) s->spin.clear(std::memory_order_release); }
};

struct Loc { int pic = -1, slot = -1, plane = 0, x = 0, y = 0; };     // registry entry, picture-store slot, position

// OHEVC_TRACE=slots: cycle counters per slot family, printed by ohevc_tables_forget (host-side tuning aid)
enum { K_TU, K_MC_HALF, K_MC, K_EMU, K_DBK, K_SAO, K_INTRA, K_PCM, K_END, K_NFAM };
const char *const kFamName[K_NFAM] = {"transform_add", "put_hevc_*(first half)", "put_hevc_*_uni/bi", "emulated_edge_mc", "loop_filter",
                                      "sao", "intra_pred", "put_pcm", "end_frame"};
unsigned long long g_prof_cycles[K_NFAM], g_prof_calls[K_NFAM];
const bool g_prof_on = ohevc::config().profile_slots;       // OHEVC_TRACE=slots
struct Prof {
    int k; unsigned long long t0;
    explicit Prof(int kk) : k(kk), t0(g_prof_on ? __builtin_ia32_rdtsc() : 0) {}
    ~Prof() { if (g_prof_on) { g_prof_cycles[k] += __builtin_ia32_rdtsc() - t0; g_prof_calls[k]++; } }
};

// which registered picture/plane contains host address p?  (called for every table slot: no divisions)
inline bool locate_in(const HostPic &hp, const uint8_t *p, Loc &out)
{
    for (int c = 0; c < 3; c++) {
        const size_t off = (size_t)(p - hp.data[c]);             // wraps to a huge value when p is below the plane
        if (off >= hp.bytes[c]) continue;
        const uint32_t o = (uint32_t)off, ls = (uint32_t)hp.linesize[c];
        uint32_t y = (uint32_t)(((uint64_t)o * hp.magic[c]) >> 32);
        int32_t xb = (int32_t)(o - y * ls);
        if (xb < 0) { y--; xb += (int32_t)ls; }
        if (xb >= hp.w[c] * hp.ps) continue;
        out.plane = c; out.x = hp.ps == 2 ? xb >> 1 : xb; out.y = (int)y;
        return true;
    }
    return false;
}

thread_local int tl_hint[2] = { -1, -1 };      // the registry entries the last two successful look-ups hit (a picture predicts from one or two others)

// Every table call resolves one to three pointers, and a snapshot copies some thirty fields: 8 % of a decoding thread's time on an
// inter-coded stream (sampled, round 5).  An entry's `gen` only moves when the entry is rewritten, so a thread keeps the snapshots it took -
// per entry, stamped with the registry (two decoders of one process have two) and the `gen` it saw - and a look-up is one load of
// `gen` plus the arithmetic.
struct KeptSnapshot { uint64_t of = 0; uint32_t gen = 1; HostPic hp; };
thread_local KeptSnapshot tl_kept[OHEVC_MAX_PICTURES + 1];
const HostPic *kept_snapshot(const Entry &e, int i)
{
    KeptSnapshot &k = tl_kept[i];
    const uint32_t g = e.gen.load(std::memory_order_acquire);
    const uint64_t reg = tl_state->reg->id;
    if (k.of != reg || k.gen != g || (g & 1)) {
        uint32_t seen = 1;
        k.of = reg;
        if (!e.snapshot(k.hp, &seen)) { k.gen = 1; return nullptr; }       // (snapshot() says no for a free entry too: slot < 0)
        k.gen = seen;
    }
    return k.hp.slot >= 0 ? &k.hp : nullptr;
}

bool locate(const uint8_t *p, Loc &out, int only_pic = -1)
{
    if (!tl_state) return false;
    if (only_pic >= 0) {
        const HostPic *hp = kept_snapshot(tl_state->pics[only_pic], only_pic);
        if (!hp || !locate_in(*hp, p, out)) return false;
        out.pic = only_pic; out.slot = hp->slot;
        return true;
    }
    const int n = tl_state->npics();
    for (int h = 0; h < 2; h++) {
        const int i = tl_hint[h];
        const HostPic *hp = i >= 0 && i < n ? kept_snapshot(tl_state->pics[i], i) : nullptr;
        if (hp && locate_in(*hp, p, out)) { out.pic = i; out.slot = hp->slot; return true; }
    }
    for (int i = 0; i < n; i++) {
        const HostPic *hp = tl_state->pics[i].slot() < 0 ? nullptr : kept_snapshot(tl_state->pics[i], i);
        if (hp && locate_in(*hp, p, out)) {
            out.pic = (int)i; out.slot = hp->slot;
            tl_hint[1] = tl_hint[0]; tl_hint[0] = i;
            return true;
        }
    }
    return false;
}

bool locate_cur(const uint8_t *p, Loc &out) { return tl_state && tl_state->cur >= 0 && locate(p, out, tl_state->cur); }

// ------------------------------------------------------------------ residuals
void note_kind(const int16_t *coeffs, int kind) { tl_pend.coeffs = coeffs; tl_pend.kind = kind; }

// col_limit (hevc_cabac.c:1923-1934): the reference's bound on where the block's non-zero coefficients lie - columns below min(col_limit, N), rows
// below min(col_limit + 4, N), the ranges its own transforms read (hevcdsp_template.c:271-291).  Only that rectangle is recorded and uploaded.
template <int LOG2> void t_idct(int16_t *coeffs, int col_limit) { note_kind(coeffs, OHEVC_TU_IDCT); tl_pend.col_limit = col_limit; }
template <int LOG2> void t_idct_dc(int16_t *coeffs) { note_kind(coeffs, OHEVC_TU_DC); }
void t_idct_4x4_luma(int16_t *coeffs) { note_kind(coeffs, OHEVC_TU_DST4); }
void t_transform_skip(int16_t *coeffs, int16_t) { note_kind(coeffs, OHEVC_TU_SKIP); }
void t_transform_rdpcm(int16_t *coeffs, int16_t, int mode)
{
    const bool after_skip = tl_pend.coeffs == coeffs && tl_pend.kind == OHEVC_TU_SKIP;
    note_kind(coeffs, after_skip ? (mode ? OHEVC_TU_SKIP_RDPCM_V : OHEVC_TU_SKIP_RDPCM_H)
                                 : (mode ? OHEVC_TU_BYPASS_RDPCM_V : OHEVC_TU_BYPASS_RDPCM_H));
}

template <int LOG2> void t_transform_add(uint8_t *dst, int16_t *coeffs, ptrdiff_t)
{
    Prof prof_(K_TU);
    Loc l;
    if (!tl_ctx || !locate_cur(dst, l)) { fail(OHEVC_ERR_STATE); return; }
    // no pending in-place transform on this pointer: the caller handed over a finished residual (transquant bypass)
    const bool pending = tl_pend.coeffs == coeffs && tl_pend.kind >= 0;
    const int kind = pending ? tl_pend.kind : OHEVC_TU_BYPASS;
    tl_pend.coeffs = nullptr; tl_pend.kind = -1;
    const int scale = l.plane ? tl_pend.cross_scale : 0;
    tl_pend.cross_scale = 0;                              // announced for exactly one chroma block
    if (l.plane == 0) { tl_pend.luma_coeffs = coeffs; tl_pend.luma_kind = kind; tl_pend.luma_log2 = LOG2; }
    int rc;
    if (scale) {
        // The host has already mixed its idea of the luma residual into this buffer -- computed from lc->tu.coeffs[0], which
        // behind recording tables still holds the RAW luma coefficients y:
        //   coded chroma block    c' = (int16)(c + ((scale * y) >> 3))     hevc_cabac.c:1942-1948
        //   no coded coefficients c' = (scale * y) >> 3, no transform call  hevc.c:1315-1330
        // Undo it, exactly (arithmetic mod 2^16): what is left are the block's own coefficients -- all zero in the second
        // case, which like a transquant-bypass block arrives without a pending transform and is recorded as one.
        const int16_t *y = tl_pend.luma_coeffs;
        if (!y || tl_pend.luma_log2 != LOG2 || y == coeffs) { fail(OHEVC_ERR_STATE); return; }
        int16_t own[1 << (2 * LOG2)];
        for (int i = 0; i < (1 << (2 * LOG2)); i++) own[i] = (int16_t)(coeffs[i] - ((scale * y[i]) >> 3));
        rc = ohevc_rec_tu_cross(tl_ctx, l.plane, l.x, l.y, LOG2, kind, own, tl_pend.luma_kind, y, scale, 1);
    } else {
        const int lim = kind == OHEVC_TU_IDCT && pending && tl_pend.col_limit > 0 ? tl_pend.col_limit : 64;
        rc = ohevc_rec_tu_limited(tl_ctx, l.plane, l.x, l.y, LOG2, kind, coeffs, 1, lim, lim + 4);
    }
    tl_pend.col_limit = 0;
    if (rc != OHEVC_OK) fail(rc);
}

void t_put_pcm(uint8_t *dst, ptrdiff_t, int width, int height, struct GetBitContext *gb, int pcm_bit_depth)
{
    Prof prof_(K_PCM);
    Loc l;
    if (!tl_ctx || !locate_cur(dst, l)) { fail(OHEVC_ERR_STATE); return; }
    const HostPic &hp = tl_state->pics[tl_state->cur].v;      // this thread's own picture
    // square blocks, or the two-squares-tall chroma block of a 4:2:2 coding block (hls_pcm_sample, hevc.c:1613-1620)
    auto ref_put_pcm = __atomic_load_n(&g_ref_put_pcm[hp.bd], __ATOMIC_RELAXED);
    if (!ref_put_pcm || (height != width && height != 2 * width) || width < 4 || width > 32) { fail(OHEVC_ERR_STATE); return; }
    // the bit reader is the reference's: let its own put_pcm unpack into scratch, then ship the samples as square blocks
    uint16_t scratch16[32 * 64];
    uint8_t *scratch = reinterpret_cast<uint8_t *>(scratch16);
    ref_put_pcm(scratch, (ptrdiff_t)width * hp.ps, width, height, gb, pcm_bit_depth);
    int16_t samples[32 * 64];
    for (int i = 0; i < width * height; i++) samples[i] = hp.ps == 2 ? (int16_t)scratch16[i] : (int16_t)scratch[i];
    int log2 = 2;
    while ((1 << log2) < width) log2++;
    for (int half = 0; half * width < height; half++) {
        int rc = ohevc_rec_tu(tl_ctx, l.plane, l.x, l.y + half * width, log2, OHEVC_TU_PCM, samples + half * width * width, 1);
        if (rc != OHEVC_OK) fail(rc);
    }
}

// ------------------------------------------------------------------ motion compensation
void t_emulated_edge_mc(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize,
                        int, int, int src_x, int src_y, int, int)
{
    Prof prof_(K_EMU);
    if (!tl_state) return;
    // src == plane_base + src_y * linesize + src_x * ps  (possibly outside the plane): recover the plane by its base
    int k = -1;
    for (int i = 0; i < 4; i++) if (tl_pend.emu[i].buf == buf) k = i;      // a buffer holds one window at a time
    if (k < 0) { k = tl_pend.emu_next; tl_pend.emu_next = (tl_pend.emu_next + 1) & 3; }
    Pending::Emu &e = tl_pend.emu[k];
    e.buf = nullptr;
    for (int i = 0; i < tl_state->npics(); i++) {
        HostPic hp;
        if (tl_state->pics[i].slot() < 0 || !tl_state->pics[i].snapshot(hp)) continue;
        for (int c = 0; c < 3; c++) {
            if (hp.data[c] && hp.linesize[c] == src_linesize &&
                src - (ptrdiff_t)src_y * src_linesize - (ptrdiff_t)src_x * hp.ps == hp.data[c]) {
                e.buf = buf; e.linesize = buf_linesize; e.slot = hp.slot; e.plane = c; e.x = src_x; e.y = src_y;
                return;
            }
        }
    }
    fail(OHEVC_ERR_STATE);
}

// resolve an MC source pointer: either inside a registered picture or inside a noted edge-emulation buffer
bool resolve_src(const uint8_t *src, int ps, int &slot, int &plane, int &sx, int &sy)
{
    // the reference's two buffers are adjacent members of HEVCLocalContext, (MAX_PB_SIZE + 7) rows each (hevc.h:1162-1163):
    // take the noted window whose base is the closest one below src
    const Pending::Emu *best = nullptr;
    for (const Pending::Emu &e : tl_pend.emu) {
        if (!e.buf) continue;
        const ptrdiff_t d = src - e.buf;
        if (d >= 0 && d < e.linesize * (64 + 7) && (!best || e.buf > best->buf)) best = &e;
    }
    if (best) {
        const ptrdiff_t d = src - best->buf;
        slot = best->slot; plane = best->plane;
        sy = best->y + (int)(d / best->linesize); sx = best->x + (int)(d % best->linesize) / ps;
        return true;
    }
    Loc l;
    if (!locate(src, l)) return false;
    slot = l.slot; plane = l.plane; sx = l.x; sy = l.y;
    return true;
}

void mc_first_half(int16_t *tmp, uint8_t *src, int mx, int my)
{
    Prof prof_(K_MC_HALF);
    if (!tl_ctx || !tl_state || tl_state->cur < 0) { fail(OHEVC_ERR_STATE); return; }
    const int ps = tl_state->pics[tl_state->cur].v.ps;
    Pending &p = tl_pend;
    if (!resolve_src(src, ps, p.bi_slot, p.bi_plane, p.bi_sx, p.bi_sy)) { p.bi_tmp = nullptr; fail(OHEVC_ERR_STATE); return; }
    p.bi_tmp = tmp; p.bi_mx = mx; p.bi_my = my;
}

void mc_record(uint8_t *dst, uint8_t *src, const int16_t *src2, int height, int width, int mx, int my,
               bool weighted, int denom, int wx0, int wx1, int ox0, int ox1)
{
    Prof prof_(K_MC);
    Loc l;
    if (!tl_ctx || !locate_cur(dst, l)) { fail(OHEVC_ERR_STATE); return; }
    const int ps = tl_state->pics[tl_state->cur].v.ps;
    ohevc_mc_job j = {};
    j.x = (uint16_t)l.x; j.y = (uint16_t)l.y; j.w = (uint8_t)width; j.h = (uint8_t)height; j.plane = (uint8_t)l.plane;
    int slot, plane, sx, sy;
    if (!resolve_src(src, ps, slot, plane, sx, sy)) { fail(OHEVC_ERR_STATE); return; }
    if (src2) {                         // bi: reference 0 is the noted first half, reference 1 is this call's src
        if (tl_pend.bi_tmp != src2) { fail(OHEVC_ERR_STATE); return; }
        j.flags |= OHEVC_MC_BI;
        j.ref0 = (int8_t)tl_pend.bi_slot; j.sx0 = (int16_t)tl_pend.bi_sx; j.sy0 = (int16_t)tl_pend.bi_sy;
        j.mx0 = (uint8_t)tl_pend.bi_mx; j.my0 = (uint8_t)tl_pend.bi_my;
        j.ref1 = (int8_t)slot; j.sx1 = (int16_t)sx; j.sy1 = (int16_t)sy; j.mx1 = (uint8_t)mx; j.my1 = (uint8_t)my;
        tl_pend.bi_tmp = nullptr;
    } else {
        j.ref0 = (int8_t)slot; j.sx0 = (int16_t)sx; j.sy0 = (int16_t)sy; j.mx0 = (uint8_t)mx; j.my0 = (uint8_t)my;
    }
    if (weighted) {
        j.flags |= OHEVC_MC_WEIGHTED;
        j.denom = (uint8_t)denom; j.wx0 = (int16_t)wx0; j.wx1 = (int16_t)wx1; j.ox0 = (int16_t)ox0; j.ox1 = (int16_t)ox1;
    }
    int rc = ohevc_rec_mc(tl_ctx, &j);
    if (rc != OHEVC_OK) fail(rc);
}

void t_put(int16_t *dst, ptrdiff_t, uint8_t *src, ptrdiff_t, int, intptr_t mx, intptr_t my, int)
{
    mc_first_half(dst, src, (int)mx, (int)my);
}
void t_uni(uint8_t *dst, ptrdiff_t, uint8_t *src, ptrdiff_t, int height, intptr_t mx, intptr_t my, int width)
{
    mc_record(dst, src, nullptr, height, width, (int)mx, (int)my, false, 0, 0, 0, 0, 0);
}
void t_uni_w(uint8_t *dst, ptrdiff_t, uint8_t *src, ptrdiff_t, int height, int denom, int wx, int ox, intptr_t mx, intptr_t my, int width)
{
    mc_record(dst, src, nullptr, height, width, (int)mx, (int)my, true, denom, wx, 0, ox, 0);
}
void t_bi(uint8_t *dst, ptrdiff_t, uint8_t *src, ptrdiff_t, int16_t *src2, ptrdiff_t, int height, intptr_t mx, intptr_t my, int width)
{
    mc_record(dst, src, src2, height, width, (int)mx, (int)my, false, 0, 0, 0, 0, 0);
}
void t_bi_w(uint8_t *dst, ptrdiff_t, uint8_t *src, ptrdiff_t, int16_t *src2, ptrdiff_t, int height, int denom, int wx0, int wx1,
            int ox0, int ox1, intptr_t mx, intptr_t my, int width)
{
    mc_record(dst, src, src2, height, width, (int)mx, (int)my, true, denom, wx0, wx1, ox0, ox1);
}

// ------------------------------------------------------------------ in-loop filters
void dbk_record(uint8_t *pix, bool vertical, int beta, const int *tc, const uint8_t *no_p, const uint8_t *no_q)
{
    Prof prof_(K_DBK);
    Loc l;
    if (!tl_ctx || !locate_cur(pix, l)) { fail(OHEVC_ERR_STATE); return; }
    ohevc_dbk_job j = {};
    j.x = (uint16_t)l.x; j.y = (uint16_t)l.y; j.plane = (uint8_t)l.plane; j.beta = (uint8_t)beta;
    j.tc[0] = (int16_t)tc[0]; j.tc[1] = (int16_t)tc[1];
    j.flags = (uint8_t)((vertical ? OHEVC_DBK_VERTICAL_EDGE : 0) | (no_p[0] ? OHEVC_DBK_NO_P0 : 0) | (no_p[1] ? OHEVC_DBK_NO_P1 : 0) |
                        (no_q[0] ? OHEVC_DBK_NO_Q0 : 0) | (no_q[1] ? OHEVC_DBK_NO_Q1 : 0));
    if (tl_state->lag_on && !vertical && l.plane > 0) {       // call-order bookkeeping is shared by the recording threads
        Guard guard_(tl_state);
        tl_state->h_edge_seq[((uint32_t)l.plane << 30) | ((uint32_t)l.y << 15) | (uint32_t)l.x] = ++tl_state->seq;
    }
    int rc = ohevc_rec_deblock(tl_ctx, &j);
    if (rc != OHEVC_OK) fail(rc);
}
void t_h_luma(uint8_t *pix, ptrdiff_t, int beta, int *tc, uint8_t *no_p, uint8_t *no_q) { dbk_record(pix, false, beta, tc, no_p, no_q); }
void t_v_luma(uint8_t *pix, ptrdiff_t, int beta, int *tc, uint8_t *no_p, uint8_t *no_q) { dbk_record(pix, true, beta, tc, no_p, no_q); }
void t_h_chroma(uint8_t *pix, ptrdiff_t, int *tc, uint8_t *no_p, uint8_t *no_q) { dbk_record(pix, false, 0, tc, no_p, no_q); }
void t_v_chroma(uint8_t *pix, ptrdiff_t, int *tc, uint8_t *no_p, uint8_t *no_q) { dbk_record(pix, true, 0, tc, no_p, no_q); }

void sao_record(uint8_t *dst, ohevc_SAOParams *sao, int *borders, int width, int height, int c_idx, int type, int restore,
                const uint8_t *ve, const uint8_t *he, const uint8_t *de)
{
    Prof prof_(K_SAO);
    Loc l;
    if (!tl_ctx || !locate_cur(dst, l)) { fail(OHEVC_ERR_STATE); return; }
    ohevc_sao_job j = {};
    j.x = (uint16_t)l.x; j.y = (uint16_t)l.y; j.w = (uint16_t)width; j.h = (uint16_t)height; j.plane = (uint8_t)l.plane;
    j.type = (uint8_t)type;
    j.klass = type == OHEVC_SAO_BAND ? sao->band_position[c_idx] : sao->eo_class[c_idx];
    j.borders = (uint8_t)((borders[0] ? 1 : 0) | (borders[1] ? 2 : 0) | (borders[2] ? 4 : 0) | (borders[3] ? 8 : 0));
    j.restore = (uint8_t)restore;
    if (restore)
        j.edges = (uint8_t)((ve[0] ? 1 : 0) | (ve[1] ? 2 : 0) | (he[0] ? 4 : 0) | (he[1] ? 8 : 0) |
                            (de[0] ? 16 : 0) | (de[1] ? 32 : 0) | (de[2] ? 64 : 0) | (de[3] ? 128 : 0));
    for (int k = 0; k < 5; k++) j.offset_val[k] = sao->offset_val[c_idx][k];
    static const bool trace = ohevc::config().trace_sao;
    if (trace)
        fprintf(stderr, "sao plane %d x %d y %d w %d h %d type %d klass %d borders %d restore %d edges %d quirks %d off %d %d %d %d %d\n", j.plane, j.x, j.y,
                j.w, j.h, j.type, j.klass, j.borders, j.restore, j.edges, j.quirks, j.offset_val[0], j.offset_val[1], j.offset_val[2],
                j.offset_val[3], j.offset_val[4]);
    if (tl_state->lag_on && c_idx > 0) {        // flags depend on calls still to come: decided in ohevc_tables_end_frame
        Guard guard_(tl_state);
        tl_state->held_sao.emplace_back(j, ++tl_state->seq);
        return;
    }
    int rc = ohevc_rec_sao(tl_ctx, &j);
    if (rc != OHEVC_OK) fail(rc);
}
// sao_filter_CTB passes (frame, sao_frame copy) as (dst, src): hevc_filter.c:270-274,308-315
void t_sao_band(uint8_t *dst, uint8_t *, ptrdiff_t, ptrdiff_t, ohevc_SAOParams *sao, int *borders, int width, int height, int c_idx)
{
    sao_record(dst, sao, borders, width, height, c_idx, OHEVC_SAO_BAND, 0, nullptr, nullptr, nullptr);
}
void t_sao_edge0(uint8_t *dst, uint8_t *, ptrdiff_t, ptrdiff_t, ohevc_SAOParams *sao, int *borders, int width, int height, int c_idx,
                 uint8_t *, uint8_t *, uint8_t *)
{
    sao_record(dst, sao, borders, width, height, c_idx, OHEVC_SAO_EDGE, 0, nullptr, nullptr, nullptr);
}
void t_sao_edge1(uint8_t *dst, uint8_t *, ptrdiff_t, ptrdiff_t, ohevc_SAOParams *sao, int *borders, int width, int height, int c_idx,
                 uint8_t *ve, uint8_t *he, uint8_t *de)
{
    sao_record(dst, sao, borders, width, height, c_idx, OHEVC_SAO_EDGE, 1, ve, he, de);
}

// ------------------------------------------------------------------ SHVC inter-layer up-sampling
// Call sequence (upsample_block_luma / upsample_block_mc, hevc_filter.c:1175-1310): emulated_edge_up_h on the base-layer frame,
// the horizontal slot into a scratch buffer, emulated_edge_up_v on that buffer, the vertical slot into the inter-layer
// reference picture.  The helpers' return values steer the caller's pointers, so they are reproduced; nothing is written.
int t_emulated_edge_up_h(uint8_t *, ptrdiff_t, const ohevc_HEVCWindow *, int, int, int bl_edge_left, int, int shift)
{
    return bl_edge_left < shift ? 0 : 1;                      // videodsp_template.c:110-126
}
int t_emulated_edge_up_v(int16_t *, ptrdiff_t, const ohevc_HEVCWindow *, int, int, int, int bl_edge_up, int, int, int shift)
{
    return bl_edge_up < shift ? 0 : 1;                        // videodsp_template.c:141-165
}

void t_up_h(int16_t *, ptrdiff_t, uint8_t *src, ptrdiff_t, int, int, int, int, int, const ohevc_HEVCWindow *, ohevc_UpsamplInf *)
{
    tl_pend.up_bl_slot = -1;
    if (!tl_ctx || !tl_state) { fail(OHEVC_ERR_STATE); return; }
    // src = plane + (bl_y - edge_top) * stride + bl_x - edge_left (+ shift): inside the plane except for the chroma rows, whose
    // first block starts at bl_y = -1 (the "- 4" of hevc_filter.c:1268,1280) -- accept a few rows of frame padding around the plane
    auto inside = [&](int i) {
        HostPic hp;
        if (tl_state->pics[i].slot() < 0 || !tl_state->pics[i].snapshot(hp)) return false;
        for (int c = 0; c < 3; c++) {
            if (!hp.data[c]) continue;
            const ptrdiff_t margin = (ptrdiff_t)8 * hp.linesize[c], off = src - (hp.data[c] - margin);
            if (off >= 0 && (size_t)off < hp.bytes[c] + 2 * (size_t)margin) { tl_pend.up_bl_slot = hp.slot; return true; }
        }
        return false;
    };
    // called three times per enhancement-layer CTB, always for the same base-layer picture: the registry entry of the last hit first
    static thread_local int up_hint = -1;
    const int n = tl_state->npics();
    if (up_hint >= 0 && up_hint < n && inside(up_hint)) return;
    for (int i = 0; i < n; i++)
        if (inside(i)) { up_hint = i; return; }
    fail(OHEVC_ERR_STATE);
}

int upsample_once(int el_slot, int bl_slot, const ohevc_HEVCWindow *w, const ohevc_UpsamplInf *u, int block_slots)
{
    Guard guard_(tl_state);
    for (int sdone : tl_state->upsampled) if (sdone == el_slot) return OHEVC_OK;
    int ew, eh, bw, bh, cfi, bd;
    int rc = ohevc_pic_info(tl_ctx, el_slot, &ew, &eh, &cfi, &bd);
    if (rc == OHEVC_OK) rc = ohevc_pic_info(tl_ctx, bl_slot, &bw, &bh, &cfi, &bd);
    if (rc != OHEVC_OK) return rc;
    ohevc_upsample_params p = {};
    p.el_width = ew; p.el_height = eh; p.bl_width = bw; p.bl_height = bh;
    p.win_left = w->left_offset; p.win_right = w->right_offset; p.win_top = w->top_offset; p.win_bottom = w->bottom_offset;
    p.add_x_luma = u->addXLum; p.add_y_luma = u->addYLum; p.scale_x_luma = u->scaleXLum; p.scale_y_luma = u->scaleYLum;
    p.add_x_chroma = u->addXCr; p.add_y_chroma = u->addYCr; p.scale_x_chroma = u->scaleXCr; p.scale_y_chroma = u->scaleYCr;
    p.idx = u->idx; p.block_slots = block_slots;
    rc = ohevc_pic_upsample(tl_ctx, el_slot, bl_slot, &p);
    if (rc == OHEVC_OK) tl_state->upsampled.push_back(el_slot);
    return rc;
}

void t_up_v(uint8_t *dst, ptrdiff_t, int16_t *, ptrdiff_t, int, int, int, int, int, int, int, const ohevc_HEVCWindow *w, ohevc_UpsamplInf *u)
{
    Loc l;
    // dst is the BASE of the inter-layer reference picture's plane (the slot adds the block position itself, :1905,1947)
    if (!tl_ctx || tl_pend.up_bl_slot < 0 || !locate(dst, l)) { fail(OHEVC_ERR_STATE); return; }
    int rc = upsample_once(l.slot, tl_pend.up_bl_slot, w, u, 1);
    if (rc != OHEVC_OK) fail(rc);
}

std::map<const void *, std::weak_ptr<Registry>> g_registries;    // keyed by ohevc_ctx_store_id

TablesState *state_of(ohevc_ctx *ctx, bool create)
{
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_states.find(ctx);
    if (it != g_states.end()) return it->second;
    if (!create) return nullptr;
    TablesState *s = new TablesState();
    std::weak_ptr<Registry> &w = g_registries[ohevc_ctx_store_id(ctx)];
    s->reg = w.lock();
    if (!s->reg) { s->reg = std::make_shared<Registry>(); w = s->reg; }
    s->pics = s->reg->pics;
    g_states[ctx] = s;
    return s;
}

}  // namespace

extern "C" int ohevc_tables_emulate_filter_lag(ohevc_ctx *ctx, int enable)
{
    TablesState *st = state_of(ctx, true);
    if (!st) return OHEVC_ERR_ARG;
    st->lag_on = enable != 0;
    return OHEVC_OK;
}

extern "C" void ohevc_hevcdsp_init_hip(ohevc_HEVCDSPContext *c, int bit_depth)
{
    if (!c) return;
    // (every decoding thread fills its own table copy and lands here: the same pointer, stored atomically)
    if (bit_depth >= 8 && bit_depth < 15 && c->put_pcm && c->put_pcm != t_put_pcm) __atomic_store_n(&g_ref_put_pcm[bit_depth], c->put_pcm, __ATOMIC_RELAXED);
    c->put_pcm = t_put_pcm;
    c->transform_add[0] = t_transform_add<2>; c->transform_add[1] = t_transform_add<3>;
    c->transform_add[2] = t_transform_add<4>; c->transform_add[3] = t_transform_add<5>;
    c->transform_skip = t_transform_skip;
    c->transform_rdpcm = t_transform_rdpcm;
    c->idct_4x4_luma = t_idct_4x4_luma;
    c->idct[0] = t_idct<2>; c->idct[1] = t_idct<3>; c->idct[2] = t_idct<4>; c->idct[3] = t_idct<5>;
    c->idct_dc[0] = t_idct_dc<2>; c->idct_dc[1] = t_idct_dc<3>; c->idct_dc[2] = t_idct_dc<4>; c->idct_dc[3] = t_idct_dc<5>;
    c->sao_band_filter = t_sao_band;
    c->sao_edge_filter[0] = t_sao_edge0;
    c->sao_edge_filter[1] = t_sao_edge1;
    for (int i = 0; i < 10; i++)
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                c->put_hevc_qpel[i][a][b] = t_put;        c->put_hevc_epel[i][a][b] = t_put;
                c->put_hevc_qpel_uni[i][a][b] = t_uni;    c->put_hevc_epel_uni[i][a][b] = t_uni;
                c->put_hevc_qpel_uni_w[i][a][b] = t_uni_w; c->put_hevc_epel_uni_w[i][a][b] = t_uni_w;
                c->put_hevc_qpel_bi[i][a][b] = t_bi;      c->put_hevc_epel_bi[i][a][b] = t_bi;
                c->put_hevc_qpel_bi_w[i][a][b] = t_bi_w;  c->put_hevc_epel_bi_w[i][a][b] = t_bi_w;
            }
    for (int i = 0; i < 3; i++) {
        c->upsample_filter_block_luma_h[i] = t_up_h; c->upsample_filter_block_cr_h[i] = t_up_h;
        c->upsample_filter_block_luma_v[i] = t_up_v; c->upsample_filter_block_cr_v[i] = t_up_v;
    }
    c->hevc_h_loop_filter_luma = c->hevc_h_loop_filter_luma_c = t_h_luma;
    c->hevc_v_loop_filter_luma = c->hevc_v_loop_filter_luma_c = t_v_luma;
    c->hevc_h_loop_filter_chroma = c->hevc_h_loop_filter_chroma_c = t_h_chroma;
    c->hevc_v_loop_filter_chroma = c->hevc_v_loop_filter_chroma_c = t_v_chroma;
}

extern "C" void ohevc_videodsp_init_hip(ohevc_VideoDSPContext *c, int)
{
    if (!c) return;
    c->emulated_edge_mc = t_emulated_edge_mc;
    c->emulated_edge_up_h = t_emulated_edge_up_h;
    c->emulated_edge_up_v = t_emulated_edge_up_v;
}

extern "C" int ohevc_tables_bind(ohevc_ctx *ctx)
{
    tl_ctx = ctx;
    tl_state = ctx ? state_of(ctx, true) : nullptr;
    tl_pend = Pending();
    return OHEVC_OK;
}

extern "C" int ohevc_tables_upsample_frame(const uint8_t *el_data0, const uint8_t *bl_data0, const ohevc_HEVCWindow *w, const ohevc_UpsamplInf *u)
{
    Loc le, lb;
    if (!tl_ctx || !w || !u || !locate(el_data0, le) || !locate(bl_data0, lb)) { fail(OHEVC_ERR_STATE); return OHEVC_ERR_STATE; }
    int rc = upsample_once(le.slot, lb.slot, w, u, 0);
    if (rc != OHEVC_OK) fail(rc);
    return rc;
}

extern "C" int ohevc_tables_host_planes(ohevc_ctx *ctx, int slot, uint8_t *data[3], int linesize[3])
{
    TablesState *s = state_of(ctx, false);
    if (!s || !data || !linesize) return OHEVC_ERR_ARG;
    for (int i = 0; i < s->npics(); i++) {
        HostPic hp;
        if (s->pics[i].slot() != slot || !s->pics[i].snapshot(hp) || hp.slot != slot) continue;
        for (int c = 0; c < 3; c++) { data[c] = hp.data[c]; linesize[c] = hp.linesize[c]; }
        return OHEVC_OK;
    }
    return OHEVC_ERR_STATE;
}

extern "C" int ohevc_tables_bs_calls(const ohevc_bs_call *calls, int n)
{
    if (!tl_ctx) { fail(OHEVC_ERR_STATE); return OHEVC_ERR_STATE; }
    const int rc = ohevc_rec_bs_calls(tl_ctx, calls, n);
    if (rc != OHEVC_OK) fail(rc);
    return rc;
}

extern "C" int ohevc_tables_keep_motion(ohevc_ctx *ctx, int log2_min_pu_size)
{
    const int rc = ohevc_frame_keep_motion(ctx, log2_min_pu_size);
    if (rc != OHEVC_OK) fail(rc);
    return rc;
}

extern "C" int ohevc_tables_bs_call(int x0, int y0, int log2_size, int flags)
{
    if (!tl_ctx) { fail(OHEVC_ERR_STATE); return OHEVC_ERR_STATE; }
    const int rc = ohevc_rec_bs_call(tl_ctx, x0, y0, log2_size, flags);
    if (rc != OHEVC_OK) fail(rc);
    return rc;
}

extern "C" int ohevc_tables_cross_component(int res_scale_val)
{
    tl_pend.cross_scale = res_scale_val;
    return OHEVC_OK;
}

extern "C" int ohevc_tables_set_concurrent(ohevc_ctx *ctx, int on)
{
    TablesState *st = state_of(ctx, true);
    if (!st) return OHEVC_ERR_ARG;
    st->concurrent = on != 0;
    return ohevc_ctx_set_concurrent(ctx, on);
}

extern "C" int ohevc_tables_register_picture(ohevc_ctx *ctx, int slot, uint8_t *const data[3], const int linesize[3])
{
    using namespace ohevc;
    OHEVC_REQUIRE(ctx != nullptr && data != nullptr && linesize != nullptr, "null argument");
    int w, h, cfi, bd;
    int rc = ohevc_pic_info(ctx, slot, &w, &h, &cfi, &bd);
    if (rc != OHEVC_OK) return rc;
    TablesState *s = state_of(ctx, true);
    HostPic hp;
    hp.slot = slot; hp.bd = bd; hp.ps = bd > 8 ? 2 : 1;
    for (int c = 0; c < 3; c++) {
        const int hs = c ? (cfi == 1 || cfi == 2) : 0, vs = c ? (cfi == 1) : 0;
        hp.data[c] = data[c]; hp.linesize[c] = linesize[c]; hp.w[c] = w >> hs; hp.h[c] = h >> vs;
    }
    OHEVC_REQUIRE(linesize[0] > 0 && linesize[1] > 0 && linesize[2] > 0, "host planes must have positive line sizes");
    hp.finish();
    static const bool trace = ohevc::config().trace_reg;
    std::lock_guard<std::mutex> g(s->reg->m);
    const int n = s->npics();
    // host memory belongs to one live picture: whoever else still lists memory overlapping these planes is a dead picture
    // whose buffers went back to the decoder's pool (the pools are per plane, so luma/chroma pairs get re-mixed, and the
    // plane's start inside a recycled buffer may differ by an alignment offset: compare ranges, not base pointers)
    for (int i = 0; i < n; i++) {
        const HostPic &o = s->pics[i].v;                 // writers are serialised by the lock: a plain view is consistent here
        if (o.slot < 0 || o.slot == slot) continue;
        bool dead = false;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                if (o.data[a] && hp.data[b] && o.data[a] < hp.data[b] + hp.bytes[b] && hp.data[b] < o.data[a] + o.bytes[a]) {
                    if (trace) fprintf(stderr, "reg: slot %d takes plane %d of entry %d (slot %d, its plane %d)\n", slot, b, i, o.slot, a);
                    dead = true;
                }
        if (dead) s->pics[i].retire();
    }
    s->upsampled.erase(std::remove(s->upsampled.begin(), s->upsampled.end(), slot), s->upsampled.end());
    if (trace) fprintf(stderr, "reg: slot %d = %p %p %p\n", slot, (void *)hp.data[0], (void *)hp.data[1], (void *)hp.data[2]);
    for (int i = 0; i < n; i++) if (s->pics[i].v.slot == slot) { s->pics[i].publish(hp); return OHEVC_OK; }
    for (int i = 0; i < n; i++) if (s->pics[i].v.slot < 0) { s->pics[i].publish(hp); return OHEVC_OK; }
    OHEVC_REQUIRE(n <= OHEVC_MAX_PICTURES, "too many registered pictures");
    s->pics[n].publish(hp);
    s->reg->n.store(n + 1, std::memory_order_release);
    return OHEVC_OK;
}

extern "C" int ohevc_tables_unregister_picture(ohevc_ctx *ctx, int slot)
{
    TablesState *s = state_of(ctx, false);
    if (!s) return OHEVC_OK;
    std::lock_guard<std::mutex> g(s->reg->m);
    for (int i = 0; i < s->npics(); i++)
        if (s->pics[i].v.slot == slot) {
            s->pics[i].retire();
            if (s->cur == (int)i) s->cur = -1;
        }
    return OHEVC_OK;
}

extern "C" int ohevc_tables_begin_frame(ohevc_ctx *ctx, int slot)
{
    using namespace ohevc;
    TablesState *s = state_of(ctx, false);
    OHEVC_REQUIRE(s != nullptr, "no picture registered");
    s->cur = -1;
    for (int i = 0; i < s->npics(); i++) if (s->pics[i].slot() == slot) s->cur = (int)i;
    OHEVC_REQUIRE(s->cur >= 0, "picture not registered");
    s->status = OHEVC_OK;
    s->seq = 0;
    s->h_edge_seq.clear();
    s->held_sao.clear();
    s->upsampled.clear();
    tl_pend = Pending();
    return ohevc_frame_begin(ctx, slot);
}

static int end_frame_common(ohevc_ctx *ctx, int download, bool async, double *issued_at = nullptr);
extern "C" int ohevc_tables_end_frame(ohevc_ctx *ctx, int download) { return end_frame_common(ctx, download, false); }
// ... and says when the issue ended (CLOCK_MONOTONIC seconds): what follows in the call is the wait for the device and the copy-back
extern "C" int ohevc_tables_end_frame2(ohevc_ctx *ctx, int download, double *issued_at) { return end_frame_common(ctx, download, false, issued_at); }
// the frame end handed to the store's issuer thread (ohevc_frame_end_async): returns at once; the copy-back (download != 0) is queued behind
// the picture's device work and ohevc_tables_fetch_picture waits for it
extern "C" int ohevc_tables_end_frame_async(ohevc_ctx *ctx, int download) { return end_frame_common(ctx, download, true); }

extern "C" int ohevc_tables_fetch_picture(ohevc_ctx *ctx, int slot) { return ohevc_pic_wait_host(ctx, slot); }

static int end_frame_common(ohevc_ctx *ctx, int download, bool async, double *issued_at)
{
    using namespace ohevc;
    TablesState *s = state_of(ctx, false);
    OHEVC_REQUIRE(s != nullptr && s->cur >= 0, "no frame begun");
    if (s->status != OHEVC_OK) {
        ohevc_frame_abort(ctx);           // publish the picture as failed: frame threads that reference it must not wait for it
        set_error("a table call failed while recording (unknown pointer or call order)");
        return s->status;
    }
    Prof prof_(K_END);
    int rc;
    // a held SAO job saw, in the reference, the samples right of its block BEFORE a horizontal edge through them was
    // filtered iff that edge's table call came after the SAO call (ohevc_hip.h, OHEVC_SAO_LAG_*)
    const HostPic &cur = s->pics[s->cur].v;
    for (auto &held : s->held_sao) {
        ohevc_sao_job &j = held.first;
        const int xr = j.x + j.w;
        if (xr < cur.w[j.plane]) {
            auto later = [&](int y) {
                auto it = s->h_edge_seq.find(((uint32_t)j.plane << 30) | ((uint32_t)y << 15) | (uint32_t)xr);
                return it != s->h_edge_seq.end() && it->second > held.second;
            };
            j.quirks = (uint8_t)((later(j.y + j.h) ? OHEVC_SAO_LAG_BELOW : 0) | (later(j.y) ? OHEVC_SAO_LAG_ABOVE : 0) |
                                 ((j.h > 8 && later(j.y + 8)) ? OHEVC_SAO_LAG_MID : 0));
        }
        if ((rc = ohevc_rec_sao(ctx, &j)) != OHEVC_OK) { ohevc_frame_abort(ctx); return rc; }
    }
    s->held_sao.clear();
    if (async) {
        const HostPic &hp = s->pics[s->cur].v;
        void *const host[3] = { hp.data[0], hp.data[1], hp.data[2] };
        const ptrdiff_t strides[3] = { hp.linesize[0], hp.linesize[1], hp.linesize[2] };
        return ohevc_frame_end_async(ctx, download ? host : nullptr, strides);
    }
    // no copy-back inside this call: the frame may be parked instead of making this thread wait for other threads' frame ends (ohevc_ctx.h)
    // (download 2: the copy-back is queued behind the frame end just issued - ohevc_pic_download_queue - and waited for by ohevc_tables_fetch_picture)
    rc = download ? ohevc_frame_end(ctx) : ohevc_frame_end_deferred(ctx);
    if (issued_at) {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        *issued_at = ts.tv_sec + 1e-9 * ts.tv_nsec;
    }
    if (rc != OHEVC_OK) return rc;
    if (download) {
        const HostPic &hp = s->pics[s->cur].v;
        void *const host[3] = { hp.data[0], hp.data[1], hp.data[2] };
        const ptrdiff_t strides[3] = { hp.linesize[0], hp.linesize[1], hp.linesize[2] };
        if ((rc = download == 2 ? ohevc_pic_download_queue(ctx, hp.slot, host, strides) : ohevc_pic_download_planes(ctx, hp.slot, host, strides)) != OHEVC_OK) return rc;
    }
    return OHEVC_OK;
}

extern "C" int ohevc_tables_set_bypass_map(ohevc_ctx *ctx, const uint8_t *is_pcm, int min_pu_width, int min_pu_height, int log2_min_pu_size)
{
    // the same switch that keeps the filter lag of the reference front-end decides between its partial restore and the standard's
    TablesState *st = state_of(ctx, false);
    return ohevc_frame_set_bypass_map(ctx, is_pcm, min_pu_width, min_pu_width, min_pu_height, log2_min_pu_size, st && st->lag_on);
}

// called by ohevc_ctx_destroy: a later ctx may be allocated at the same address and must not inherit this registry
extern "C" void ohevc_tables_forget(ohevc_ctx *ctx)
{
    if (g_prof_on) {
        for (int k = 0; k < K_NFAM; k++)
            if (g_prof_calls[k])
                fprintf(stderr, "slot profile: %-26s %9llu calls %8.2f Mcycles %6.0f cycles/call\n", kFamName[k], g_prof_calls[k],
                        g_prof_cycles[k] * 1e-6, (double)g_prof_cycles[k] / g_prof_calls[k]);
        memset(g_prof_cycles, 0, sizeof(g_prof_cycles));
        memset(g_prof_calls, 0, sizeof(g_prof_calls));
    }
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_states.find(ctx);
    if (it != g_states.end()) {
        if (tl_state == it->second) { tl_state = nullptr; tl_ctx = nullptr; }
        delete it->second;
        g_states.erase(it);
    }
}

extern "C" int ohevc_tables_status(ohevc_ctx *ctx)
{
    TablesState *s = state_of(ctx, false);
    // a table call from a thread nobody bound produced no job: never let that pass silently
    if (g_unbound_calls.exchange(0, std::memory_order_relaxed) > 0) {
        ohevc::set_error("table slots were called from a thread without ohevc_tables_bind (slice threads need a bind per worker)");
        if (s && s->status == OHEVC_OK) s->status = OHEVC_ERR_STATE;
        return OHEVC_ERR_STATE;
    }
    return s ? s->status : OHEVC_OK;
}

extern "C" int ohevc_tables_intra_pred(const ohevc_intra_geom *geom, int x0, int y0, int log2_size, int c_idx, int mode,
                                       int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right)
{
    if (!tl_ctx) { fail(OHEVC_ERR_STATE); return OHEVC_ERR_STATE; }
    ohevc_intra_job j;
    int rc = ohevc_intra_make_job(geom, x0, y0, log2_size, c_idx, mode, cand_bottom_left, cand_left, cand_up_left, cand_up, cand_up_right, &j);
    if (rc == OHEVC_OK) rc = ohevc_rec_intra(tl_ctx, &j);
    if (rc != OHEVC_OK) fail(rc);
    return rc;
}

extern "C" int ohevc_tables_intra_pred_cip(const ohevc_intra_geom *geom, int log2_min_pu_size, const uint8_t *pred_flag,
                                           ptrdiff_t pred_flag_stride, int intra_value, int x0, int y0, int log2_size, int c_idx, int mode,
                                           int cand_bottom_left, int cand_left, int cand_up_left, int cand_up, int cand_up_right)
{
    Prof prof_(K_INTRA);
    if (!tl_ctx) { fail(OHEVC_ERR_STATE); return OHEVC_ERR_STATE; }
    ohevc_intra_job j;
    ohevc_intra_cip cip;
    int rc = ohevc_intra_make_job_cip(geom, log2_min_pu_size, pred_flag, pred_flag_stride, intra_value, x0, y0, log2_size, c_idx, mode,
                                      cand_bottom_left, cand_left, cand_up_left, cand_up, cand_up_right, &j, &cip);
    if (rc == OHEVC_OK) rc = ohevc_rec_intra_cip(tl_ctx, &j, &cip);
    if (rc != OHEVC_OK) fail(rc);
    return rc;
}
