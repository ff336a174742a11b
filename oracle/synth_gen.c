/*
 * oracle/synth_gen.c -- TEST INFRASTRUCTURE ONLY: the slice-data half of the HEVC stream synthesiser.
 *
 * No HEVC bitstream exists in this environment and the reference ships none, so streams are made here:
 *   - parameter sets and slice headers are written by oracle/pystream.py (plain bit writing, ITU-T H.265 7.3.2 / 7.3.6
 *     in the form the reference parses them, hevc_ps.c:1097-2400, hevc.c:520-1050);
 *   - slice DATA is produced by running the reference's own parser (hevc.c:2300-2600 coding quadtree,
 *     hevc_cabac.c:659-1950 syntax elements) with the five CABAC primitives replaced (synth_hooks.h): each requested
 *     bin is drawn from a seeded generator with a per-context bias and simultaneously fed to the arithmetic ENCODER
 *     below (H.265 9.3.4.x encoding process: low/range/outstanding bits, rangeTabLps, transIdxLps; flush after
 *     terminate bins, byte alignment, raw pcm_sample bytes).
 * The bytes collected per slice are returned to Python, which appends them to the slice header and escapes the NAL.
 * The result is pinned by decoding it with the UNTOUCHED reference decoder (_ref/libopenhevc_c.so): its output frames
 * must equal the frames this generator reconstructed while generating (tests/test_stream_cpu.py).
 *
 * The context states live in the reference's HEVCLocalContext (cabac_state[], initialised by the reference's own
 * cabac_init_state(), hevc_cabac.c:582-604, format (pStateIdx << 1) | valMps); only the transition rule is restated.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "libavcodec/hevc.h"

/* ---- H.265 Table 9-46 (rangeTabLps) and Table 9-47 (transIdxLps); checked against the reference's packed
 *      ff_h264_cabac_tables by tests/test_stream_cpu.py::test_cabac_tables_match_reference */
static const uint8_t range_tab_lps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195},
    {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166}, { 95, 116, 137, 158}, { 90, 110, 130, 150},
    { 85, 104, 123, 142}, { 81,  99, 117, 135}, { 77,  94, 111, 128}, { 73,  89, 105, 122}, { 69,  85, 100, 116},
    { 66,  80,  95, 110}, { 62,  76,  90, 104}, { 59,  72,  86,  99}, { 56,  69,  81,  94}, { 53,  65,  77,  89},
    { 51,  62,  73,  85}, { 48,  59,  69,  80}, { 46,  56,  66,  76}, { 43,  53,  63,  72}, { 41,  50,  59,  69},
    { 39,  48,  56,  65}, { 37,  45,  54,  62}, { 35,  43,  51,  59}, { 33,  41,  48,  56}, { 32,  39,  46,  53},
    { 30,  37,  43,  50}, { 29,  35,  41,  48}, { 27,  33,  39,  45}, { 26,  31,  37,  43}, { 24,  30,  35,  41},
    { 23,  28,  33,  39}, { 22,  27,  32,  37}, { 21,  26,  30,  35}, { 20,  24,  29,  33}, { 19,  23,  27,  31},
    { 18,  22,  26,  30}, { 17,  21,  25,  28}, { 16,  20,  23,  27}, { 15,  19,  22,  25}, { 14,  18,  21,  24},
    { 14,  17,  20,  23}, { 13,  16,  19,  22}, { 12,  15,  18,  21}, { 12,  14,  17,  20}, { 11,  14,  16,  19},
    { 11,  13,  15,  18}, { 10,  12,  15,  17}, { 10,  12,  14,  16}, {  9,  11,  13,  15}, {  9,  11,  12,  14},
    {  8,  10,  12,  14}, {  8,   9,  11,  13}, {  7,   9,  11,  12}, {  7,   9,  10,  12}, {  7,   8,  10,  11},
    {  6,   8,   9,  11}, {  6,   7,   9,  10}, {  6,   7,   8,   9}, {  2,   2,   2,   2},
};
static const uint8_t trans_idx_lps[64] = {
     0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22,
    22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36,
    37, 37, 37, 38, 38, 63,
};
const uint8_t *ohsyn_table_range_lps(void) { return &range_tab_lps[0][0]; }
const uint8_t *ohsyn_table_trans_lps(void) { return trans_idx_lps; }

/* ---- bit sink ---- */
typedef struct {
    uint8_t *buf;
    size_t   cap, bits;
} BitSink;

static void sink_put(BitSink *b, int bit)
{
    if ((b->bits >> 3) >= b->cap) {
        size_t ncap = b->cap ? b->cap * 2 : 1 << 16;
        b->buf = realloc(b->buf, ncap);
        memset(b->buf + b->cap, 0, ncap - b->cap);
        b->cap = ncap;
    }
    if (bit)
        b->buf[b->bits >> 3] |= 0x80 >> (b->bits & 7);
    b->bits++;
}
static void sink_align_zero(BitSink *b) { while (b->bits & 7) sink_put(b, 0); }
static void sink_bytes(BitSink *b, const uint8_t *p, int n)
{
    int i, k;
    for (i = 0; i < n; i++)
        for (k = 7; k >= 0; k--)
            sink_put(b, (p[i] >> k) & 1);
}

/* ---- arithmetic encoder (H.265 9.3.4.x "encoding process", informative) ---- */
typedef struct {
    uint32_t low, range;
    int      outstanding, first;
} ArithEnc;

static void enc_start(ArithEnc *e) { e->low = 0; e->range = 510; e->outstanding = 0; e->first = 1; }

static void enc_put(ArithEnc *e, BitSink *b, int bit)
{
    if (e->first)
        e->first = 0;
    else
        sink_put(b, bit);
    while (e->outstanding > 0) {
        sink_put(b, !bit);
        e->outstanding--;
    }
}

static void enc_renorm(ArithEnc *e, BitSink *b)
{
    while (e->range < 256) {
        if (e->low < 256) {
            enc_put(e, b, 0);
        } else if (e->low >= 512) {
            e->low -= 512;
            enc_put(e, b, 1);
        } else {
            e->low -= 256;
            e->outstanding++;
        }
        e->range <<= 1;
        e->low   <<= 1;
    }
}

static void enc_decision(ArithEnc *e, BitSink *b, uint8_t *state, int bin)
{
    int p = *state >> 1, mps = *state & 1;
    uint32_t lps = range_tab_lps[p][(e->range >> 6) & 3];
    e->range -= lps;
    if (bin != mps) {
        e->low  += e->range;
        e->range = lps;
        if (p == 0)
            mps = !mps;
        p = trans_idx_lps[p];
    } else if (p < 62) {
        p++;
    }
    *state = (uint8_t)((p << 1) | mps);
    enc_renorm(e, b);
}

static void enc_bypass(ArithEnc *e, BitSink *b, int bin)
{
    e->low <<= 1;
    if (bin)
        e->low += e->range;
    if (e->low >= 1024) {
        enc_put(e, b, 1);
        e->low -= 1024;
    } else if (e->low < 512) {
        enc_put(e, b, 0);
    } else {
        e->low -= 512;
        e->outstanding++;
    }
}

/* terminate bin; bin == 1 also flushes: the last bit written (always 1) is the rbsp_stop_one_bit / alignment bit */
static void enc_terminate(ArithEnc *e, BitSink *b, int bin)
{
    e->range -= 2;
    if (bin) {
        e->low  += e->range;
        e->range = 2;
        enc_renorm(e, b);
        enc_put(e, b, (e->low >> 9) & 1);
        sink_put(b, (e->low >> 8) & 1);
        sink_put(b, 1);
    } else {
        enc_renorm(e, b);
    }
}

/* ---- generator state (single-threaded test tool: one global instance) ---- */
#define OHSYN_MAX_SLICES   64
#define OHSYN_MAX_SUBSTR   4096

static struct {
    uint64_t rng;
    uint16_t prob[HEVC_CONTEXTS];       /* P(bin == 1) * 65536 per context index */
    uint16_t bypass_prob;
    uint16_t pcm_prob;
    ArithEnc enc;
    BitSink  sink[OHSYN_MAX_SLICES];    /* payload of slice segment k of the current access unit */
    int      nslices;                   /* slice segments started in this access unit */
    int      planned[OHSYN_MAX_SLICES]; /* CTUs in slice segment k */
    int      ctus_done;
    int      ended[OHSYN_MAX_SLICES];   /* end_of_slice_segment_flag = 1 was produced for segment k */
    int      flushed;                   /* the encoder was flushed and nothing has been coded since */
    int      nsub[OHSYN_MAX_SLICES];
    uint32_t sub_start[OHSYN_MAX_SLICES][OHSYN_MAX_SUBSTR]; /* byte offset where substream j of slice k starts */
    uint8_t  scratch[2 * 64 * 64 * 2 + 64];  /* raw pcm bytes of one CU (the reference reads them through a pointer) */
    uint64_t nbins, nbypass;
    int      error;
    int      last_ref_idx;              /* SHVC: the ref_idx_lX parsed in front of the mvd_coding() in progress */
    int      zero_mvd;                  /* SHVC: abs_mvd_greater0_flag bins are 0 (vector into the inter-layer reference picture) */
} G;

static uint32_t rnd16(void)
{
    /* xorshift64* */
    G.rng ^= G.rng >> 12;
    G.rng ^= G.rng << 25;
    G.rng ^= G.rng >> 27;
    return (uint32_t)((G.rng * 0x2545F4914F6CDD1DULL) >> 48);
}

static BitSink *cur_sink(void)
{
    if (G.nslices <= 0) {
        G.error = 1;
        return &G.sink[0];
    }
    return &G.sink[G.nslices - 1];
}

void ohsyn_reset(uint64_t seed)
{
    int i;
    G.rng = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    if (!G.rng)
        G.rng = 1;
    for (i = 0; i < HEVC_CONTEXTS; i++)
        G.prob[i] = 32768;
    G.bypass_prob = 32768;
    G.pcm_prob = 3277;
    G.error = 0;
}

/* prob[i] in [0,1]: P(bin = 1) for context index i (layout: elem_offset[], hevc_cabac.c:98-155) */
void ohsyn_set_probs(const float *prob, int n, float bypass, float pcm)
{
    int i;
    for (i = 0; i < n && i < HEVC_CONTEXTS; i++)
        G.prob[i] = (uint16_t)(prob[i] * 65535.0f);
    G.bypass_prob = (uint16_t)(bypass * 65535.0f);
    G.pcm_prob = (uint16_t)(pcm * 65535.0f);
}

/* before each access unit: how many CTUs each slice segment of the picture will hold (in decoding order) */
void ohsyn_begin_au(const int *ctus_per_slice, int n)
{
    int i;
    for (i = 0; i < OHSYN_MAX_SLICES; i++) {
        G.sink[i].bits = 0;
        if (G.sink[i].buf)
            memset(G.sink[i].buf, 0, G.sink[i].cap);
        G.nsub[i] = 0;
        G.planned[i] = i < n ? ctus_per_slice[i] : 0;
        G.ended[i] = 0;
    }
    G.nslices = 0;
    G.error = 0;
}

int ohsyn_num_slices(void) { return G.error ? -1 : G.nslices; }
int ohsyn_slice_payload(int k, const uint8_t **p)
{
    if (k < 0 || k >= G.nslices || (G.sink[k].bits & 7) || !G.ended[k])
        return -1;      /* the parser gave up inside this segment: the random syntax was not decodable */
    *p = G.sink[k].buf;
    return (int)(G.sink[k].bits >> 3);
}
int ohsyn_slice_substreams(int k, const uint32_t **starts)
{
    if (k < 0 || k >= G.nslices)
        return -1;
    *starts = G.sub_start[k];
    return G.nsub[k];
}
void ohsyn_stats(uint64_t *bins, uint64_t *bypass) { *bins = G.nbins; *bypass = G.nbypass; }

static void new_substream(void)
{
    int k = G.nslices - 1;
    if (k >= 0 && G.nsub[k] < OHSYN_MAX_SUBSTR)
        G.sub_start[k][G.nsub[k]++] = (uint32_t)(cur_sink()->bits >> 3);
    enc_start(&G.enc);
    G.flushed = 1;
}

/* ---- the five primitives the reference's parser calls (synth_hooks.h) ---- */

/* slice (segment) start: cabac_init_decoder(), hevc_cabac.c:572-580 */
void ohsyn_init_decoder(CABACContext *c, const uint8_t *buf, int buf_size)
{
    (void)c; (void)buf; (void)buf_size;
    if (G.nslices >= OHSYN_MAX_SLICES) {
        G.error = 1;
        return;
    }
    G.nslices++;
    G.ctus_done = 0;
    new_substream();
}

int ohsyn_bin(CABACContext *c, uint8_t *state)
{
    HEVCLocalContext *lc = (HEVCLocalContext *)((uint8_t *)c - offsetof(HEVCLocalContext, cc));
    ptrdiff_t idx = state - lc->cabac_state;
    int bin = rnd16() < ((idx >= 0 && idx < HEVC_CONTEXTS) ? G.prob[idx] : 32768u);
    if (G.zero_mvd)                 /* the only context-coded bins of mvd_coding() that matter: abs_mvd_greater0_flag[0..1] (hevc_cabac.c:898-900) */
        bin = 0;
    enc_decision(&G.enc, cur_sink(), state, bin);
    G.flushed = 0;
    G.nbins++;
    return bin;
}

int ohsyn_bypass(CABACContext *c)
{
    int bin = rnd16() < G.bypass_prob;
    (void)c;
    enc_bypass(&G.enc, cur_sink(), bin);
    G.flushed = 0;
    G.nbypass++;
    return bin;
}

static void terminate_and_align(void)
{
    enc_terminate(&G.enc, cur_sink(), 1);
    sink_align_zero(cur_sink());
    G.flushed = 1;
}

/* the only direct get_cabac_terminate() left once end_of_slice / pcm_flag are renamed: end_of_subset_one_bit at the
 * start of a WPP row (ff_hevc_cabac_init, hevc_cabac.c:636).  Its value must be 1. */
int ohsyn_terminate(CABACContext *c)
{
    (void)c;
    terminate_and_align();
    return 1;
}

/* n == 0: cabac_reinit() at a WPP row / tile start (hevc_cabac.c:567-570); n > 0: pcm_sample bytes (hevc.c:1603) */
const uint8_t *ohsyn_skip_bytes(CABACContext *c, int n)
{
    int i;
    (void)c;
    if (!G.flushed)                 /* tile change: the reference skips end_of_subset_one_bit implicitly */
        terminate_and_align();
    if (n > (int)sizeof(G.scratch)) {
        G.error = 1;
        n = sizeof(G.scratch);
    }
    if (n > 0) {
        for (i = 0; i < n; i++)
            G.scratch[i] = (uint8_t)rnd16();
        sink_bytes(cur_sink(), G.scratch, n);
        enc_start(&G.enc);          /* 9.3.2.5: the engine restarts after the pcm samples, same substream */
        G.flushed = 1;
    } else {
        new_substream();
    }
    return G.scratch;
}

/* end_of_slice_segment_flag (hevc.c:2582): 1 exactly when the planned number of CTUs has been produced */
int ohsyn_end_of_slice_flag(HEVCContext *s)
{
    int k = G.nslices - 1, last;
    (void)s;
    G.ctus_done++;
    last = k < 0 || G.ctus_done >= G.planned[k];
    if (last) {
        if (k >= 0)
            G.ended[k] = 1;
        terminate_and_align();      /* the flush's final 1 bit is rbsp_stop_one_bit, then rbsp_alignment_zero_bits */
    } else {
        enc_terminate(&G.enc, cur_sink(), 0);
        G.flushed = 0;
    }
    return last;
}

/* pcm_flag (hevc.c:2415) */
int ohsyn_pcm_flag(HEVCContext *s)
{
    int bin = rnd16() < G.pcm_prob;
    (void)s;
    if (bin) {
        terminate_and_align();      /* pcm_alignment_zero_bits; ohsyn_skip_bytes(n) follows */
    } else {
        enc_terminate(&G.enc, cur_sink(), 0);
        G.flushed = 0;
    }
    return bin;
}

/* ---- SHVC: ref_idx_lX and mvd_coding() (hevc.c:2027-2060).  In an enhancement-layer slice a prediction unit that names the inter-layer
 *      reference picture gets a zero motion vector difference; its predictor is zero already (every vector into that picture is, and a
 *      candidate never crosses from a short-term to a long-term reference: hevc_mvs.c) ---- */
int ohsyn_ref_idx_lx(HEVCContext *s, int num_ref_idx_lx)
{
    G.last_ref_idx = ff_hevc_ref_idx_lx_decode(s, num_ref_idx_lx);
    return G.last_ref_idx;
}

void ohsyn_mvd_coding(HEVCContext *s, int x0, int y0, int log2_cb_size /* the reference passes the list here */)
{
    int lx = log2_cb_size;
    if (s->nuh_layer_id && s->inter_layer_ref && s->ref && s->ref->refPicList[s->slice_idx] && (lx == 0 || lx == 1) &&
        s->ref->refPicList[s->slice_idx][lx].ref[G.last_ref_idx] == s->inter_layer_ref)
        G.zero_mvd = 1;
    ff_hevc_hls_mvd_coding(s, x0, y0, log2_cb_size);
    G.zero_mvd = 0;
    G.last_ref_idx = 0;
}
