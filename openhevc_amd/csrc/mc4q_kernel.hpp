// mc4q_kernel.hpp -- four blocks of at most 8x8 samples in the matrix-core tile of mc4_kernel (included by mc_kernels.hip inside namespace ohevc,
// behind mc4_kernel.hpp whose operand tables, plane split and loaders it uses).
//
// Most prediction blocks of a real stream are small: 8x8 / 8x4 / 4x8 luma blocks and ALL 4:2:0 chroma of luma blocks up to 16x16.  Through
// mc4_kernel each of them costs a whole 16x16 tile - two window loads per lane, four MFMAs - for a quarter of its samples (13 % of the HBM
// roofline on 8x8 blocks where 16x16 blocks reach 25-30 %).  The window of a block of at most 8x8 samples is at most 15x15, so a tile holds four:
//
//   pass 1, pair (a, b)     the 8 K slots of slot group g are 8 window columns of ONE block - groups 0, 1: columns 0..15 of a, groups 2, 3: of b - and
//                           the constant operand is block diagonal: output column n < 8 is x = n of a (taps on a's slots only), n >= 8 is x = n - 8
//                           of b.  The 15 window rows are the 16 rows of ONE MFMA; the second MFMA of the tile takes the pair (c, d).
//   pass 2, per pair        the pass-1 result of a pair sits in K slots 0..3 of every slot group (rows 4g .. 4g + 3), slots 4..7 multiply zeros; column
//                           y < 8 carries the vertical taps of one block of the pair, y >= 8 of the other.  Of the 16 x 16 results the two quadrants
//                           where the row's block is the column's block are that pair's blocks.  Pair (a, b) is laid out a: (x < 8, y < 8),
//                           b: (x >= 8, y >= 8) and pair (c, d) the other way round - c: (x < 8, y >= 8), d: (x >= 8, y < 8) - so that EVERY lane of the
//                           result layout (lane = y, registers = 4 consecutive x) ends up with four samples of exactly one block.
//
// Per block everything may differ (plane, phases, references, weights): what mc4_kernel keeps in scalars per tile every lane fetches for the
// block its role works for (see the kernel).
// Exactness: the same int8-plane products with the same constants as mc4_kernel (every output column still sees one complete filter whose taps sum to 64).

// The plane record of a lane's reference picture (the four blocks of a quad may predict from different pictures): a vector load per lane.
struct Mc4qRef { unsigned long w0, w1, w2; };                    // data | stride, width | height, -
__device__ __forceinline__ Mc4qRef mc4q_ref(const ohevc_plane *refs, int ref, int plane)
{
    typedef const MC4_GLOBAL unsigned long *lptr;
    lptr pr = (lptr)(refs + (3 * ref + plane));
    return Mc4qRef{ pr[0], pr[1], pr[2] };
}
// 8 samples of window row r of the lane's block, columns 8 * half ..
// 8-bit samples: the window starts wherever the motion vector says; the lane loads the three ALIGNED dwords that hold its 8 samples and
// returns the byte offset, mc4q_finish shifts (v_alignbyte_b32): 1.5-2 % on the 8x8 rows (profiles/r5v_*).  The same for 16-bit samples
// (dwordx4 + dword at a multiple of 4 instead of one dwordx4 at a multiple of 2) ran 3 % slower and needs a 65th register: not done.
template <typename Pixel>
__device__ __forceinline__ unsigned mc4q_issue(const Mc4qRef &rec, int wx0, int wy0, int wh, int r, int half, unsigned (&out)[sizeof(Pixel) == 2 ? 4 : 3])
{
    mc4_gptr base = (mc4_gptr)rec.w0;
    const unsigned stride = (unsigned)rec.w1;
    const int xmax = (int)(rec.w1 >> 32) - 1, ymax = (int)(unsigned)rec.w2 - 1;
    const int col0 = wx0 + 8 * half;
    int wr = r < wh ? r : wh - 1;
    int y = wy0 + wr;
    y = y < 0 ? 0 : y > ymax ? ymax : y;
    const unsigned rowoff = __umul24((unsigned)y, stride);
    if (sizeof(Pixel) == 1 && col0 >= 4 && col0 + 11 <= xmax) {
        const unsigned long a = (unsigned long)(base + (rowoff + (unsigned)col0));
        __builtin_memcpy(out, (const void *)(mc4_gptr)(a & ~3ul), 12);
        return (unsigned)a & 3u;
    }
    if (col0 >= 0 && col0 + 7 <= xmax) {
        __builtin_memcpy(out, (const void *)(base + (rowoff + (unsigned)col0 * (unsigned)sizeof(Pixel))), sizeof(Pixel) == 2 ? 16 : 8);
    } else {
        const u32x4 e = mc4_gather_edge<Pixel>(base + rowoff, col0, xmax);
        out[0] = e.x; out[1] = e.y;
        if (sizeof(Pixel) == 2) { out[2] = e.z; out[3] = e.w; }
    }
    if (sizeof(Pixel) == 1) out[2] = 0;
    return 0;
}

// one reference of a quad: raw[pair] = the lane's 8 samples (memory-side lane map) -> v[k] = the 14-bit intermediate of the lane's block at
// (x = 4 (g & 1) + k, y = lane & 7), g = lane >> 4.  b1[pair] / b2[pair]: the lane's constant operands (see the kernel).
template <typename Pixel>
__device__ __forceinline__ void mc4q_finish(const unsigned (&raw)[2][sizeof(Pixel) == 2 ? 4 : 3], const unsigned (&shift)[2], const u32x2 (&b1)[2], const unsigned (&b2)[2], int bit_depth, int lane,
                                            unsigned (&seen)[2], int *v)
{
    constexpr bool WIDE = sizeof(Pixel) == 2;
    const mc4_v4i zero = { 0, 0, 0, 0 };
    const int src = (lane & 15) * 4 + (lane >> 4);               // operand-side lane (row lane & 15, slot group lane >> 4) <- memory-side lane
    const int K2 = 8192 + (WIDE ? 0 : 8192 * 64);
    mc4_v4i e[2];
#pragma unroll
    for (int pair = 0; pair < 2; pair++) {
        unsigned w[WIDE ? 4 : 2];
#pragma unroll
        for (int k = 0; k < (WIDE ? 4 : 2); k++)
            w[k] = (unsigned)__shfl((int)(WIDE ? raw[pair][k] : __builtin_amdgcn_alignbyte(raw[pair][k + 1], raw[pair][k], shift[pair])), src);
        const long bop1 = mc4_op(b1[pair].x, b1[pair].y);
        mc4_v4i d;
        int hv[4];
        if (WIDE) {
            seen[pair] = w[0] | w[1] | w[WIDE ? 2 : 0] | w[WIDE ? 3 : 1];
            d = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_hi(w[0], w[1]), mc4_hi(w[WIDE ? 2 : 0], w[WIDE ? 3 : 1])), bop1, zero, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = (d[k] << 8) + 8192;
            d = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_lo(w[0], w[1]), mc4_lo(w[WIDE ? 2 : 0], w[WIDE ? 3 : 1])), bop1, d, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) hv[k] = d[k] >> (bit_depth - 8);
        } else {
            d = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(w[0] ^ 0x80808080u, w[1] ^ 0x80808080u), bop1, zero, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) hv[k] = d[k];
        }
        // (columns of the other block of the pair multiplied zeros: hv there is 0 resp. the bare plane offset - finite, and never looked at)
        const unsigned p0 = mc4_pk16(hv[0], hv[1]), p1 = mc4_pk16(hv[2], hv[3]);
        const long bop2 = mc4_op(b2[pair], 0u);                  // K slots 4..7 of every group: zero taps (and zero samples)
        e[pair] = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_hi(p0, p1), 0u), bop2, zero, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) e[pair][k] = (e[pair][k] << 8) + K2;
        e[pair] = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_lo(p0, p1), 0x80808080u ^ 0x80808080u), bop2, e[pair], 0, 0, 0);
    }
    const bool own = ((lane & 15) < 8) == ((lane >> 4) < 2);     // quadrants of pair (a, b); the other two hold pair (c, d)
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (own ? e[0][k] : e[1][k]) >> 6;
}

// 256 threads = 4 wavefronts (one LDS copy of the operand tables); a wavefront takes UNITS quads (quad u of the wavefront: jobs 4q .. 4q + 3,
// q = first + u) and has the loads of all of them in flight before it computes the first (mc4_kernel).
#ifdef OHEVC_HIPEMU
#define MC4Q_OCCUPANCY(units)
#else
#define MC4Q_OCCUPANCY(units) __attribute__((amdgpu_waves_per_eu((units) == 1 ? 8 : 4, 8)))     // one quad per wavefront: 8 wavefronts per SIMD (<= 64 VGPRs)
#endif
// TWIN (lab build, ohevc_debug_set_mc_variant(102 / 103)): 1 = the kernel's memory traffic without its arithmetic - records, samples and the
// store of SOMETHING derived from the lane's own samples; 2 = its arithmetic without the sample loads (made-up samples; records and store
// as usual).  The two bracket the kernel: what it would cost if either part were free.
template <typename Pixel, int UNITS, int TWIN>
__global__ __launch_bounds__(256) MC4Q_OCCUPANCY(UNITS) void mc4q_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs, int n_ref_slots, const ohevc_mc_job *__restrict__ jobs,
                                                   int njobs, int bit_depth, unsigned *__restrict__ wild_mask)
{
    constexpr bool WIDE = sizeof(Pixel) == 2;
    constexpr int kRefSlots = 64;                                                                // plane records kept in LDS (3 x 24 bytes per picture)
    __shared__ u32x2 tabs[2][12][64];
    __shared__ unsigned long ref_tab[kRefSlots * 9];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The plane records of the reference pictures come with the operand tables, i.e. together with the job records: a quad is then two dependent
    // rounds of loads (records, samples) instead of three (job records, plane records, samples).  (More pictures than kRefSlots: from memory.)
    const bool refs_in_lds = n_ref_slots <= kRefSlots;
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(&kMc4);
        u32x4 *dl = reinterpret_cast<u32x4 *>(&tabs[0][0][0]);
        const u32x4 t0 = src[tid], t1 = src[256 + tid], t2 = src[512 + tid];
        const int words = refs_in_lds ? n_ref_slots * 9 : 0;
        typedef const MC4_GLOBAL unsigned long *lptr;
        unsigned long r0 = 0, r1 = 0, r2 = 0;
        if (tid < words) r0 = ((lptr)refs)[tid];
        if (tid + 256 < words) r1 = ((lptr)refs)[tid + 256];
        if (tid + 512 < words) r2 = ((lptr)refs)[tid + 512];
        dl[tid] = t0; dl[256 + tid] = t1; dl[512 + tid] = t2;
        if (tid < words) ref_tab[tid] = r0;
        if (tid + 256 < words) ref_tab[tid + 256] = r1;
        if (tid + 512 < words) ref_tab[tid + 512] = r2;
        static_assert(kRefSlots * 9 <= 768, "three rounds of 256 words");
    }
    const int per_xcd = gridDim.x >> 3;                                                          // an XCD takes a contiguous eighth of the list (mc4_kernel)
    const int q = (((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3)) * 4 + wave;
    static_assert(UNITS == 1, "one quad per wavefront");
    // The quad's four job records, 32 dwords, one per lane (lane L: dword L & 7 of job 4q + ((L & 31) >> 3); a job behind the list: the last
    // one again, never stored).  Every lane then FETCHES the dwords of the block its role works for through the cross-lane network
    // (ds_bpermute_b32: no LDS memory involved) and unpacks them itself.  Rounds 3's form read the records with scalar loads, unpacked them on
    // the scalar unit and SELECTED per lane, field by field - v_mov + v_cndmask per field and role: 264 of the kernel's 593 vector
    // instructions, and the kernel is bound by vector-instruction issue (SQ_ACTIVE_INST_VALU 95 % of the launch, profiles/r03x_*).
    static_assert(sizeof(ohevc_mc_job) == 32, "eight dwords per job record");
    unsigned jw;
    {
        const int j = 4 * q + ((lane & 31) >> 3);
        typedef const MC4_GLOBAL unsigned *gq;
        jw = ((gq)(jobs + (j < njobs ? j : njobs - 1)))[lane & 7];
    }
    __syncthreads();                                                                             // the tables are in LDS
    if (4 * q >= njobs) return;
    auto field = [&](int block_bytes, int k) -> unsigned { return (unsigned)__builtin_amdgcn_ds_bpermute(block_bytes + 4 * k, (int)jw); };      // dword k of block
    // ---- memory side: lane (r = lane >> 2, g = lane & 3) loads 8 samples of window row r, columns 8 (g & 1) .., of block g >> 1 of each pair
    // (two dependent rounds - job records, samples - each issued for every pair and reference before the first use)
    unsigned raw[2][2][WIDE ? 4 : 3] = {}, shift[2][2] = {};
    // any job of the quad bi-predicted (wave-uniform): dword 1 of the four records
    const bool any_bi = __ballot(lane < 32 && (lane & 7) == 1 && ((jw >> 24) & OHEVC_MC_BI) != 0) != 0;
    {
        const int r = lane >> 2, half = lane & 1;
        const int first_off = (lane & 3) < 2 ? 0 : 32;
        int plane_m[2], h_m[2], sx0[2], sy0[2], sx1[2], sy1[2];
        Mc4qRef rec[2][2];
#pragma unroll
        for (int pair = 0; pair < 2; pair++) {
            const int blk = 64 * pair + first_off;                                               // the lane's block of the pair, as a byte offset into the 32 lanes
            const unsigned d1 = field(blk, 1), d2 = field(blk, 2), d3 = field(blk, 3), d5 = field(blk, 5);
            const bool bi = ((d1 >> 24) & OHEVC_MC_BI) != 0;
            plane_m[pair] = (int)((d1 >> 16) & 0xff); h_m[pair] = (int)((d1 >> 8) & 0xff);
            sx0[pair] = (int)(short)(d2 & 0xffff); sy0[pair] = (int)d2 >> 16;
            sx1[pair] = bi ? (int)(short)(d3 & 0xffff) : sx0[pair]; sy1[pair] = bi ? (int)d3 >> 16 : sy0[pair];       // (uni: loaded again, weighted 0)
            const int ref0 = (int)(signed char)(d5 & 0xff), ref1 = bi ? (int)(signed char)((d5 >> 8) & 0xff) : ref0;
            auto plane_record = [&](int ref) {
                if (!refs_in_lds) return mc4q_ref(refs, ref, plane_m[pair]);
                const unsigned long *e = &ref_tab[(3 * ref + plane_m[pair]) * 3];
                return Mc4qRef{ e[0], e[1], e[2] };
            };
            rec[0][pair] = plane_record(ref0);
            if (any_bi) rec[1][pair] = plane_record(ref1);
        }
#pragma unroll
        for (int pair = 0; pair < 2; pair++) {
            const int before = plane_m[pair] == 0 ? 3 : 1, taps = plane_m[pair] == 0 ? 8 : 4;
            if (TWIN == 2) {                                                                     // no sample loads: something the compiler cannot fold
                for (int k = 0; k < (WIDE ? 4 : 3); k++) { raw[0][pair][k] = (jw * 0x9e3779b1u + (unsigned)(lane * 16 + k)) & (WIDE ? 0x03ff03ffu : ~0u); raw[1][pair][k] = raw[0][pair][k] ^ 0x00550055u; }
                continue;
            }
            shift[0][pair] = mc4q_issue<Pixel>(rec[0][pair], sx0[pair] - before, sy0[pair] - before, h_m[pair] + taps - 1, r, half, raw[0][pair]);
            if (any_bi) shift[1][pair] = mc4q_issue<Pixel>(rec[1][pair], sx1[pair] - before, sy1[pair] - before, h_m[pair] + taps - 1, r, half, raw[1][pair]);
        }
    }
    // ---- operand side: lane (n = lane & 15, g = lane >> 4)
    const int n = lane & 15, g = lane >> 4, maxv = (1 << bit_depth) - 1;
    const bool lowcol = n < 8, lowgrp = g < 2;
    {
        u32x2 b1[2][2];
        unsigned b2[2][2];
#pragma unroll
        for (int pair = 0; pair < 2; pair++) {
            // pass 1: column n belongs to block (n < 8 ? first : second) of the pair, its taps sit on that block's slot groups only
            const int blk1 = 64 * pair + (lowcol ? 0 : 32);
            const unsigned c1d1 = field(blk1, 1), c1d4 = field(blk1, 4);
            const int ph1 = ((c1d1 >> 16) & 0xff) == 0 ? 0 : 4, tl = (g & 1) * 16 + (n & 7);
            const bool mine = lowcol == lowgrp;
            const u32x2 t0 = tabs[0][ph1 + (int)(c1d4 & 0xff)][tl], t1 = tabs[0][ph1 + (int)((c1d4 >> 16) & 0xff)][tl];
            b1[0][pair] = mine ? t0 : u32x2{ 0u, 0u };
            b1[1][pair] = mine ? t1 : u32x2{ 0u, 0u };
            // pass 2: column y < 8 carries the vertical taps of a (pair 0) / d (pair 1), y >= 8 those of b / c; rows 4g .. 4g + 3 of the 15-row window
            const int blk2 = 64 * pair + ((lowcol == (pair == 0)) ? 0 : 32);
            const unsigned c2d1 = field(blk2, 1), c2d4 = field(blk2, 4);
            const int ph2 = ((c2d1 >> 16) & 0xff) == 0 ? 0 : 4, tl2 = g * 16 + (n & 7);
            b2[0][pair] = tabs[1][ph2 + (int)((c2d4 >> 8) & 0xff)][tl2].x;
            b2[1][pair] = tabs[1][ph2 + (int)(c2d4 >> 24)][tl2].x;
        }
        unsigned seen0[2] = { 0, 0 }, seen1[2] = { 0, 0 };
        int v0[4], v1[4] = { 0, 0, 0, 0 };
        if (TWIN == 1) {                                                                         // no arithmetic: the lane's own samples go out
            for (int k = 0; k < 4; k++) { v0[k] = (int)((raw[0][0][k & 1] ^ raw[0][1][k & 1] ^ shift[0][0]) & 0xff) << 6; v1[k] = any_bi ? (int)((raw[1][0][k & 1] ^ raw[1][1][k & 1]) & 0xff) << 6 : 0; }
        } else {
            mc4q_finish<Pixel>(raw[0], shift[0], b1[0], b2[0], bit_depth, lane, seen0, v0);
            if (any_bi) mc4q_finish<Pixel>(raw[1], shift[1], b1[1], b2[1], bit_depth, lane, seen1, v1);
        }
        // ---- result side: lane (y = lane & 15, g): four samples x = 4 (g & 1) .. + 3 of row y & 7 of block  a: y < 8, g < 2   d: y < 8, g >= 2   c: y >= 8, g < 2   b: y >= 8, g >= 2
        const int blk = lowcol ? (lowgrp ? 0 : 3) : (lowgrp ? 2 : 1);
        const unsigned m0 = field(32 * blk, 0), m1 = field(32 * blk, 1), m5 = field(32 * blk, 5), m6 = field(32 * blk, 6), m7 = field(32 * blk, 7);
        const int m_x = (int)(m0 & 0xffff), m_y = (int)(m0 >> 16), m_w = (int)(m1 & 0xff), m_h = (int)((m1 >> 8) & 0xff), m_plane = (int)((m1 >> 16) & 0xff);
        const int m_flags = (int)(m1 >> 24), m_denom = (int)((m5 >> 16) & 0xff);
        const int m_wx0 = (int)(short)(m6 & 0xffff), m_wx1 = (int)m6 >> 16, m_ox0 = (int)(short)(m7 & 0xffff), m_ox1 = (int)m7 >> 16;
        bool skip = 4 * q + blk >= njobs;
        if (WIDE) {
            // samples above the bit depth's range: mc3_redo_kernel computes the block (mc4_kernel).  `seen` lives on the operand side: rows of the
            // pair's first block in slot groups 0, 1, of its second block in groups 2, 3
            const unsigned wild_bits = 0x10001u * (unsigned)(0xffff & ~maxv);
            const bool s0 = ((seen0[0] | seen1[0]) & wild_bits) != 0, s1 = ((seen0[1] | seen1[1]) & wild_bits) != 0;
            const bool wa = __ballot(s0 && lowgrp) != 0, wb = __ballot(s0 && !lowgrp) != 0, wc = __ballot(s1 && lowgrp) != 0, wd = __ballot(s1 && !lowgrp) != 0;
            if (lane < 4) {
                const bool wl = lane == 0 ? wa : lane == 1 ? wb : lane == 2 ? wc : wd;
                if (wl && 4 * q + lane < njobs) atomicOr(&wild_mask[4 * q + lane], 1u);
            }
            skip = skip || (blk == 0 ? wa : blk == 1 ? wb : blk == 2 ? wc : wd);
        }
        const bool bi = (m_flags & OHEVC_MC_BI) != 0, weighted = (m_flags & OHEVC_MC_WEIGHTED) != 0;
        int w0, w1, off, sh, add;
        if (!weighted) {
            sh = (bi ? 15 : 14) - bit_depth; w0 = 1; w1 = bi ? 1 : 0; off = mc_round(sh, bit_depth); add = 0;
        } else if (!bi) {
            sh = m_denom + 14 - bit_depth; w0 = m_wx0; w1 = 0; off = mc_round(sh, bit_depth); add = m_ox0 * (1 << (bit_depth - 8));
        } else {
            const int log2wd = m_denom + 14 - bit_depth;
            sh = log2wd + 1; w0 = m_wx0; w1 = m_wx1; off = (m_ox0 * (1 << (bit_depth - 8)) + m_ox1 * (1 << (bit_depth - 8)) + 1) << log2wd; add = 0;
        }
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = __mul24(v1[k], w1) + (__mul24(v0[k], w0) + off);
            const int out = (t >> sh) + add;
            o[k] = (unsigned)(out < 0 ? 0 : out > maxv ? maxv : out);
        }
        const int sy = n & 7, sx = 4 * (g & 1);
        if (skip || sy >= m_h || sx >= m_w) return;
        unsigned pk[WIDE ? 2 : 1];
        if (WIDE) { pk[0] = o[0] | (o[1] << 16); pk[WIDE ? 1 : 0] = o[2] | (o[3] << 16); }
        else      pk[0] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        const unsigned char *const pbase = m_plane == 0 ? PLANE_PTR3(dst, 0) : m_plane == 1 ? PLANE_PTR3(dst, 1) : PLANE_PTR3(dst, 2);
        const int pstride = m_plane == 0 ? PLANE_STRIDE3(dst, 0) : m_plane == 1 ? PLANE_STRIDE3(dst, 1) : PLANE_STRIDE3(dst, 2);
        MC4_GLOBAL unsigned char *p = (MC4_GLOBAL unsigned char *)pbase + (__umul24((unsigned)(m_y + sy), (unsigned)pstride) + (unsigned)(m_x + sx) * (unsigned)sizeof(Pixel));
        if (m_w - sx >= 4) {
            __builtin_memcpy((void *)p, pk, sizeof(pk));
        } else {                                                      // widths 2 and 6 (chroma of 4- and 12-wide blocks)
            for (int k = 0; k < m_w - sx; k++)
                reinterpret_cast<MC4_GLOBAL Pixel *>(p)[k] = (Pixel)(WIDE ? pk[WIDE ? k >> 1 : 0] >> (16 * (k & 1)) : pk[0] >> (8 * k));
        }
    }
}
