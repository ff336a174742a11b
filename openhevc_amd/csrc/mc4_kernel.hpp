// mc4_kernel.hpp -- motion compensation on the matrix cores (included by mc_kernels.hip inside namespace ohevc).
//
// mc3_kernel's 16x16 tile costs ~4 us of load -> LDS -> barrier -> filter -> LDS -> barrier -> filter -> store and ~560 VALU
// instructions per lane; measured out of HBM it sustains 5-18 % of the roofline.  The separable 8- / 4-tap filter is
// a product with two banded (Toeplitz) matrices,
//     H[r][x] = sum_c W[r][c] F1[c][x]          F1[c][x] = fh[c - x]        W = the (w + T - 1) x (h + T - 1) reference window
//     V[x][y] = sum_r H[r][x] F2[r][y]          F2[r][y] = fv[r - y]
// and v_mfma_i32_16x16x32_i8 multiplies int8 exactly, so a tile needs no LDS, no barrier and no cross-lane traffic at all:
//   * operand slots.  For 16x16 MFMAs lane l supplies row / column l & 15 and 8 K slots of slot group l >> 4; which K index a slot
//     means is ours to choose as long as A and B agree.  Pass 1: group g takes window columns 8g .. 8g + 7 (32 >= 23), so the A operand of
//     a lane is 8 consecutive samples of ONE window row: one (unaligned) 8- / 16-byte global load per lane - issued with the lanes of a
//     quad on the same row (coalesced) and handed to the operand's lane through ds_bpermute_b32 -, and two MFMAs cover window rows 0..15 and 16..31.  Their results (lane = x, registers = rows 4g + r / 16 + 4g + r) ARE the A
//     operand of pass 2 once its constant operand lists the rows in that order; pass 2 comes out with lane = y and registers = 4
//     consecutive x: one 4- / 8-byte store per lane.
//   * int8 planes.  8-bit samples go in as p - 128; intermediates and >8-bit samples as x = 256 hi + (lo + 128).  Every filter's taps sum
//     to 64, so the offsets are constants (128 * 64 per plane) that ride in the `(acc << 8) + K` between the two MFMAs of a pair.
//   * the constant operands (per phase and lane, 8 bytes) come from two 6 KB tables.
// Exactness: every partial sum fits int32 and the planes reassemble the reference's int values (hevcdsp_template.c:731-983,
// 1185-1247); samples above the bit depth's range (16-bit planes, mc3_kernel's comment) are detected on the loaded words and left to
// mc3_redo_kernel exactly as before.
typedef int mc4_v4i __attribute__((ext_vector_type(4)));

struct Mc4Tabs { unsigned b1[12][64][2]; unsigned b2[12][64][2]; };    // phase: luma 0..3, chroma 4 + (0..7)
constexpr Mc4Tabs make_mc4_tabs()
{
    constexpr int L[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    constexpr int Cc[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 },
                               { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
    Mc4Tabs t{};
    for (int ph = 0; ph < 12; ph++) {
        const int taps = ph < 4 ? 8 : 4;
        for (int l = 0; l < 64; l++) {
            const int n = l & 15, g = l >> 4;
            for (int d = 0; d < 2; d++) {
                unsigned w1 = 0, w2 = 0;
                for (int e = 0; e < 4; e++) {
                    const int j = 4 * d + e;
                    const int t1 = 8 * g + j - n;                                         // pass 1: slot = window column 8g + j
                    const int row = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);             // pass 2: slot = window row (D layout of pass 1)
                    const int t2 = row - n;
                    const int v1 = t1 >= 0 && t1 < taps ? (ph < 4 ? L[ph][t1] : Cc[ph - 4][t1]) : 0;
                    const int v2 = t2 >= 0 && t2 < taps ? (ph < 4 ? L[ph][t2] : Cc[ph - 4][t2]) : 0;
                    w1 |= (unsigned)(v1 & 0xff) << (8 * e);
                    w2 |= (unsigned)(v2 & 0xff) << (8 * e);
                }
                t.b1[ph][l][d] = w1; t.b2[ph][l][d] = w2;
            }
        }
    }
    return t;
}
__device__ const Mc4Tabs kMc4 = make_mc4_tabs();

__device__ __forceinline__ unsigned mc4_pk16(int a, int b) { return __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x05040100u); }                 // low halves of a, b
__device__ __forceinline__ unsigned mc4_lo(unsigned p0, unsigned p1) { return __builtin_amdgcn_perm(p1, p0, 0x06040200u) ^ 0x80808080u; }           // low bytes of 4 int16, - 128
__device__ __forceinline__ unsigned mc4_hi(unsigned p0, unsigned p1) { return __builtin_amdgcn_perm(p1, p0, 0x07050301u); }                         // high bytes of 4 int16

// ---- one reference of one tile, in two halves so that a wavefront can have the loads of several tiles in flight
template <typename Pixel> struct Mc4Raw { unsigned w[2][sizeof(Pixel) == 2 ? 4 : 2]; };      // [row block][8 samples of one window row]

#ifdef OHEVC_HIPEMU
#define MC4_GLOBAL
#define MC4_CONST
#else
#define MC4_GLOBAL __attribute__((address_space(1)))             // the planes live in global memory: global_load / global_store, not flat_*
#define MC4_CONST __attribute__((address_space(4)))              // read-only for the kernel's lifetime + wave-uniform address: s_load
#endif
typedef const MC4_GLOBAL unsigned char *mc4_gptr;

// A window that crosses the left / right picture edge (emulated_edge_mc, videodsp_template.c:26-100): 8 clamped sample loads per lane.
// Out of line on purpose - inlined, its address arithmetic is hoisted in front of the branch and every tile pays for it.
template <typename Pixel>
__device__ __attribute__((noinline)) u32x4 mc4_gather_edge(mc4_gptr row, int col0, int xmax)
{
    constexpr bool WIDE = sizeof(Pixel) == 2;
    unsigned w[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int j = 0; j < 8; j++) {
        int x = col0 + j;
        x = x < 0 ? 0 : x > xmax ? xmax : x;
        const unsigned s = *reinterpret_cast<const MC4_GLOBAL Pixel *>(row + (unsigned)x * (unsigned)sizeof(Pixel));
        if (WIDE) w[j >> 1] |= s << (16 * (j & 1));
        else      w[j >> 2] |= s << (8 * (j & 3));
    }
    return u32x4{ w[0], w[1], w[2], w[3] };                       // by value: a pointer argument would push the callers' registers to scratch
}

// (wx0, wy0): picture position of window sample (0, 0); wh: window rows that exist (rows beyond repeat the last one: their taps are 0).
template <typename Pixel>
__device__ __forceinline__ void mc4_issue(const ohevc_plane &ref, int wx0, int wy0, int wh, int lane, Mc4Raw<Pixel> &raw)
{
    // memory-side lane map: 4 consecutive lanes take 32 consecutive samples of one window row (one 32- / 64-byte piece per quad of
    // lanes; with the operand-side map - lane & 15 = row - every lane of a load would touch its own cache line)
    const int r = lane >> 2, g = lane & 3;
    mc4_gptr base = (mc4_gptr)ref.data;
    const int xmax = ref.width - 1, ymax = ref.height - 1;
    const int col0 = wx0 + 8 * g;
    const bool fast = wx0 >= 0 && wx0 + 31 <= xmax;               // wave-uniform: every lane's 8 samples lie inside its row
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        int wr = 16 * blk + r;
        wr = wr < wh ? wr : wh - 1;
        int y = wy0 + wr;
        y = y < 0 ? 0 : y > ymax ? ymax : y;
        const unsigned rowoff = __umul24((unsigned)y, (unsigned)ref.stride);                    // rows < 2^16, strides < 2^24, planes far below 4 GiB
        if (fast) {
            __builtin_memcpy(raw.w[blk], (const void *)(base + (rowoff + (unsigned)col0 * (unsigned)sizeof(Pixel))), sizeof(raw.w[blk]));     // one global_load_dwordx2 / x4, any alignment
        } else {
            const u32x4 e = mc4_gather_edge<Pixel>(base + rowoff, col0, xmax);
            raw.w[blk][0] = e.x; raw.w[blk][1] = e.y;
            if (sizeof(Pixel) == 2) { raw.w[blk][sizeof(Pixel) == 2 ? 2 : 0] = e.z; raw.w[blk][sizeof(Pixel) == 2 ? 3 : 1] = e.w; }
        }
    }
}

// the 14-bit intermediate of the tile: v[r] = sample (x = 4 (lane >> 4) + r, y = lane & 15)
__device__ __forceinline__ long mc4_op(unsigned lo, unsigned hi) { return (long)(((unsigned long)hi << 32) | lo); }
template <typename Pixel>
__device__ __forceinline__ void mc4_finish(const Mc4Raw<Pixel> &raw, u32x2 b1, u32x2 b2, int bit_depth, int lane, unsigned &seen, int *v)
{
    constexpr bool WIDE = sizeof(Pixel) == 2;
    const mc4_v4i zero = { 0, 0, 0, 0 };
    const long bop1 = mc4_op(b1.x, b1.y), bop2 = mc4_op(b2.x, b2.y);
    int hv[8];
    const int src = (lane & 15) * 4 + (lane >> 4);               // operand-side lane (row lane & 15, slot group lane >> 4) <- memory-side lane
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        unsigned w[WIDE ? 4 : 2];
#pragma unroll
        for (int k = 0; k < (WIDE ? 4 : 2); k++) w[k] = (unsigned)__shfl((int)raw.w[blk][k], src);       // ds_bpermute_b32
        mc4_v4i d;
        if (WIDE) {
            seen |= w[0] | w[1] | w[WIDE ? 2 : 0] | w[WIDE ? 3 : 1];
            d = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_hi(w[0], w[1]), mc4_hi(w[WIDE ? 2 : 0], w[WIDE ? 3 : 1])), bop1, zero, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = (d[k] << 8) + 8192;                               // + 128 * sum(taps): the low plane is lo - 128
            d = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_lo(w[0], w[1]), mc4_lo(w[WIDE ? 2 : 0], w[WIDE ? 3 : 1])), bop1, d, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) hv[4 * blk + k] = d[k] >> (bit_depth - 8);
        } else {
            d = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(w[0] ^ 0x80808080u, w[1] ^ 0x80808080u), bop1, zero, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) hv[4 * blk + k] = d[k];                                  // = h - 8192: still an int16; undone by K2 below
        }
    }
    const unsigned p0 = mc4_pk16(hv[0], hv[1]), p1 = mc4_pk16(hv[2], hv[3]), p2 = mc4_pk16(hv[4], hv[5]), p3 = mc4_pk16(hv[6], hv[7]);
    const int K2 = 8192 + (WIDE ? 0 : 8192 * 64);
    mc4_v4i e = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_hi(p0, p1), mc4_hi(p2, p3)), bop2, zero, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = (e[k] << 8) + K2;
    e = __builtin_amdgcn_mfma_i32_16x16x32_i8(mc4_op(mc4_lo(p0, p1), mc4_lo(p2, p3)), bop2, e, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = e[k] >> 6;
}

// Work unit = one 16x16 tile of one job; a wavefront takes UNITS of them and keeps the loads of all of them in flight before it
// computes the first (a lone tile is ~3 dependent memory round trips - job record, reference-plane record, samples - for 0.3 us of
// arithmetic).  256 threads = 4 wavefronts share one LDS copy of the constant-operand tables.
//   MULTI == false   every job is at most 16x16 (what the ctx layer records): wavefront q of the grid takes jobs 4q .. 4q + 3
//   MULTI == true    blockIdx.x * 4 + wave = the job, blockIdx.y * 4 + i = its tile (jobs of up to 64x64: up to 16 tiles)
constexpr int MC4_UNITS = 4;
template <typename Pixel, bool MULTI>
__global__ __launch_bounds__(256) void mc4_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs, const ohevc_mc_job *__restrict__ jobs, int njobs,
                                                  int bit_depth, unsigned *__restrict__ wild_mask)
{
    __shared__ u32x2 tabs[2][12][64];                                                            // 12 KB: kMc4
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(&kMc4);
        u32x4 *dl = reinterpret_cast<u32x4 *>(&tabs[0][0][0]);
        static_assert(sizeof(kMc4) == sizeof(tabs) && sizeof(tabs) == 3 * 256 * 16, "table copy");
        const u32x4 t0 = src[tid], t1 = src[256 + tid], t2 = src[512 + tid];
        dl[tid] = t0; dl[256 + tid] = t1; dl[512 + tid] = t2;
    }
    struct Unit { bool valid, bi; ohevc_mc_job jb; int tx, ty, tw, th, slot; ohevc_plane r0, r1; };
    Unit u[MC4_UNITS];
    // Workgroups go to the 8 XCDs round-robin and every XCD has its own L2: give each XCD a CONTIGUOUS eighth of the job list (jobs
    // arrive in picture order), so that the overlapping reference windows of neighbouring tiles meet in one L2 instead of being
    // fetched by up to 8.  (The launcher rounds the grid up to a multiple of 8; the surplus workgroups find no job.)
    const int per_xcd = gridDim.x >> 3;
    const int q = (((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3)) * 4 + wave;
    // Three rounds, each issued for all units before the first result is needed: job records, reference-plane records (both scalar
    // loads), samples.  Unit by unit the three dependent round trips would add up four times.
    typedef const MC4_CONST u32x4 *cptr;
    u32x4 jw[MC4_UNITS][2];
#pragma unroll
    for (int i = 0; i < MC4_UNITS; i++) {
        const int j = MULTI ? q : q * MC4_UNITS + i;
        u[i].valid = j < njobs;
        cptr jp = (cptr)(jobs + (u[i].valid ? j : njobs - 1));
        jw[i][0] = jp[0]; jw[i][1] = jp[1];
    }
#ifndef OHEVC_HIPEMU
    // the words of all four records that the reference-plane addresses are made of (plane | flags, ref0 | ref1), named as operands of ONE
    // empty asm: the four loads above leave together and are waited for once.  Left alone, the compiler interleaves the records' loads with
    // branches it makes of the bi / valid selects below - seven dependent scalar round trips in front of the first sample load.
    asm volatile("" : "+s"(jw[0][0].y), "+s"(jw[0][1].y), "+s"(jw[1][0].y), "+s"(jw[1][1].y), "+s"(jw[2][0].y), "+s"(jw[2][1].y), "+s"(jw[3][0].y), "+s"(jw[3][1].y));
    static_assert(MC4_UNITS == 4, "the operand list above");
#endif
    const int maxv = (1 << bit_depth) - 1;
#pragma unroll
    for (int i = 0; i < MC4_UNITS; i++) {
        const unsigned words[8] = { jw[i][0].x, jw[i][0].y, jw[i][0].z, jw[i][0].w, jw[i][1].x, jw[i][1].y, jw[i][1].z, jw[i][1].w };
        __builtin_memcpy(&u[i].jb, words, sizeof(ohevc_mc_job));
        const ohevc_mc_job &jb = u[i].jb;
        const int ntx = (jb.w + 15) >> 4, nty = (jb.h + 15) >> 4;
        const int t = MULTI ? (int)blockIdx.y * MC4_UNITS + i : 0;
        const int tyi = ntx == 1 ? t : ntx == 2 ? t >> 1 : ntx == 4 ? t >> 2 : t / 3, txi = t - tyi * ntx;
        u[i].valid = u[i].valid && tyi < nty;
        u[i].tx = txi * 16; u[i].ty = tyi * 16;
        u[i].tw = jb.w - u[i].tx < 16 ? jb.w - u[i].tx : 16; u[i].th = jb.h - u[i].ty < 16 ? jb.h - u[i].ty : 16;
        u[i].bi = jb.flags & OHEVC_MC_BI;
        u[i].slot = tyi * ntx + txi;
        static_assert(sizeof(ohevc_plane) == 24, "plane record");
        typedef const MC4_CONST unsigned long *lptr;
        const int bimask = -(int)((jb.flags & OHEVC_MC_BI) != 0);                   // (a select here comes back as a branch around slot1)
        int slot0 = 3 * jb.ref0 + jb.plane, slot1 = 3 * (jb.ref0 ^ ((jb.ref0 ^ jb.ref1) & bimask)) + jb.plane;
#ifndef OHEVC_HIPEMU
        asm volatile("" : "+s"(slot0), "+s"(slot1));             // selected, not branched around (see above)
#endif
        lptr p0 = (lptr)(refs + slot0), p1 = (lptr)(refs + slot1);
        const unsigned long a0[3] = { p0[0], p0[1], p0[2] }, a1[3] = { p1[0], p1[1], p1[2] };
        __builtin_memcpy(&u[i].r0, a0, 24); __builtin_memcpy(&u[i].r1, a1, 24);
    }
    Mc4Raw<Pixel> raw[MC4_UNITS][2];
#pragma unroll
    for (int i = 0; i < MC4_UNITS; i++) {
        const ohevc_mc_job &jb = u[i].jb;
        if (u[i].valid) {
            const int before = jb.plane == 0 ? 3 : 1, taps = jb.plane == 0 ? 8 : 4;
            mc4_issue<Pixel>(u[i].r0, jb.sx0 + u[i].tx - before, jb.sy0 + u[i].ty - before, u[i].th + taps - 1, lane, raw[i][0]);
            if (u[i].bi) mc4_issue<Pixel>(u[i].r1, jb.sx1 + u[i].tx - before, jb.sy1 + u[i].ty - before, u[i].th + taps - 1, lane, raw[i][1]);
        }
    }
    __syncthreads();                                                                             // the tables are in LDS
    const unsigned wild_bits = 0x10001u * (unsigned)(0xffff & ~maxv);
#pragma unroll
    for (int i = 0; i < MC4_UNITS; i++) {
        if (!u[i].valid) continue;
        const ohevc_mc_job &jb = u[i].jb;
        const bool bi = u[i].bi, weighted = jb.flags & OHEVC_MC_WEIGHTED;
        const int ph = jb.plane == 0 ? 0 : 4;
        unsigned seen = 0;
        int v0[4], v1[4] = { 0, 0, 0, 0 };
        mc4_finish<Pixel>(raw[i][0], tabs[0][ph + jb.mx0][lane], tabs[1][ph + jb.my0][lane], bit_depth, lane, seen, v0);
        if (bi) mc4_finish<Pixel>(raw[i][1], tabs[0][ph + jb.mx1][lane], tabs[1][ph + jb.my1][lane], bit_depth, lane, seen, v1);
        const int job = MULTI ? q : q * MC4_UNITS + i;
        if (sizeof(Pixel) == 2 && __ballot((seen & wild_bits) != 0) != 0) {                      // mc3_redo_kernel computes this tile
            if (lane == 0) atomicOr(&wild_mask[job], 1u << u[i].slot);
            continue;
        }
        // the four cases of the reference (put_hevc_*_uni / _bi / _uni_w / _bi_w, hevcdsp_template.c:640-722, 796-983) as ONE expression
        // ((v0 * w0 + v1 * w1 + off) >> sh) + add with wave-uniform parameters; products of a 16-bit intermediate and a 9-bit weight
        int w0, w1, off, sh, add;
        if (!weighted) {
            sh = (bi ? 15 : 14) - bit_depth; w0 = 1; w1 = bi ? 1 : 0; off = mc_round(sh, bit_depth); add = 0;
        } else if (!bi) {
            sh = jb.denom + 14 - bit_depth; w0 = jb.wx0; w1 = 0; off = mc_round(sh, bit_depth); add = jb.ox0 * (1 << (bit_depth - 8));
        } else {
            const int log2wd = jb.denom + 14 - bit_depth;
            sh = log2wd + 1; w0 = jb.wx0; w1 = jb.wx1; off = (jb.ox0 * (1 << (bit_depth - 8)) + jb.ox1 * (1 << (bit_depth - 8)) + 1) << log2wd; add = 0;
        }
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = __mul24(v1[k], w1) + (__mul24(v0[k], w0) + off);
            const int out = (t >> sh) + add;
            o[k] = (unsigned)(out < 0 ? 0 : out > maxv ? maxv : out);
        }
        // back to the memory-side lane map: lane (row lane >> 2, 4 samples from x = 4 (lane & 3)) - a quad of lanes writes one row
        unsigned pk[sizeof(Pixel) == 2 ? 2 : 1];
        if (sizeof(Pixel) == 2) { pk[0] = o[0] | (o[1] << 16); pk[sizeof(Pixel) == 2 ? 1 : 0] = o[2] | (o[3] << 16); }
        else                    pk[0] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        const int from = (lane & 3) * 16 + (lane >> 2);
#pragma unroll
        for (int k = 0; k < (sizeof(Pixel) == 2 ? 2 : 1); k++) pk[k] = (unsigned)__shfl((int)pk[k], from);
        const int sy = lane >> 2, sx = 4 * (lane & 3);
        if (sy >= u[i].th || sx >= u[i].tw) continue;
        MC4_GLOBAL unsigned char *p = (MC4_GLOBAL unsigned char *)PLANE_PTR3(dst, jb.plane) + (__umul24((unsigned)(jb.y + u[i].ty + sy), (unsigned)PLANE_STRIDE3(dst, jb.plane)) + (unsigned)(jb.x + u[i].tx + sx) * (unsigned)sizeof(Pixel));
        if (u[i].tw - sx >= 4) {
            __builtin_memcpy((void *)p, pk, sizeof(pk));          // one 4- / 8-byte store
        } else {                                                  // widths 2 and 6 (chroma of 4- and 12-wide blocks)
            for (int k = 0; k < u[i].tw - sx; k++)
                reinterpret_cast<MC4_GLOBAL Pixel *>(p)[k] = (Pixel)(sizeof(Pixel) == 2 ? pk[sizeof(Pixel) == 2 ? k >> 1 : 0] >> (16 * (k & 1)) : pk[0] >> (8 * k));
        }
    }
}
