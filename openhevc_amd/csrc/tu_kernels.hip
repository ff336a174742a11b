// tu_kernels.hip -- residual (TU) family for gfx950: inverse transform + add into the picture plane.
//
// Replaces, in batched form, the reference's per-block sequence (hevc_cabac.c:1868-1949)
//     idct[log2-2] / idct_dc / idct_4x4_luma / transform_skip [+ transform_rdpcm]   (hevcdsp_template.c:114-326)
//     transform_add[log2-2](dst, coeffs, stride)                                      (hevcdsp_template.c:45-111)
//
// Main kernel (8x8 .. 32x32 inverse DCT + add), one 64-lane wavefront per 64/N blocks:
//   * lane = (block g, index i).  Pass 1: lane owns COLUMN i; pass 2: lane owns ROW i; the epilogue adds
//     row i of the residual to row i of the prediction and writes it back with 8/16-byte accesses.
//   * coefficients go HBM -> LDS with 16-byte loads, are re-read as int16 pairs, and every multiply-add runs
//     on v_dot2c_i32_i16 (2 int16 MACs per VALU op, exact 32-bit accumulate): the partial butterfly's odd
//     half of each level is a dot product over same-parity inputs, so inputs are kept as packed pairs
//     (j, j') of the same butterfly level ("pair order", see ord_lo/ord_hi).
//   * the column->row transpose goes through wave-private LDS: pass 1 scatters int16 results to
//     [row][slot(i)], pass 2 gathers its row with ds_read_b128 already in pair order.
//   * both clip_int16 stages are v_cvt_pk_i16_i32 (saturating pack); rounding constants ride in the DC lane of
//     the butterfly; the final clip_pixel is a saturating packed add + packed max/min.
// No floating point.  The shipped kernel uses no MFMA (int16 x int8 butterflies; it runs at the memory system's rate for this
// read/write mix); tu_idct32_mfma_kernel below is the matrix-core form of the 32x32 case, bit-exact, same speed.
#include <atomic>
#include "common.hpp"
#include "intra_body.hpp"
#include "tu_generic.hpp"

namespace ohevc {

// ------------------------------------------------------------------ the HEVC core transform matrix
// 64*sqrt(2)*cos(m*pi/64) rounded as in the standard; entry (r, c) of the 32-point matrix is
// +-mag[(2c+1)*r mod 128 folded into one quadrant]  (same numbers as libavcodec/hevcdsp.c:879-944).
__host__ __device__ constexpr int cos_mag(int m)
{
    constexpr int t[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                            61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
    return t[m];
}
__host__ __device__ constexpr int dct32(int r, int c)
{
    int m = (r * (2 * c + 1)) & 127;
    return m <= 32 ? cos_mag(m) : m <= 64 ? -cos_mag(64 - m) : m <= 96 ? -cos_mag(m - 64) : cos_mag(128 - m);
}

// "pair order" of an N-point input vector: N/2 dwords; dword m holds inputs (ord_lo, ord_hi).
//   first N/4 dwords: the odd inputs (1,3),(5,7),...   (odd half of the outermost butterfly level)
//   then the same rule applied to the even inputs (halved), down to the 4-point base {(1,3),(0,2)}.
__host__ __device__ constexpr int ord_lo(int n, int m)
{
    return n == 4 ? (m == 0 ? 1 : 0) : m < n / 4 ? 4 * m + 1 : 2 * ord_lo(n / 2, m - n / 4);
}
__host__ __device__ constexpr int ord_hi(int n, int m)
{
    return n == 4 ? (m == 0 ? 3 : 2) : m < n / 4 ? 4 * m + 3 : 2 * ord_hi(n / 2, m - n / 4);
}
// inverse map: int16 slot (2*dword + half) of input index i
__host__ __device__ constexpr int slot_of(int n, int i)
{
    if ((i & (n / 2 - 1)) == 0) return n - 2 + (i ? 1 : 0);
    int t = 0;
    while (!((i >> t) & 1)) t++;
    return n - (n >> t) + (i >> (t + 1));
}
template <int N> constexpr bool pair_order_ok()
{
    for (int m = 0; m < N / 2; m++)
        if (slot_of(N, ord_lo(N, m)) != 2 * m || slot_of(N, ord_hi(N, m)) != 2 * m + 1) return false;
    return true;
}
static_assert(pair_order_ok<4>() && pair_order_ok<8>() && pair_order_ok<16>() && pair_order_ok<32>(), "pair order");

// compile-time tables (indexing them with an unrolled loop counter folds to an immediate)
template <int N> struct PairTab {
    int lo[N / 2], hi[N / 2];
    constexpr PairTab() : lo{}, hi{}
    {
        for (int m = 0; m < N / 2; m++) { lo[m] = ord_lo(N, m); hi[m] = ord_hi(N, m); }
    }
};

__device__ __forceinline__ int slot_of_rt(int n, int i)
{
    if ((i & (n / 2 - 1)) == 0) return n - 2 + (i ? 1 : 0);
    int t = __builtin_ctz(i);
    return n - (n >> t) + (i >> (t + 1));
}

// M-point inverse transform of inputs in pair order -> out[0..M-1] (natural order), exact int32.
// out[k] = sum_j T_M[j][k] * x[j] + init, with T_M[j][k] = dct32(j * 32/M, k); evaluated as the even/odd
// partial butterfly of hevcdsp_template.c:210-262 (any evaluation order is bit-identical: no overflow).
template <int M, bool ASM = false> struct Idct1D {
    static __device__ __forceinline__ void run(const unsigned *p, int *out, int init)
    {
        int e[M / 2];
        Idct1D<M / 2, ASM>::run(p + M / 4, e, init);
#pragma unroll
        for (int i = 0; i < M / 2; i++) {
            int o;
            if constexpr (ASM) {
                o = dot2_i16_first(p[0], pack16(dct32(1 * (32 / M), i), dct32(3 * (32 / M), i)));
            } else {
                o = dot2_i16(p[0], pack16(dct32(1 * (32 / M), i), dct32(3 * (32 / M), i)), 0);
            }
#pragma unroll
            for (int m = 1; m < M / 4; m++)
                o = dot2_i16(p[m], pack16(dct32((4 * m + 1) * (32 / M), i), dct32((4 * m + 3) * (32 / M), i)), o);
            out[i]         = e[i] + o;
            out[M - 1 - i] = e[i] - o;
        }
    }
};
template <bool ASM> struct Idct1D<2, ASM> {     // inputs (x0, x_{M/2}) of the enclosing 4-point level: the "+-64" lane
    static __device__ __forceinline__ void run(const unsigned *p, int *out, int init)
    {
        out[0] = dot2_i16(p[0], pack16(64, 64), init);
        out[1] = dot2_i16(p[0], pack16(64, -64), init);
    }
};

template <int LOG2N> struct TuLayout {
    static constexpr int N   = 1 << LOG2N;
    static constexpr int BPW = 64 / N;                                  // blocks per wavefront
    // wave-private LDS tile per block: N rows of N int16, row stride padded so that the 16 rows a
    // ds_read_b128 lane group touches fall into 16 different 16-byte bank slots; block bases skewed so
    // that blocks sharing a 32-lane half do not collide on the int16 column accesses.
    static constexpr int RS  = LOG2N == 5 ? 80 : LOG2N == 4 ? 48 : 16;  // bytes
    static constexpr int BLK = LOG2N == 5 ? 32 * 80 : LOG2N == 4 ? 16 * 48 + 32 : 8 * 16 + 16;
    static constexpr int WAVE_BYTES = BPW * BLK;
};

template <int N, typename Pixel>
__device__ __forceinline__ void load_row(const unsigned char *row, unsigned *px, bool valid)
{
    constexpr int ROWDW = N * (int)sizeof(Pixel) / 4;
    constexpr int VEC   = ROWDW >= 4 ? 4 : ROWDW;                        // dwords per access (4, 2 or 1)
    if (valid) {
#pragma unroll
        for (int v = 0; v < ROWDW / VEC; v++) {
            if constexpr (VEC == 4) {
                u32x4 t = *reinterpret_cast<const u32x4 *>(row + 16 * v);
                px[4 * v] = t.x; px[4 * v + 1] = t.y; px[4 * v + 2] = t.z; px[4 * v + 3] = t.w;
            } else if constexpr (VEC == 2) {
                u32x2 t = *reinterpret_cast<const u32x2 *>(row + 8 * v);
                px[2 * v] = t.x; px[2 * v + 1] = t.y;
            } else {
                px[v] = *reinterpret_cast<const unsigned *>(row + 4 * v);
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < ROWDW; d++) px[d] = 0;
    }
}

template <int N, typename Pixel>
__device__ __forceinline__ void finish_row(unsigned char *row, unsigned *px, const int *res, int bit_depth, bool valid)
{
    constexpr int ROWDW = N * (int)sizeof(Pixel) / 4;
    constexpr int VEC   = ROWDW >= 4 ? 4 : ROWDW;
    const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
#pragma unroll
    for (int d = 0; d < ROWDW; d++) {
        if constexpr (sizeof(Pixel) == 1) {
            unsigned u01 = __builtin_amdgcn_perm(0u, px[d], 0x0c010c00u);
            unsigned u23 = __builtin_amdgcn_perm(0u, px[d], 0x0c030c02u);
            unsigned s01 = add_clamp_px2(sat_pack_i16(res[4 * d], res[4 * d + 1]), u01, max2);
            unsigned s23 = add_clamp_px2(sat_pack_i16(res[4 * d + 2], res[4 * d + 3]), u23, max2);
            px[d] = __builtin_amdgcn_perm(s23, s01, 0x06040200u);
        } else {
            px[d] = add_clamp_upx2(sat_pack_i16(res[2 * d], res[2 * d + 1]), px[d], max2);
        }
    }
    if (valid) {
#pragma unroll
        for (int v = 0; v < ROWDW / VEC; v++) {
            if constexpr (VEC == 4) {
                u32x4 t = { px[4 * v], px[4 * v + 1], px[4 * v + 2], px[4 * v + 3] };
                *reinterpret_cast<u32x4 *>(row + 16 * v) = t;
            } else if constexpr (VEC == 2) {
                u32x2 t = { px[2 * v], px[2 * v + 1] };
                *reinterpret_cast<u32x2 *>(row + 8 * v) = t;
            } else {
                *reinterpret_cast<unsigned *>(row + 4 * v) = px[v];
            }
        }
    }
}

// The same with the prediction still as N ints (what the intra predictors leave) and the residual ALREADY as N / 2 packed pairs of
// clip_int16'ed values (what intra_chain_residual_kernel left in the arena): the pairs go straight into the packed add - no unpacking of the
// residual into ints, no packing of the prediction into pixels and back.  A lane of a 32x32 block does this for 32 samples, and a level of the
// intra chain lasts as long as its largest block.
template <int N, typename Pixel>
__device__ __forceinline__ void finish_row_pairs(unsigned char *row, const int *pred, const unsigned *res2, int bit_depth, bool valid)
{
    constexpr int ROWDW = N * (int)sizeof(Pixel) / 4;
    constexpr int VEC   = ROWDW >= 4 ? 4 : ROWDW;
    const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
    unsigned px[ROWDW];
#pragma unroll
    for (int d = 0; d < ROWDW; d++) {
        if constexpr (sizeof(Pixel) == 1) {
            const unsigned u01 = (unsigned)pred[4 * d] | ((unsigned)pred[4 * d + 1] << 16), u23 = (unsigned)pred[4 * d + 2] | ((unsigned)pred[4 * d + 3] << 16);
            const unsigned s01 = add_clamp_px2(res2[2 * d], u01, max2), s23 = add_clamp_px2(res2[2 * d + 1], u23, max2);
            px[d] = __builtin_amdgcn_perm(s23, s01, 0x06040200u);
        } else {
            px[d] = add_clamp_upx2(res2[d], (unsigned)pred[2 * d] | ((unsigned)pred[2 * d + 1] << 16), max2);
        }
    }
    if (valid) {
#pragma unroll
        for (int v = 0; v < ROWDW / VEC; v++) {
            if constexpr (VEC == 4) {
                u32x4 t = { px[4 * v], px[4 * v + 1], px[4 * v + 2], px[4 * v + 3] };
                *reinterpret_cast<u32x4 *>(row + 16 * v) = t;
            } else if constexpr (VEC == 2) {
                u32x2 t = { px[2 * v], px[2 * v + 1] };
                *reinterpret_cast<u32x2 *>(row + 8 * v) = t;
            } else {
                *reinterpret_cast<unsigned *>(row + 4 * v) = px[v];
            }
        }
    }
}

// res[0..N-1] (already >> shift, any int32) + prediction row -> clipped pixels, in place in HBM.
// clip_pixel(pred + clip_int16(r)) is what transform_add computes; sat_pack_i16 is the clip_int16.
template <int N, typename Pixel>
__device__ __forceinline__ void add_row_store(unsigned char *row, const int *res, int bit_depth, bool valid)
{
    unsigned px[N * (int)sizeof(Pixel) / 4];
    load_row<N, Pixel>(row, px, valid);
    finish_row<N, Pixel>(row, px, res, bit_depth, valid);
}

// ------------------------------------------------------------------ 8x8 / 16x16 / 32x32 IDCT + add
// VARIANT bits (A/B switches, see ohevc_debug.h): 1 = fetch the prediction row before the transform instead of
// after it; 2 = read coefficients straight from HBM as int16 columns (no LDS staging pass).
template <int LOG2N, typename Pixel, int VARIANT>
__device__ __forceinline__ void tu_idct_add_body(unsigned char *lds, int wg, const PlaneSet planes, const ohevc_tu_job *__restrict__ jobs,
                                                 int njobs, const int16_t *__restrict__ coeffs, int bit_depth)
{
    using L = TuLayout<LOG2N>;
    constexpr int N = L::N, RS = L::RS;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / N, i = lane % N;
    const int job0 = (wg * 4 + wave) * L::BPW;
    constexpr bool WG_EPILOGUE = (VARIANT & 512) && LOG2N >= 4;      // the epilogue below synchronises the whole workgroup
    if constexpr (!WG_EPILOGUE) { if (job0 >= njobs) return; }         // wave-uniform; no workgroup barrier is used in that form
    const bool valid = job0 + g < njobs;
    const u32x4 jraw = reinterpret_cast<const u32x4 *>(jobs)[valid ? job0 + g : njobs - 1];
    const int jx = jraw.x & 0xffff, jy = jraw.x >> 16, jplane = jraw.y & 0xff;
    const unsigned coeff_off = jraw.z;
    unsigned char *blk = lds + wave * L::WAVE_BYTES + g * L::BLK;

    unsigned char *row = PLANE_PTR3(planes, jplane) + (size_t)(jy + i) * PLANE_STRIDE3(planes, jplane) + (size_t)jx * sizeof(Pixel);
    unsigned px[N * (int)sizeof(Pixel) / 4];
    if constexpr (VARIANT & 1) load_row<N, Pixel>(row, px, valid);

    unsigned p[N / 2];
    constexpr PairTab<N> pt{};
    if constexpr (VARIANT & 2) {
        // ---- A'. coefficients straight from HBM: lane i reads column i, one int16 per row
        const unsigned short *col = reinterpret_cast<const unsigned short *>(coeffs + coeff_off) + i;
#pragma unroll
        for (int m = 0; m < N / 2; m++)
            p[m] = (unsigned)col[pt.lo[m] * N] | ((unsigned)col[pt.hi[m] * N] << 16);
    } else {
        // ---- A. coefficients HBM -> LDS (row-major, padded rows), 16 bytes per lane per access
        const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + coeff_off);
#pragma unroll
        for (int q = 0; q < N / 8; q++) {
            const int c = q * N + i;                        // 16-byte chunk index inside the block
            u32x4 v;
            if constexpr (VARIANT & 1024) v = __builtin_nontemporal_load(src + c);      // read exactly once: do not keep it in L2 / MALL
            else v = src[c];
            *reinterpret_cast<u32x4 *>(blk + (c / (N / 8)) * RS + (c % (N / 8)) * 16) = v;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- B. pass 1: column i, inputs gathered as same-level pairs
        const unsigned short *col = reinterpret_cast<const unsigned short *>(blk) + i;
#pragma unroll
        for (int m = 0; m < N / 2; m++)
            p[m] = (unsigned)col[pt.lo[m] * (RS / 2)] | ((unsigned)col[pt.hi[m] * (RS / 2)] << 16);
    }
    __builtin_amdgcn_wave_barrier();
    int t[N];
    constexpr bool ASM = (VARIANT & 128) != 0;
    Idct1D<N, ASM>::run(p, t, 64);                          // + (1 << 6), then >> 7, clip_int16
    {
        unsigned short *dst = reinterpret_cast<unsigned short *>(blk) + slot_of_rt(N, i);
#pragma unroll
        for (int r = 0; r < N; r += 2) {
            const unsigned pk = sat_pack_i16(t[r] >> 7, t[r + 1] >> 7);
            dst[r * (RS / 2)]       = (unsigned short)(pk & 0xffffu);
            dst[(r + 1) * (RS / 2)] = (unsigned short)(pk >> 16);
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- C. pass 2: row i, already in pair order in LDS
    {
        const u32x4 *rowp = reinterpret_cast<const u32x4 *>(blk + i * RS);
#pragma unroll
        for (int q = 0; q < N / 8; q++) {
            const u32x4 v = rowp[q];
            p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
        }
    }
    const int shift2 = 20 - bit_depth;
    Idct1D<N, ASM>::run(p, t, 1 << (shift2 - 1));
#pragma unroll
    for (int k = 0; k < N; k++) t[k] >>= shift2;

    // ---- D. residual row + prediction row -> plane
    if constexpr (WG_EPILOGUE) {
        // Workgroup-wide coalesced epilogue: the 4 * BPW blocks of the workgroup form one strip of 256 samples.  Residual rows go to LDS
        // as int16, then thread T takes 16 bytes of pixels of strip row T / CPR: a wave instruction covers whole 256-byte (8-bit) /
        // 512-byte (16-bit) row segments -- full 128-byte lines wherever the blocks of the batch are horizontal neighbours (they are in
        // z-order / raster job lists).  The 64-sample strips of the wave-private form (VARIANT 16) give the memory pipeline twice as many
        // line requests per byte at 8 bit; profiles/r02*_ab_tu_variants.txt.
        constexpr int SRS = 512 + 16;                          // bytes per strip row: 256 samples of int16 + pad
        constexpr int PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 256 / CH_PX, RPI = 256 / CPR, ITER = N / RPI;
        static_assert(N * SRS + 4 * L::BPW * 8 <= 4 * L::WAVE_BYTES, "output strip + job records must fit the workgroup's tiles");
        unsigned *jrec = reinterpret_cast<unsigned *>(lds + N * SRS);
        __syncthreads();                                       // every wave has consumed its coefficient tile
        {
            u32x4 *d = reinterpret_cast<u32x4 *>(lds + i * SRS + (wave * L::BPW + g) * (N * 2));
#pragma unroll
            for (int q = 0; q < N / 8; q++) {
                u32x4 v = { sat_pack_i16(t[8 * q], t[8 * q + 1]), sat_pack_i16(t[8 * q + 2], t[8 * q + 3]),
                            sat_pack_i16(t[8 * q + 4], t[8 * q + 5]), sat_pack_i16(t[8 * q + 6], t[8 * q + 7]) };
                d[q] = v;
            }
            if (i == 0) { jrec[2 * (wave * L::BPW + g)] = jraw.x; jrec[2 * (wave * L::BPW + g) + 1] = valid ? (jraw.y & 0xffu) : 0xffffffffu; }
        }
        __syncthreads();
        const int tid = threadIdx.x, c = tid % CPR, r0 = tid / CPR;
        const int bsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;
        const unsigned oxy = jrec[2 * bsel], opl = jrec[2 * bsel + 1];
        const bool ovalid = opl != 0xffffffffu;
        const int ostride = PLANE_STRIDE3(planes, opl);
        unsigned char *obase = PLANE_PTR3(planes, opl) + (size_t)(oxy >> 16) * ostride + (size_t)((oxy & 0xffff) + pxoff) * PXB;
        const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
        u32x4 pr[ITER];
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            pr[k] = u32x4{ 0, 0, 0, 0 };
            if (ovalid) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride);
        }
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const int rr = r0 + k * RPI;
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(lds + rr * SRS + c * CH_PX * 2);
            u32x4 o;
            if constexpr (PXB == 1) {
                const u32x4 ra = rp[0], rb = rp[1];
                const unsigned res[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
                const unsigned pd[4] = { pr[k].x, pr[k].y, pr[k].z, pr[k].w };
                unsigned od[4];
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) {
                    const unsigned u01 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c010c00u), u23 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c030c02u);
                    const unsigned a01 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4]), bitcast<s16x2>(u01)));
                    const unsigned a23 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4 + 1]), bitcast<s16x2>(u23)));
                    od[d4] = sat_pack_u8_i16(a01) | (sat_pack_u8_i16(a23) << 16);
                }
                o = u32x4{ od[0], od[1], od[2], od[3] };
            } else {
                const u32x4 ra = rp[0];
                o = u32x4{ add_clamp_upx2(ra.x, pr[k].x, max2), add_clamp_upx2(ra.y, pr[k].y, max2),
                           add_clamp_upx2(ra.z, pr[k].z, max2), add_clamp_upx2(ra.w, pr[k].w, max2) };
            }
            if (ovalid) *reinterpret_cast<u32x4 *>(obase + (size_t)rr * ostride) = o;
        }
    } else if constexpr ((VARIANT & 16) && LOG2N >= 4) {
        // Coalesced epilogue: the wave's 64/N blocks form one 64-sample-wide strip.  Residual rows go to LDS as int16
        // (natural order), then lane L takes 16 bytes of pixels of row L / CPR: one wave instruction covers RPI whole
        // rows of the strip, i.e. full 64-byte (8-bit) / 128-byte (16-bit) segments per row instead of 16-byte pieces.
        constexpr int ORS = 144;                               // 128 bytes of int16 per strip row + 16 pad
        constexpr int PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 64 / CH_PX, RPI = 64 / CPR, ITER = N / RPI;
        static_assert(N * ORS <= L::WAVE_BYTES, "output strip must fit the wave tile");
        unsigned char *wbase = lds + wave * L::WAVE_BYTES;
        __builtin_amdgcn_wave_barrier();
        {
            u32x4 *d = reinterpret_cast<u32x4 *>(wbase + i * ORS + g * (N * 2));
#pragma unroll
            for (int q = 0; q < N / 8; q++) {
                u32x4 v = { sat_pack_i16(t[8 * q], t[8 * q + 1]), sat_pack_i16(t[8 * q + 2], t[8 * q + 3]),
                            sat_pack_i16(t[8 * q + 4], t[8 * q + 5]), sat_pack_i16(t[8 * q + 6], t[8 * q + 7]) };
                d[q] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int c = lane % CPR, r0 = lane / CPR;
        const int gsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;
        const unsigned oxy = __shfl(jraw.x, gsel * N), opl = __shfl(jraw.y, gsel * N) & 0xff;
        const bool ovalid = job0 + gsel < njobs;
        const int ostride = PLANE_STRIDE3(planes, opl);
        unsigned char *obase = PLANE_PTR3(planes, opl) + (size_t)(oxy >> 16) * ostride + (size_t)((oxy & 0xffff) + pxoff) * PXB;
        const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const int rr = r0 + k * RPI;
            u32x4 pr = { 0, 0, 0, 0 };
            if (ovalid) pr = *reinterpret_cast<const u32x4 *>(obase + (size_t)rr * ostride);
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(wbase + rr * ORS + c * CH_PX * 2);
            u32x4 o;
            if constexpr (PXB == 1) {
                const u32x4 ra = rp[0], rb = rp[1];
                const unsigned res[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
                const unsigned pd[4] = { pr.x, pr.y, pr.z, pr.w };
                unsigned od[4];
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) {
                    const unsigned u01 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c010c00u), u23 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c030c02u);
                    if constexpr (ASM) {      // 8-bit content only (max2 == 255): saturating add, then clamp+pack in one op
                        const unsigned a01 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4]), bitcast<s16x2>(u01)));
                        const unsigned a23 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4 + 1]), bitcast<s16x2>(u23)));
                        od[d4] = sat_pack_u8_i16(a01) | (sat_pack_u8_i16(a23) << 16);
                    } else {
                        const unsigned s01 = add_clamp_px2(res[2 * d4], u01, max2), s23 = add_clamp_px2(res[2 * d4 + 1], u23, max2);
                        od[d4] = __builtin_amdgcn_perm(s23, s01, 0x06040200u);
                    }
                }
                o = u32x4{ od[0], od[1], od[2], od[3] };
            } else {
                const u32x4 ra = rp[0];
                o = u32x4{ add_clamp_upx2(ra.x, pr.x, max2), add_clamp_upx2(ra.y, pr.y, max2),
                           add_clamp_upx2(ra.z, pr.z, max2), add_clamp_upx2(ra.w, pr.w, max2) };
            }
            if (ovalid) *reinterpret_cast<u32x4 *>(obase + (size_t)rr * ostride) = o;
        }
    } else {
        if constexpr (!(VARIANT & 1)) load_row<N, Pixel>(row, px, valid);
        finish_row<N, Pixel>(row, px, t, bit_depth, valid);
    }
}

template <int LOG2N, typename Pixel, int VARIANT>
__global__ __launch_bounds__(256) void tu_idct_add_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs,
                                                          int njobs, const int16_t *__restrict__ coeffs, int bit_depth)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TuLayout<LOG2N>::WAVE_BYTES];
    tu_idct_add_body<LOG2N, Pixel, VARIANT>(lds, blockIdx.x, planes, jobs, njobs, coeffs, bit_depth);
}

// ------------------------------------------------------------------ 32x32 IDCT + add on the matrix cores
// The VALU form above is issue-bound: 737 VALU instructions per wave (2 blocks) x 4 cycles = the 0.93 ms it takes for 2^20 blocks
// (DESIGN.md 3.1).  The transform IS a matrix product, and v_mfma_i32_32x32x32_i8 multiplies int8 exactly; int16 inputs go in as two
// byte planes:      x = 256 * hi + (lo + 128),  hi = x >> 8,  lo = (x & 255) - 128  (both int8)
//   =>  sum_k T[k][y] * x[k] = 256 * sum T*hi + sum T*lo + 128 * sum_k T[k][y]            (|T| <= 90: exact in int32)
// i.e. two MFMAs per pass and block, the constant and the rounding term preloaded into the accumulator of the second one.
//
// Fragment layout (cdna_hip_programming.md "Fragment layout"; tools/probe_mfma_layout.py checks it on the device):
// A[m = lane & 31][k <- (lane >> 5, byte)], B[k][n = lane & 31] with the same k map, D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31].
// Which k a (half, byte) slot means is ours to choose as long as both operands agree:
//   pass 1 (columns):  D1[c][y] = sum_k C[k][c] * T[k][y].  A = the block, read COLUMN-wise out of its plain row-major LDS copy with
//           ds_read_b64_tr_b16 (each 16-lane group takes a [4 rows][16 columns] piece, a lane ends up with 4 consecutive rows of its
//           column): lane (c = l & 31, h = l >> 5) holds rows 16h .. 16h + 15, byte j = row k1(h, j) = 16h + j.  B = T in that order
//           (constant).  D1 comes out with lane = y and registers = c in the order k2(h, r) = (r & 3) + 8 (r >> 2) + 4h - exactly an operand of
//   pass 2 (rows):     D2[x][y] = sum_c T[c][x] * P1[y][c].  A = T in the k2 order (constant), B = P1 straight from the registers.
//           D2: lane = y, registers = x (groups of 4 consecutive x) -> 8-byte LDS writes into the strip tile of the coalesced epilogue.
// Memory side: everything moves in 16-byte accesses (a wave64 memory instruction costs the same address cycles whatever its width -
// a first version with 2-byte column loads straight from HBM was bit-exact and slower than the VALU kernel).  Each wave loops over block
// pairs, three deep: pair p is in LDS, pair p+1 in registers on its way to LDS, pair p+2 in flight from HBM (job records one further).
// tools/emulate_idct32_mfma.py runs the index algebra in numpy against the oracle.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int mfma_k1(int h, int j) { return 16 * h + j; }
__host__ __device__ constexpr int mfma_k2(int h, int j) { return (j & 3) + 8 * (j >> 2) + 4 * h; }
struct MfmaIdctTabs {
    int b1[64][4];        // pass-1 B operand of lane l: bytes T[k1(h, j)][l & 31]
    int a2[64][4];        // pass-2 A operand of lane l: bytes T[k2(h, j)][l & 31]
    int colsum[32];       // sum_k T[k][y]
};
constexpr MfmaIdctTabs make_mfma_idct_tabs()
{
    MfmaIdctTabs t{};
    for (int l = 0; l < 64; l++)
        for (int d = 0; d < 4; d++) {
            unsigned b = 0, a = 0;
            for (int e = 0; e < 4; e++) {
                const int j = 4 * d + e;
                b |= (unsigned)(dct32(mfma_k1(l >> 5, j), l & 31) & 0xff) << (8 * e);
                a |= (unsigned)(dct32(mfma_k2(l >> 5, j), l & 31) & 0xff) << (8 * e);
            }
            t.b1[l][d] = (int)b; t.a2[l][d] = (int)a;
        }
    for (int y = 0; y < 32; y++) {
        int sum = 0;
        for (int k = 0; k < 32; k++) sum += dct32(k, y);
        t.colsum[y] = sum;
    }
    return t;
}
__device__ const MfmaIdctTabs kMfmaIdct = make_mfma_idct_tabs();

// two dwords of int16 pairs -> the int8 plane of their high / low bytes (low: - 128, i.e. bit 7 flipped)
__device__ __forceinline__ int hi_bytes(unsigned w0, unsigned w1) { return (int)__builtin_amdgcn_perm(w1, w0, 0x07050301u); }
__device__ __forceinline__ int lo_bytes(unsigned w0, unsigned w1) { return (int)(__builtin_amdgcn_perm(w1, w0, 0x06040200u) ^ 0x80808080u); }

// ds_read_b64_tr_b16: within each group of 16 lanes, lane i supplies the (8-byte aligned) address of 4 int16; lane l receives element
// l & 3 of what lanes (l >> 2), 4 + (l >> 2), 8 + (l >> 2), 12 + (l >> 2) fetched.  With lane i pointing at row i >> 2, columns 4 (i & 3) ..
// of a [4][16] piece, lane l gets rows 0..3 of column l.
__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned char *p)
{
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (v4s __attribute__((address_space(3))) *)(__attribute__((address_space(3))) unsigned char *)p);
    return bitcast<u32x2>(r);
}

#ifdef OHEVC_LAB      // measurement forms kept for A/B (make LAB=1): the loop kernels the shipped tile kernel was derived from
// VARIANT bit 0: fetch the prediction rows before the transform; bit 1: no register prefetch of the next pair (for the A/B)
template <typename Pixel, int VARIANT>
__global__ __launch_bounds__(256) void tu_idct32_mfma_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int njobs,
                                                             const int16_t *__restrict__ coeffs, int bit_depth)
{
    constexpr int N = 32, ORS = 144, STRIP = N * ORS, COEF = 2 * N * N * 2, WAVE_BYTES = COEF + STRIP;   // 4096 + 4608 per wave
    constexpr int PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 64 / CH_PX, RPI = 64 / CPR, ITER = N / RPI;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVE_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    unsigned char *ctile = lds + wave * WAVE_BYTES, *strip = ctile + COEF;
    const v4i b1 = { kMfmaIdct.b1[lane][0], kMfmaIdct.b1[lane][1], kMfmaIdct.b1[lane][2], kMfmaIdct.b1[lane][3] };
    const v4i a2 = { kMfmaIdct.a2[lane][0], kMfmaIdct.a2[lane][1], kMfmaIdct.a2[lane][2], kMfmaIdct.a2[lane][3] };
    const int shift2 = 20 - bit_depth;
    v16i init1, init2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        init1[r] = 64 + 128 * kMfmaIdct.colsum[n];                                       // + (1 << 6), then >> 7
        init2[r] = (1 << (shift2 - 1)) + 128 * kMfmaIdct.colsum[mfma_k2(h, r)];
    }
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    // transposing read: this lane's piece of the [4 rows][16 columns] block of its 16-lane group (rows 16h + 4t .., columns 16 * bit 4 ..)
    const int u = lane & 15;
    const unsigned char *trp = ctile + (16 * h + (u >> 2)) * (N * 2) + (16 * ((lane >> 4) & 1) + 4 * (u & 3)) * 2;
    const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
    const int c = lane % CPR, r0 = lane / CPR, gsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;

    const int npairs = (njobs + 1) >> 1, stride = gridDim.x * 4;
    const int first = blockIdx.x * 4 + wave;
    if (first >= npairs) return;
    auto job_of = [&](int pair, int g) {
        const int j = pair * 2 + g;
        return reinterpret_cast<const u32x4 *>(jobs)[j < njobs ? j : njobs - 1];
    };
    auto fetch = [&](const u32x4 &j0, const u32x4 &j1, u32x4 *cq) {            // 4 x 16 bytes per lane = 2 blocks
        const u32x4 *s0 = reinterpret_cast<const u32x4 *>(coeffs + j0.z), *s1 = reinterpret_cast<const u32x4 *>(coeffs + j1.z);
        cq[0] = s0[lane]; cq[1] = s0[64 + lane]; cq[2] = s1[lane]; cq[3] = s1[64 + lane];
    };
    auto to_lds = [&](const u32x4 *cq) {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<u32x4 *>(ctile + (q * 64 + lane) * 16) = cq[q];
    };
    // job records: current pair, +1 (its coefficients are in `cq`), +2 (fetched this iteration), +3 (record load in flight)
    u32x4 jc[2] = { job_of(first, 0), job_of(first, 1) };
    u32x4 j1[2] = { job_of(first + stride, 0), job_of(first + stride, 1) };
    u32x4 j2[2] = { job_of(first + 2 * stride, 0), job_of(first + 2 * stride, 1) };
    u32x4 cq[4];
    fetch(jc[0], jc[1], cq);
    to_lds(cq);
    fetch(j1[0], j1[1], cq);
    for (int pair = first; pair < npairs; pair += stride) {
        const int job0 = pair * 2;
        const u32x4 j3[2] = { job_of(pair + 3 * stride, 0), job_of(pair + 3 * stride, 1) };
        __builtin_amdgcn_wave_barrier();
        unsigned w[2][8];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const u32x2 v = lds_read_tr16(trp + g * (N * N * 2) + t * (4 * N * 2));
                w[g][2 * t] = v.x; w[g][2 * t + 1] = v.y;
            }
        // epilogue addressing of this pair; optionally its prediction rows now
        const unsigned oxy = gsel ? jc[1].x : jc[0].x, opl = (gsel ? jc[1].y : jc[0].y) & 0xff;
        const bool ovalid = job0 + gsel < njobs;
        const int ostride = PLANE_STRIDE3(planes, opl);
        unsigned char *obase = PLANE_PTR3(planes, opl) + (size_t)(oxy >> 16) * ostride + (size_t)((oxy & 0xffff) + pxoff) * PXB;
        u32x4 pr[ITER];
        if constexpr (VARIANT & 1) {
#pragma unroll
            for (int k = 0; k < ITER; k++) {
                pr[k] = u32x4{ 0, 0, 0, 0 };
                if (ovalid) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the LDS copy is consumed (DS operations of a wave execute in order): the next pair moves in, the one after is requested
        if constexpr (!(VARIANT & 2)) {
            to_lds(cq);
            fetch(j2[0], j2[1], cq);
        }
#pragma unroll
        for (int g = 0; g < 2; g++) {
            // ---- pass 1
            const v4i ahi = { hi_bytes(w[g][0], w[g][1]), hi_bytes(w[g][2], w[g][3]), hi_bytes(w[g][4], w[g][5]), hi_bytes(w[g][6], w[g][7]) };
            const v4i alo = { lo_bytes(w[g][0], w[g][1]), lo_bytes(w[g][2], w[g][3]), lo_bytes(w[g][4], w[g][5]), lo_bytes(w[g][6], w[g][7]) };
            const v16i dhi = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, b1, zero, 0, 0, 0);
            const v16i dlo = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, b1, init1, 0, 0, 0);
            unsigned pk[8];
#pragma unroll
            for (int q = 0; q < 8; q++)
                pk[q] = sat_pack_i16(((dhi[2 * q] << 8) + dlo[2 * q]) >> 7, ((dhi[2 * q + 1] << 8) + dlo[2 * q + 1]) >> 7);
            // ---- pass 2
            const v4i bhi = { hi_bytes(pk[0], pk[1]), hi_bytes(pk[2], pk[3]), hi_bytes(pk[4], pk[5]), hi_bytes(pk[6], pk[7]) };
            const v4i blo = { lo_bytes(pk[0], pk[1]), lo_bytes(pk[2], pk[3]), lo_bytes(pk[4], pk[5]), lo_bytes(pk[6], pk[7]) };
            const v16i ehi = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, bhi, zero, 0, 0, 0);
            const v16i elo = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, blo, init2, 0, 0, 0);
            // ---- residual row y = n of block g: registers 4q'..4q'+3 are x = 8q' + 4h .. + 3
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                u32x2 v;
                v.x = sat_pack_i16(((ehi[4 * qq] << 8) + elo[4 * qq]) >> shift2, ((ehi[4 * qq + 1] << 8) + elo[4 * qq + 1]) >> shift2);
                v.y = sat_pack_i16(((ehi[4 * qq + 2] << 8) + elo[4 * qq + 2]) >> shift2, ((ehi[4 * qq + 3] << 8) + elo[4 * qq + 3]) >> shift2);
                *reinterpret_cast<u32x2 *>(strip + n * ORS + g * (N * 2) + (8 * qq + 4 * h) * 2) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- epilogue: as in tu_idct_add_body (coalesced strip form): lane L takes 16 bytes of pixels of row L / CPR
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const int rr = r0 + k * RPI;
            if constexpr (!(VARIANT & 1)) {
                pr[k] = u32x4{ 0, 0, 0, 0 };
                if (ovalid) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)rr * ostride);
            }
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(strip + rr * ORS + c * CH_PX * 2);
            u32x4 o;
            if constexpr (PXB == 1) {
                const u32x4 ra = rp[0], rb = rp[1];
                const unsigned res[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
                const unsigned pd[4] = { pr[k].x, pr[k].y, pr[k].z, pr[k].w };
                unsigned od[4];
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) {
                    const unsigned u01 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c010c00u), u23 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c030c02u);
                    const unsigned a01 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4]), bitcast<s16x2>(u01)));
                    const unsigned a23 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4 + 1]), bitcast<s16x2>(u23)));
                    od[d4] = sat_pack_u8_i16(a01) | (sat_pack_u8_i16(a23) << 16);
                }
                o = u32x4{ od[0], od[1], od[2], od[3] };
            } else {
                const u32x4 ra = rp[0];
                o = u32x4{ add_clamp_upx2(ra.x, pr[k].x, max2), add_clamp_upx2(ra.y, pr[k].y, max2),
                           add_clamp_upx2(ra.z, pr[k].z, max2), add_clamp_upx2(ra.w, pr[k].w, max2) };
            }
            if (ovalid) *reinterpret_cast<u32x4 *>(obase + (size_t)rr * ostride) = o;
        }
        if constexpr (VARIANT & 2) {               // unpipelined form: fetch the next pair only now
            __builtin_amdgcn_wave_barrier();
            to_lds(cq);
            fetch(j2[0], j2[1], cq);
        }
        jc[0] = j1[0]; jc[1] = j1[1]; j1[0] = j2[0]; j1[1] = j2[1]; j2[0] = j3[0]; j2[1] = j3[1];
    }
}

// ------------------------------------------------------------------ 32x32 IDCT + add on the matrix cores, workgroup tiles
// What the measurements of round 2 say about the two kernels above (profiles/r02d_sq_counters_dot2_vs_mfma.txt): the dot2 form is
// VALU-issue bound (737 VALU instructions per block pair keep the SIMDs 97 % busy: its time does not depend on the memory pattern at
// all), the matrix-core form above needs 308 but waits 73 % of its life in s_waitcnt: its loop loads and stores under exec masks
// (`if (ovalid)`), so the compiler cannot count what is in flight and falls back to vmcnt(0) - the prefetch never overlaps anything.
// And the box itself moves this read/write mix at 5.2-5.3 TB/s when the pixels travel as 64-byte row pieces, 5.65-5.7 TB/s as 128- /
// 256-byte pieces (tools/hbm_probe.hip, profiles/r02e_hbm_probe.jsonl).  So this form:
//   * a workgroup owns a TILE of 8 blocks (256 samples wide when the blocks are horizontal neighbours, as they are in z-order / raster
//     job lists): wave w transforms blocks 2w, 2w + 1 on the matrix cores exactly as above, the residuals of all four waves meet in one
//     LDS strip and the epilogue moves whole 256-byte (8 bit) / 512-byte (16 bit) row segments;
//   * the loop body has NO branch and no exec-masked memory operation (the host hands over whole tiles only; prefetch indices are
//     clamped instead of predicated), so every s_waitcnt is an exact count: the coefficients of tile t + 1 (non-temporal loads: read
//     exactly once) and the job records of tile t + 2 stay in flight across the transform and the stores of tile t;
//   * prediction rows are requested at the top of an iteration, before the coefficient hand-over, and consumed at its end.
template <typename Pixel, int MINW, bool ABLATE = false>
__global__ __launch_bounds__(256, MINW) void tu_idct32_tile_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int ntiles,
                                                             const int16_t *__restrict__ coeffs, int bit_depth)
{
    constexpr int N = 32, CT = 2 * N * N * 2, SRS = 512 + 16, STRIP = N * SRS;      // 4096 bytes of coefficients per wave; strip rows of 256 int16 + pad
    constexpr int PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 256 / CH_PX, RPI = 256 / CPR, ITER = N / RPI;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * CT + STRIP];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 31, h = lane >> 5;
    unsigned char *ctile = lds + wave * CT, *strip = lds + 4 * CT;
    const v4i b1 = { kMfmaIdct.b1[lane][0], kMfmaIdct.b1[lane][1], kMfmaIdct.b1[lane][2], kMfmaIdct.b1[lane][3] };
    const v4i a2 = { kMfmaIdct.a2[lane][0], kMfmaIdct.a2[lane][1], kMfmaIdct.a2[lane][2], kMfmaIdct.a2[lane][3] };
    const int shift2 = 20 - bit_depth;
    v16i init1, init2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        init1[r] = 64 + 128 * kMfmaIdct.colsum[n];                                       // + (1 << 6), then >> 7
        init2[r] = (1 << (shift2 - 1)) + 128 * kMfmaIdct.colsum[mfma_k2(h, r)];
    }
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const int u = lane & 15;
    const unsigned char *trp = ctile + (16 * h + (u >> 2)) * (N * 2) + (16 * ((lane >> 4) & 1) + 4 * (u & 3)) * 2;
    const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
    // epilogue: thread T owns 16 bytes of pixels of strip rows r0 + k * RPI, inside block `bsel` of the tile
    const int c = tid % CPR, r0 = tid / CPR, bsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;

    const int stride = gridDim.x, last = ntiles - 1;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;                                  // workgroup-uniform
    // job records (16 bytes: x | y << 16, plane | .., coeff_off, ..): only the words a role needs are loaded
    auto coff = [&](int t, int b) { return reinterpret_cast<const unsigned *>(jobs)[((size_t)(t < last ? t : last) * 8 + b) * 4 + 2]; };
    auto erec = [&](int t) { return reinterpret_cast<const u32x2 *>(jobs)[((size_t)(t < last ? t : last) * 8 + bsel) * 2]; };
    auto fetch = [&](unsigned o0, unsigned o1, u32x4 *cq) {                    // 4 x 16 bytes per lane = this wave's 2 blocks
        const u32x4 *s0 = reinterpret_cast<const u32x4 *>(coeffs + o0), *s1 = reinterpret_cast<const u32x4 *>(coeffs + o1);
        cq[0] = __builtin_nontemporal_load(s0 + lane); cq[1] = __builtin_nontemporal_load(s0 + 64 + lane);
        cq[2] = __builtin_nontemporal_load(s1 + lane); cq[3] = __builtin_nontemporal_load(s1 + 64 + lane);
    };
    auto to_lds = [&](const u32x4 *cq) {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<u32x4 *>(ctile + (q * 64 + lane) * 16) = cq[q];
    };
    // in flight at the top of iteration t: the coefficients of tile t + 1 (registers), the coefficient offsets of tile t + 2 (jn) and,
    // requested there, of tile t + 3; this thread's epilogue records of tiles t, t + 1 and (requested) t + 2
    u32x2 je = erec(tile), jen = erec(tile + stride);
    u32x4 cq[4];
    fetch(coff(tile, 2 * wave), coff(tile, 2 * wave + 1), cq);
    to_lds(cq);
    fetch(coff(tile + stride, 2 * wave), coff(tile + stride, 2 * wave + 1), cq);
    unsigned jn[2] = { coff(tile + 2 * stride, 2 * wave), coff(tile + 2 * stride, 2 * wave + 1) };     // the tile whose coefficients are requested next
    for (; tile < ntiles; tile += stride) {
        const unsigned jnn[2] = { coff(tile + 3 * stride, 2 * wave), coff(tile + 3 * stride, 2 * wave + 1) };
        const u32x2 jenn = erec(tile + 2 * stride);
        __syncthreads();                                         // the previous tile's epilogue has read the strip
        unsigned w[2][8];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const u32x2 v = lds_read_tr16(trp + g * (N * N * 2) + t * (4 * N * 2));
                w[g][2 * t] = v.x; w[g][2 * t + 1] = v.y;
            }
        // prediction rows of this tile: requested now, consumed after the transform
        const unsigned opl = je.y & 0xff;
        const int ostride = PLANE_STRIDE3(planes, opl);
        unsigned char *obase = PLANE_PTR3(planes, opl) + (size_t)(je.x >> 16) * ostride + (size_t)((je.x & 0xffff) + pxoff) * PXB;
        u32x4 pr[ITER];
#pragma unroll
        for (int k = 0; k < ITER; k++) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride);
        __builtin_amdgcn_wave_barrier();
        // this wave's LDS copy is consumed (DS operations of a wave execute in order): the next tile's blocks move in, the ones after are requested
        to_lds(cq);
        fetch(jn[0], jn[1], cq);
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if constexpr (ABLATE) {          // bottleneck analysis only (NOT a transform): same memory and LDS traffic, no matrix work
#pragma unroll
                for (int qq = 0; qq < 4; qq++)
                    *reinterpret_cast<u32x2 *>(strip + n * SRS + (2 * wave + g) * (N * 2) + (8 * qq + 4 * h) * 2) = u32x2{ w[g][2 * qq], w[g][2 * qq + 1] };
                continue;
            }
            // ---- pass 1
            const v4i ahi = { hi_bytes(w[g][0], w[g][1]), hi_bytes(w[g][2], w[g][3]), hi_bytes(w[g][4], w[g][5]), hi_bytes(w[g][6], w[g][7]) };
            const v4i alo = { lo_bytes(w[g][0], w[g][1]), lo_bytes(w[g][2], w[g][3]), lo_bytes(w[g][4], w[g][5]), lo_bytes(w[g][6], w[g][7]) };
            const v16i dhi = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, b1, zero, 0, 0, 0);
            const v16i dlo = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, b1, init1, 0, 0, 0);
            unsigned pk[8];
#pragma unroll
            for (int q = 0; q < 8; q++)
                pk[q] = sat_pack_i16(((dhi[2 * q] << 8) + dlo[2 * q]) >> 7, ((dhi[2 * q + 1] << 8) + dlo[2 * q + 1]) >> 7);
            // ---- pass 2
            const v4i bhi = { hi_bytes(pk[0], pk[1]), hi_bytes(pk[2], pk[3]), hi_bytes(pk[4], pk[5]), hi_bytes(pk[6], pk[7]) };
            const v4i blo = { lo_bytes(pk[0], pk[1]), lo_bytes(pk[2], pk[3]), lo_bytes(pk[4], pk[5]), lo_bytes(pk[6], pk[7]) };
            const v16i ehi = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, bhi, zero, 0, 0, 0);
            const v16i elo = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, blo, init2, 0, 0, 0);
            // ---- residual row y = n of block 2 * wave + g: registers 4q'..4q'+3 are x = 8q' + 4h .. + 3
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                u32x2 v;
                v.x = sat_pack_i16(((ehi[4 * qq] << 8) + elo[4 * qq]) >> shift2, ((ehi[4 * qq + 1] << 8) + elo[4 * qq + 1]) >> shift2);
                v.y = sat_pack_i16(((ehi[4 * qq + 2] << 8) + elo[4 * qq + 2]) >> shift2, ((ehi[4 * qq + 3] << 8) + elo[4 * qq + 3]) >> shift2);
                *reinterpret_cast<u32x2 *>(strip + n * SRS + (2 * wave + g) * (N * 2) + (8 * qq + 4 * h) * 2) = v;
            }
        }
        __syncthreads();                                         // the strip is complete
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const int rr = r0 + k * RPI;
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(strip + rr * SRS + c * CH_PX * 2);
            u32x4 o;
            if constexpr (PXB == 1) {
                const u32x4 ra = rp[0], rb = rp[1];
                const unsigned res[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
                const unsigned pd[4] = { pr[k].x, pr[k].y, pr[k].z, pr[k].w };
                unsigned od[4];
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) {
                    const unsigned u01 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c010c00u), u23 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c030c02u);
                    const unsigned a01 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4]), bitcast<s16x2>(u01)));
                    const unsigned a23 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4 + 1]), bitcast<s16x2>(u23)));
                    od[d4] = sat_pack_u8_i16(a01) | (sat_pack_u8_i16(a23) << 16);
                }
                o = u32x4{ od[0], od[1], od[2], od[3] };
            } else {
                const u32x4 ra = rp[0];
                o = u32x4{ add_clamp_upx2(ra.x, pr[k].x, max2), add_clamp_upx2(ra.y, pr[k].y, max2),
                           add_clamp_upx2(ra.z, pr[k].z, max2), add_clamp_upx2(ra.w, pr[k].w, max2) };
            }
            *reinterpret_cast<u32x4 *>(obase + (size_t)rr * ostride) = o;
        }
        jn[0] = jnn[0]; jn[1] = jnn[1]; je = jen; jen = jenn;
    }
}

// ------------------------------------------------------------------ the same tiles, two of them ahead
// The tile kernel above without its matrix work runs no faster (profiles/r02h): with 3-4 workgroups per CU and one tile of coefficients
// per workgroup in flight a CU has ~96 KB outstanding, and this memory system needs about twice that to reach the 5.7 TB/s the same
// access pattern gets from a kernel that is nothing but loads (tools/hbm_probe: mix31_tiled256_nt).  The register file is the only
// place big enough to park requests (512 KB per CU against 160 KB of LDS), so this form spends registers on prefetch instead of on
// the transform:
//   * pass 2 runs TRANSPOSED (A = the pass-1 fragment, B = the constant matrix): the byte-plane correction 128 * sum_c T[c][x] then
//     depends on the lane only, like pass 1's, and both become one register each instead of sixteen;
//   * the low-byte product accumulates straight onto (high-byte product << 8) + that constant: 16 v_lshl_add between the two MFMAs of
//     a pass instead of 16 after them, and only one 16-register accumulator is alive at a time;
//   * what that frees holds a SECOND tile of coefficients: tiles t + 1 and t + 2 are in registers, t + 3 is requested while t is
//     transformed; the loop is unrolled by two so that each register set keeps its name (no copies, every s_waitcnt an exact count).
// Residuals now come out column-wise (lane = x, registers = 16 rows): the strip takes them as 2-byte stores.
template <typename Pixel>
__global__ __launch_bounds__(256, 4) void tu_idct32_tile2_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int ntiles,
                                                                 const int16_t *__restrict__ coeffs, int bit_depth)
{
    constexpr int N = 32, CT = 2 * N * N * 2, SRS = 512 + 16, STRIP = N * SRS;
    constexpr int PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 256 / CH_PX, RPI = 256 / CPR, ITER = N / RPI;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * CT + STRIP];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 31, h = lane >> 5;
    unsigned char *ctile = lds + wave * CT, *strip = lds + 4 * CT;
    const v4i b1 = { kMfmaIdct.b1[lane][0], kMfmaIdct.b1[lane][1], kMfmaIdct.b1[lane][2], kMfmaIdct.b1[lane][3] };
    const v4i a2 = { kMfmaIdct.a2[lane][0], kMfmaIdct.a2[lane][1], kMfmaIdct.a2[lane][2], kMfmaIdct.a2[lane][3] };
    const int shift2 = 20 - bit_depth;
    const int k1 = 64 + 128 * kMfmaIdct.colsum[n], k2 = (1 << (shift2 - 1)) + 128 * kMfmaIdct.colsum[n];
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const int u = lane & 15;
    const unsigned char *trp = ctile + (16 * h + (u >> 2)) * (N * 2) + (16 * ((lane >> 4) & 1) + 4 * (u & 3)) * 2;
    const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
    const int c = tid % CPR, r0 = tid / CPR, bsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;
    unsigned char *scol = strip + (2 * wave) * (N * 2) + n * 2 + (4 * h) * SRS;          // residual (row 4h, column n) of this wave's first block

    const int stride = gridDim.x, last = ntiles - 1;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;                                  // workgroup-uniform
    auto coff = [&](int t, int b) { return reinterpret_cast<const unsigned *>(jobs)[((size_t)(t < last ? t : last) * 8 + b) * 4 + 2]; };
    auto erec = [&](int t) { return reinterpret_cast<const u32x2 *>(jobs)[((size_t)(t < last ? t : last) * 8 + bsel) * 2]; };
    auto fetch = [&](unsigned o0, unsigned o1, u32x4 (&cq)[4]) {
        const u32x4 *s0 = reinterpret_cast<const u32x4 *>(coeffs + o0), *s1 = reinterpret_cast<const u32x4 *>(coeffs + o1);
        cq[0] = __builtin_nontemporal_load(s0 + lane); cq[1] = __builtin_nontemporal_load(s0 + 64 + lane);
        cq[2] = __builtin_nontemporal_load(s1 + lane); cq[3] = __builtin_nontemporal_load(s1 + 64 + lane);
    };
    auto to_lds = [&](const u32x4 (&cq)[4]) {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<u32x4 *>(ctile + (q * 64 + lane) * 16) = cq[q];
    };
    // one tile: its coefficients are in LDS; `cq` holds the next tile's and is refilled with the tile at offsets (f0, f1)
    auto pixel_base = [&](const u32x2 je, int &ostride) {
        const unsigned opl = je.y & 0xff;
        ostride = PLANE_STRIDE3(planes, opl);
        return PLANE_PTR3(planes, opl) + (size_t)(je.x >> 16) * ostride + (size_t)((je.x & 0xffff) + pxoff) * PXB;
    };
    auto load_pred = [&](const u32x2 je, u32x4 (&pr)[ITER]) {
        int ostride;
        const unsigned char *obase = pixel_base(je, ostride);
#pragma unroll
        for (int k = 0; k < ITER; k++) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride);
    };
    // One tile: its coefficients are in LDS, its prediction rows in `pr`.  `cq` holds the next tile's coefficients and is refilled with
    // the tile at offsets (f0, f1); `prn` receives the next tile's prediction rows.  Requests leave in the order they are consumed
    // (prediction rows of t + 1, then coefficients of t + 3): a wait for the oldest one never drains a younger one.
    auto body = [&](u32x4 (&cq)[4], unsigned f0, unsigned f1, const u32x2 je, const u32x2 jen, const u32x4 (&pr)[ITER], u32x4 (&prn)[ITER]) {
        __syncthreads();                                         // the previous tile's epilogue has read the strip
        unsigned w[2][8];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const u32x2 v = lds_read_tr16(trp + g * (N * N * 2) + t * (4 * N * 2));
                w[g][2 * t] = v.x; w[g][2 * t + 1] = v.y;
            }
        load_pred(jen, prn);
        __builtin_amdgcn_wave_barrier();
        to_lds(cq);
        fetch(f0, f1, cq);
#pragma unroll
        for (int g = 0; g < 2; g++) {
            // ---- pass 1: D1[c][y], lane = y, registers = c
            const v4i ahi = { hi_bytes(w[g][0], w[g][1]), hi_bytes(w[g][2], w[g][3]), hi_bytes(w[g][4], w[g][5]), hi_bytes(w[g][6], w[g][7]) };
            const v4i alo = { lo_bytes(w[g][0], w[g][1]), lo_bytes(w[g][2], w[g][3]), lo_bytes(w[g][4], w[g][5]), lo_bytes(w[g][6], w[g][7]) };
            v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, b1, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = (d[r] << 8) + k1;
            d = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, b1, d, 0, 0, 0);
            unsigned pk[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pk[q] = sat_pack_i16(d[2 * q] >> 7, d[2 * q + 1] >> 7);
            // ---- pass 2, transposed: D2[y][x] = sum_c P1[y][c] * T[c][x]: A = the pass-1 fragment, B = the matrix; lane = x, registers = y
            const v4i phi = { hi_bytes(pk[0], pk[1]), hi_bytes(pk[2], pk[3]), hi_bytes(pk[4], pk[5]), hi_bytes(pk[6], pk[7]) };
            const v4i plo = { lo_bytes(pk[0], pk[1]), lo_bytes(pk[2], pk[3]), lo_bytes(pk[4], pk[5]), lo_bytes(pk[6], pk[7]) };
            v16i e = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, a2, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) e[r] = (e[r] << 8) + k2;
            e = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, a2, e, 0, 0, 0);
            // ---- register r = row (r & 3) + 8 (r >> 2) + 4 h of column n
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const unsigned v = sat_pack_i16(e[r] >> shift2, e[r + 1] >> shift2);
                unsigned char *p0 = scol + g * (N * 2) + ((r & 3) + 8 * (r >> 2)) * SRS;
                *reinterpret_cast<unsigned short *>(p0) = (unsigned short)(v & 0xffffu);
                *reinterpret_cast<unsigned short *>(p0 + SRS) = (unsigned short)(v >> 16);
            }
        }
        __syncthreads();                                         // the strip is complete
        int ostride;
        unsigned char *obase = pixel_base(je, ostride);
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const int rr = r0 + k * RPI;
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(strip + rr * SRS + c * CH_PX * 2);
            u32x4 o;
            if constexpr (PXB == 1) {
                const u32x4 ra = rp[0], rb = rp[1];
                const unsigned res[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
                const unsigned pd[4] = { pr[k].x, pr[k].y, pr[k].z, pr[k].w };
                unsigned od[4];
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) {
                    const unsigned u01 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c010c00u), u23 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c030c02u);
                    const unsigned a01 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4]), bitcast<s16x2>(u01)));
                    const unsigned a23 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4 + 1]), bitcast<s16x2>(u23)));
                    od[d4] = sat_pack_u8_i16(a01) | (sat_pack_u8_i16(a23) << 16);
                }
                o = u32x4{ od[0], od[1], od[2], od[3] };
            } else {
                const u32x4 ra = rp[0];
                o = u32x4{ add_clamp_upx2(ra.x, pr[k].x, max2), add_clamp_upx2(ra.y, pr[k].y, max2),
                           add_clamp_upx2(ra.z, pr[k].z, max2), add_clamp_upx2(ra.w, pr[k].w, max2) };
            }
            *reinterpret_cast<u32x4 *>(obase + (size_t)rr * ostride) = o;
        }
    };

    // at the top of a loop trip for tile t: LDS = t, pra = prediction rows of t, cqa = t + s, cqb = t + 2s; offsets j3 / j4 = tiles t + 3s /
    // t + 4s; epilogue records je0 / je1 / je2 = tiles t, t + s, t + 2s
    u32x4 cqa[4], cqb[4], pra[ITER], prb[ITER];
    fetch(coff(tile, 2 * wave), coff(tile, 2 * wave + 1), cqa);
    to_lds(cqa);
    u32x2 je0 = erec(tile), je1 = erec(tile + stride), je2 = erec(tile + 2 * stride);
    load_pred(je0, pra);
    fetch(coff(tile + stride, 2 * wave), coff(tile + stride, 2 * wave + 1), cqa);
    fetch(coff(tile + 2 * stride, 2 * wave), coff(tile + 2 * stride, 2 * wave + 1), cqb);
    unsigned j3[2] = { coff(tile + 3 * stride, 2 * wave), coff(tile + 3 * stride, 2 * wave + 1) };
    unsigned j4[2] = { coff(tile + 4 * stride, 2 * wave), coff(tile + 4 * stride, 2 * wave + 1) };
    auto trip = [&]() {                                          // two tiles
        const unsigned j5[2] = { coff(tile + 5 * stride, 2 * wave), coff(tile + 5 * stride, 2 * wave + 1) };
        const unsigned j6[2] = { coff(tile + 6 * stride, 2 * wave), coff(tile + 6 * stride, 2 * wave + 1) };
        const u32x2 je3 = erec(tile + 3 * stride), je4 = erec(tile + 4 * stride);
        body(cqa, j3[0], j3[1], je0, je1, pra, prb);
        body(cqb, j4[0], j4[1], je1, je2, prb, pra);
        j3[0] = j5[0]; j3[1] = j5[1]; j4[0] = j6[0]; j4[1] = j6[1]; je0 = je2; je1 = je3; je2 = je4;
        tile += 2 * stride;
    };
    // The first trip is peeled: the compiler's s_waitcnt counts at the loop head are the more cautious of "from the prologue" and "from
    // the previous trip"; behind a whole trip both histories are the same and the counts in the loop stay exact.
    if (tile + stride < ntiles) {
        trip();
        while (tile + stride < ntiles) trip();
    }
    if (tile < ntiles) body(cqa, j3[0], j3[1], je0, je1, pra, prb);      // an odd tile is left (what it requests is clamped and never used)
}

#endif  // OHEVC_LAB

// ------------------------------------------------------------------ one tile per workgroup, no loop
// The pure-traffic kernel below reaches the batch's ceiling (0.77 ms for 2^20 blocks at 8 bit) with nothing but occupancy: 8 short-lived
// workgroups per CU, every one with its whole tile in flight from its first instruction.  The persistent forms above stay 9 % short of it
// whatever their prefetch depth.  This form keeps the matrix-core transform of tile2 (transposed pass 2, chained accumulators: few
// registers) and drops the loop: the strip re-uses the coefficient tiles' LDS (17 KB per workgroup), registers decide the occupancy.
template <typename Pixel, int MINW>
__global__ __launch_bounds__(256, MINW) void tu_idct32_tile1_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int ntiles,
                                                                    const int16_t *__restrict__ coeffs, int bit_depth)
{
    constexpr int N = 32, CT = 2 * N * N * 2, SRS = 512 + 16, STRIP = N * SRS;
    constexpr int PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 256 / CH_PX, RPI = 256 / CPR, ITER = N / RPI;
    __shared__ __attribute__((aligned(16))) unsigned char lds[STRIP > 4 * CT ? STRIP : 4 * CT];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    unsigned char *ctile = lds + wave * CT, *strip = lds;
    const int c = tid % CPR, r0 = tid / CPR, bsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;
    // ---- everything this workgroup will read from HBM is requested first
    const unsigned *jw = reinterpret_cast<const unsigned *>(jobs) + (size_t)tile * 32;
    const unsigned o0 = jw[(2 * wave) * 4 + 2], o1 = jw[(2 * wave + 1) * 4 + 2];
    const u32x2 je = reinterpret_cast<const u32x2 *>(jw)[bsel * 2];
    const u32x4 *s0 = reinterpret_cast<const u32x4 *>(coeffs + o0), *s1 = reinterpret_cast<const u32x4 *>(coeffs + o1);
    u32x4 cq[4] = { __builtin_nontemporal_load(s0 + lane), __builtin_nontemporal_load(s0 + 64 + lane),
                    __builtin_nontemporal_load(s1 + lane), __builtin_nontemporal_load(s1 + 64 + lane) };
    const unsigned opl = je.y & 0xff;
    const int ostride = PLANE_STRIDE3(planes, opl);
    unsigned char *obase = PLANE_PTR3(planes, opl) + (size_t)(je.x >> 16) * ostride + (size_t)((je.x & 0xffff) + pxoff) * PXB;
    u32x4 pr[ITER];
#pragma unroll
    for (int k = 0; k < ITER; k++) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride);
    // ---- constants (while the loads fly)
    const v4i b1 = { kMfmaIdct.b1[lane][0], kMfmaIdct.b1[lane][1], kMfmaIdct.b1[lane][2], kMfmaIdct.b1[lane][3] };
    const v4i a2 = { kMfmaIdct.a2[lane][0], kMfmaIdct.a2[lane][1], kMfmaIdct.a2[lane][2], kMfmaIdct.a2[lane][3] };
    const int shift2 = 20 - bit_depth;
    const int k1 = 64 + 128 * kMfmaIdct.colsum[n], k2 = (1 << (shift2 - 1)) + 128 * kMfmaIdct.colsum[n];
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const int u = lane & 15;
    const unsigned char *trp = ctile + (16 * h + (u >> 2)) * (N * 2) + (16 * ((lane >> 4) & 1) + 4 * (u & 3)) * 2;
    const unsigned maxv = (1u << bit_depth) - 1u, max2 = maxv | (maxv << 16);
    unsigned char *scol = strip + (2 * wave) * (N * 2) + n * 2 + (4 * h) * SRS;
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<u32x4 *>(ctile + (q * 64 + lane) * 16) = cq[q];
    __builtin_amdgcn_wave_barrier();
    unsigned w[2][8];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32x2 v = lds_read_tr16(trp + g * (N * N * 2) + t * (4 * N * 2));
            w[g][2 * t] = v.x; w[g][2 * t + 1] = v.y;
        }
    __syncthreads();                                             // every wave has its coefficients in registers: the strip may overwrite the tiles
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const v4i ahi = { hi_bytes(w[g][0], w[g][1]), hi_bytes(w[g][2], w[g][3]), hi_bytes(w[g][4], w[g][5]), hi_bytes(w[g][6], w[g][7]) };
        const v4i alo = { lo_bytes(w[g][0], w[g][1]), lo_bytes(w[g][2], w[g][3]), lo_bytes(w[g][4], w[g][5]), lo_bytes(w[g][6], w[g][7]) };
        v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, b1, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) d[r] = (d[r] << 8) + k1;
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, b1, d, 0, 0, 0);
        unsigned pk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pk[q] = sat_pack_i16(d[2 * q] >> 7, d[2 * q + 1] >> 7);
        const v4i phi = { hi_bytes(pk[0], pk[1]), hi_bytes(pk[2], pk[3]), hi_bytes(pk[4], pk[5]), hi_bytes(pk[6], pk[7]) };
        const v4i plo = { lo_bytes(pk[0], pk[1]), lo_bytes(pk[2], pk[3]), lo_bytes(pk[4], pk[5]), lo_bytes(pk[6], pk[7]) };
        v16i e = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, a2, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) e[r] = (e[r] << 8) + k2;
        e = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, a2, e, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const unsigned v = sat_pack_i16(e[r] >> shift2, e[r + 1] >> shift2);
            unsigned char *p0 = scol + g * (N * 2) + ((r & 3) + 8 * (r >> 2)) * SRS;
            *reinterpret_cast<unsigned short *>(p0) = (unsigned short)(v & 0xffffu);
            *reinterpret_cast<unsigned short *>(p0 + SRS) = (unsigned short)(v >> 16);
        }
    }
    __syncthreads();                                             // the strip is complete
#pragma unroll
    for (int k = 0; k < ITER; k++) {
        const int rr = r0 + k * RPI;
        const u32x4 *rp = reinterpret_cast<const u32x4 *>(strip + rr * SRS + c * CH_PX * 2);
        u32x4 o;
        if constexpr (PXB == 1) {
            const u32x4 ra = rp[0], rb = rp[1];
            const unsigned res[8] = { ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w };
            const unsigned pd[4] = { pr[k].x, pr[k].y, pr[k].z, pr[k].w };
            unsigned od[4];
#pragma unroll
            for (int d4 = 0; d4 < 4; d4++) {
                const unsigned u01 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c010c00u), u23 = __builtin_amdgcn_perm(0u, pd[d4], 0x0c030c02u);
                const unsigned a01 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4]), bitcast<s16x2>(u01)));
                const unsigned a23 = bitcast<unsigned>(__builtin_elementwise_add_sat(bitcast<s16x2>(res[2 * d4 + 1]), bitcast<s16x2>(u23)));
                od[d4] = sat_pack_u8_i16(a01) | (sat_pack_u8_i16(a23) << 16);
            }
            o = u32x4{ od[0], od[1], od[2], od[3] };
        } else {
            const u32x4 ra = rp[0];
            o = u32x4{ add_clamp_upx2(ra.x, pr[k].x, max2), add_clamp_upx2(ra.y, pr[k].y, max2),
                       add_clamp_upx2(ra.z, pr[k].z, max2), add_clamp_upx2(ra.w, pr[k].w, max2) };
        }
        *reinterpret_cast<u32x4 *>(obase + (size_t)rr * ostride) = o;
    }
    (void)ntiles;
}

#ifdef OHEVC_LAB      // traffic-only / probe / ablation / software-pipelined dot2 kernels: bottleneck analysis, never shipped
// Bottleneck analysis only (NOT a transform): the memory traffic of one tile per workgroup - coefficients, prediction rows, result rows at
// the addresses the job records name - and nothing else: no LDS, no loop, 8 workgroups per CU.  What this reaches under bench.py is the
// ceiling of the batch's own buffers and addressing for ANY residual kernel.
template <typename Pixel>
__global__ __launch_bounds__(256) void tu_tile_traffic_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int ntiles,
                                                              const int16_t *__restrict__ coeffs, int bit_depth)
{
    constexpr int N = 32, PXB = (int)sizeof(Pixel), CH_PX = 16 / PXB, CPR = 256 / CH_PX, RPI = 256 / CPR, ITER = N / RPI;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, tile = blockIdx.x;
    const int c = tid % CPR, r0 = tid / CPR, bsel = (c * CH_PX) / N, pxoff = (c * CH_PX) % N;
    const u32x4 *jobv = reinterpret_cast<const u32x4 *>(jobs) + (size_t)tile * 8;
    const u32x4 j0 = jobv[2 * wave], j1 = jobv[2 * wave + 1], je = jobv[bsel];
    const u32x4 *s0 = reinterpret_cast<const u32x4 *>(coeffs + j0.z), *s1 = reinterpret_cast<const u32x4 *>(coeffs + j1.z);
    const u32x4 c0 = __builtin_nontemporal_load(s0 + lane), c1 = __builtin_nontemporal_load(s0 + 64 + lane);
    const u32x4 c2 = __builtin_nontemporal_load(s1 + lane), c3 = __builtin_nontemporal_load(s1 + 64 + lane);
    const unsigned opl = je.y & 0xff;
    const int ostride = PLANE_STRIDE3(planes, opl);
    unsigned char *obase = PLANE_PTR3(planes, opl) + (size_t)(je.x >> 16) * ostride + (size_t)((je.x & 0xffff) + pxoff) * PXB;
    u32x4 pr[ITER];
#pragma unroll
    for (int k = 0; k < ITER; k++) pr[k] = *reinterpret_cast<const u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride);
#pragma unroll
    for (int k = 0; k < ITER; k++) {
        pr[k].x += c0.x ^ c2.y; pr[k].y ^= c1.y + c3.z; pr[k].z += c2.z ^ c0.w; pr[k].w ^= c3.w + c1.x;
        *reinterpret_cast<u32x4 *>(obase + (size_t)(r0 + k * RPI) * ostride) = pr[k];
    }
    (void)bit_depth;
}

// raw ds_read_b64_tr_b16 of a 2048-byte LDS image filled with int16 i at index i, every lane at its own byte address: what
// tools/probe_mfma_layout.py checks the transposing read of the kernel above against
__global__ __launch_bounds__(64) void lds_tr16_probe_kernel(const int *__restrict__ addr, u32x2 *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned char img[2048];
    for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<short *>(img)[i] = (short)i;
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = lds_read_tr16(img + addr[blockIdx.x * 64 + threadIdx.x]);
}

// one v_mfma_i32_32x32x32_i8 per probe on raw lane data (a, b: 64 lanes x 16 bytes; d: 64 lanes x 16 int32): lets a test recover
// the fragment layout the kernel above relies on (tools/probe_mfma_layout.py)
__global__ __launch_bounds__(64) void mfma_i8_probe_kernel(const v4i *__restrict__ a, const v4i *__restrict__ b, v16i *__restrict__ d)
{
    const v16i zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const int i = blockIdx.x * 64 + threadIdx.x;
    d[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[i], zero, 0, 0, 0);
}

// ------------------------------------------------------------------ ablations (NOT bit-exact; bottleneck analysis only)
// MODE 0: same loads/stores as tu_idct_add_kernel but no LDS and no transform: lane i adds row i of the raw
//         coefficients to prediction row i  -> the memory-pattern ceiling of the lane-per-row layout.
// MODE 1: as MODE 0 but with the LDS staging/transposes kept (transform arithmetic removed).
template <int LOG2N, typename Pixel, int MODE>
__global__ __launch_bounds__(256) void tu_ablation_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs,
                                                          int njobs, const int16_t *__restrict__ coeffs, int bit_depth)
{
    using L = TuLayout<LOG2N>;
    constexpr int N = L::N, RS = L::RS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * L::WAVE_BYTES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / N, i = lane % N;
    const int job0 = (blockIdx.x * 4 + wave) * L::BPW;
    if (job0 >= njobs) return;
    const bool valid = job0 + g < njobs;
    const u32x4 jraw = reinterpret_cast<const u32x4 *>(jobs)[valid ? job0 + g : njobs - 1];
    const int jx = jraw.x & 0xffff, jy = jraw.x >> 16, jplane = jraw.y & 0xff;
    unsigned char *row = PLANE_PTR3(planes, jplane) + (size_t)(jy + i) * PLANE_STRIDE3(planes, jplane) + (size_t)jx * sizeof(Pixel);
    int t[N];
    if constexpr (MODE == 0) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + jraw.z + i * N);
#pragma unroll
        for (int q = 0; q < N / 8; q++) {
            const u32x4 v = src[q];
            const unsigned d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) { t[8 * q + 2 * k] = (int)(short)(d[k] & 0xffffu) >> 4; t[8 * q + 2 * k + 1] = (int)d[k] >> 20; }
        }
    } else {
        unsigned char *blk = lds + wave * L::WAVE_BYTES + g * L::BLK;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + jraw.z);
#pragma unroll
        for (int q = 0; q < N / 8; q++) {
            const int c = q * N + i;
            *reinterpret_cast<u32x4 *>(blk + (c / (N / 8)) * RS + (c % (N / 8)) * 16) = src[c];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned p[N / 2];
        constexpr PairTab<N> pt{};
        const unsigned short *col = reinterpret_cast<const unsigned short *>(blk) + i;
#pragma unroll
        for (int m = 0; m < N / 2; m++)
            p[m] = (unsigned)col[pt.lo[m] * (RS / 2)] | ((unsigned)col[pt.hi[m] * (RS / 2)] << 16);
        __builtin_amdgcn_wave_barrier();
        unsigned short *dst = reinterpret_cast<unsigned short *>(blk) + slot_of_rt(N, i);
#pragma unroll
        for (int r = 0; r < N; r += 2) {
            dst[r * (RS / 2)]       = (unsigned short)(p[r / 2] & 0xffffu);
            dst[(r + 1) * (RS / 2)] = (unsigned short)(p[r / 2] >> 16);
        }
        __builtin_amdgcn_wave_barrier();
        const u32x4 *rowp = reinterpret_cast<const u32x4 *>(blk + i * RS);
#pragma unroll
        for (int q = 0; q < N / 8; q++) {
            const u32x4 v = rowp[q];
            const unsigned d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) { t[8 * q + 2 * k] = (int)(short)(d[k] & 0xffffu) >> 4; t[8 * q + 2 * k + 1] = (int)d[k] >> 20; }
        }
    }
    add_row_store<N, Pixel>(row, t, bit_depth, valid);
}

// ------------------------------------------------------------------ persistent, software-pipelined form
// Same arithmetic as tu_idct_add_kernel.  Each wavefront walks the job list with a grid-sized stride and keeps
// the NEXT iteration's coefficients (16 bytes x N/8 per lane) and the job record after that in flight in
// registers while it transforms the current blocks, so every resident wave always has an HBM request pending
// (memory-level parallelism no longer depends on how many co-resident waves happen to be in their load phase).
template <int LOG2N, typename Pixel, int VARIANT>
__global__ __launch_bounds__(256, (VARIANT & 8) ? 5 : 1) void tu_idct_add_pipe_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs,
                                                               int njobs, const int16_t *__restrict__ coeffs, int bit_depth)
{
    using L = TuLayout<LOG2N>;
    constexpr int N = L::N, RS = L::RS, NQ = N / 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * L::WAVE_BYTES];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / N, i = lane % N;
    const int step = gridDim.x * 4 * L::BPW;
    int job0 = (blockIdx.x * 4 + wave) * L::BPW;
    if (job0 >= njobs) return;
    unsigned char *blk = lds + wave * L::WAVE_BYTES + g * L::BLK;
    const u32x4 *jobv = reinterpret_cast<const u32x4 *>(jobs);
    constexpr PairTab<N> pt{};
    const int shift2 = 20 - bit_depth;

    u32x4 jraw = jobv[job0 + g < njobs ? job0 + g : njobs - 1];
    u32x4 cur[NQ];
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + jraw.z);
#pragma unroll
        for (int q = 0; q < NQ; q++) cur[q] = src[q * N + i];
    }
    int job0n = job0 + step;
    u32x4 jraw_n = jraw;
    if (job0n < njobs) jraw_n = jobv[job0n + g < njobs ? job0n + g : njobs - 1];

    for (;;) {
        const bool has_next = job0n < njobs;                   // wave-uniform
        u32x4 nxt[NQ];
        u32x4 jraw_nn = jraw_n;
        if (has_next) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + jraw_n.z);
#pragma unroll
            for (int q = 0; q < NQ; q++) nxt[q] = src[q * N + i];
            const int job0nn = job0n + step;
            if (job0nn < njobs) jraw_nn = jobv[job0nn + g < njobs ? job0nn + g : njobs - 1];
        } else {
#pragma unroll
            for (int q = 0; q < NQ; q++) nxt[q] = cur[q];
        }
        const bool valid = job0 + g < njobs;
        const int jx = jraw.x & 0xffff, jy = jraw.x >> 16, jplane = jraw.y & 0xff;
        unsigned char *row = PLANE_PTR3(planes, jplane) + (size_t)(jy + i) * PLANE_STRIDE3(planes, jplane) + (size_t)jx * sizeof(Pixel);
        unsigned px[N * (int)sizeof(Pixel) / 4];
        if constexpr (VARIANT & 1) load_row<N, Pixel>(row, px, valid);

#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int c = q * N + i;
            *reinterpret_cast<u32x4 *>(blk + (c / NQ) * RS + (c % NQ) * 16) = cur[q];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned p[N / 2];
        {
            const unsigned short *col = reinterpret_cast<const unsigned short *>(blk) + i;
#pragma unroll
            for (int m = 0; m < N / 2; m++)
                p[m] = (unsigned)col[pt.lo[m] * (RS / 2)] | ((unsigned)col[pt.hi[m] * (RS / 2)] << 16);
        }
        __builtin_amdgcn_wave_barrier();
        int t[N];
        Idct1D<N, false>::run(p, t, 64);
        {
            unsigned short *dst = reinterpret_cast<unsigned short *>(blk) + slot_of_rt(N, i);
#pragma unroll
            for (int r = 0; r < N; r += 2) {
                const unsigned pk = sat_pack_i16(t[r] >> 7, t[r + 1] >> 7);
                dst[r * (RS / 2)]       = (unsigned short)(pk & 0xffffu);
                dst[(r + 1) * (RS / 2)] = (unsigned short)(pk >> 16);
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            const u32x4 *rowp = reinterpret_cast<const u32x4 *>(blk + i * RS);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const u32x4 v = rowp[q];
                p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        Idct1D<N, false>::run(p, t, 1 << (shift2 - 1));
#pragma unroll
        for (int k = 0; k < N; k++) t[k] >>= shift2;
        if constexpr (!(VARIANT & 1)) load_row<N, Pixel>(row, px, valid);
        finish_row<N, Pixel>(row, px, t, bit_depth, valid);

        if (!has_next) break;
        job0 = job0n; job0n += step;
        jraw = jraw_n; jraw_n = jraw_nn;
#pragma unroll
        for (int q = 0; q < NQ; q++) cur[q] = nxt[q];
    }
}

#endif  // OHEVC_LAB

// ------------------------------------------------------------------ 4x4 IDCT / DST: one lane per block
template <typename Pixel, bool DST>
__device__ __forceinline__ void tu_4x4_body(int wg, const PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int njobs,
                                            const int16_t *__restrict__ coeffs, int bit_depth)
{
    const int job = wg * 256 + threadIdx.x;
    if (job >= njobs) return;
    const u32x4 jraw = reinterpret_cast<const u32x4 *>(jobs)[job];
    const int jx = jraw.x & 0xffff, jy = jraw.x >> 16, jplane = jraw.y & 0xff;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(coeffs + jraw.z);
    const u32x4 a = src[0], b = src[1];
    const unsigned raw[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    int c[4][4];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c[k / 2][(k % 2) * 2]     = (int)(short)(raw[k] & 0xffffu);
        c[k / 2][(k % 2) * 2 + 1] = (int)raw[k] >> 16;
    }
    auto tr = [](int x0, int x1, int x2, int x3, int add, int &y0, int &y1, int &y2, int &y3) {
        if constexpr (DST) {        // inverse DST-VII, hevcdsp_template.c:170-203
            y0 = 29 * x0 + 74 * x1 + 84 * x2 + 55 * x3 + add;
            y1 = 55 * x0 + 74 * x1 - 29 * x2 - 84 * x3 + add;
            y2 = 74 * (x0 - x2 + x3) + add;
            y3 = 84 * x0 - 74 * x1 + 55 * x2 - 29 * x3 + add;
        } else {                    // TR_4, hevcdsp_template.c:210-222
            const int e0 = 64 * (x0 + x2) + add, e1 = 64 * (x0 - x2) + add;
            const int o0 = 83 * x1 + 36 * x3, o1 = 36 * x1 - 83 * x3;
            y0 = e0 + o0; y1 = e1 + o1; y2 = e1 - o1; y3 = e0 - o0;
        }
    };
    auto clip16 = [](int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; };
#pragma unroll
    for (int i = 0; i < 4; i++) {   // columns
        int y0, y1, y2, y3;
        tr(c[0][i], c[1][i], c[2][i], c[3][i], 64, y0, y1, y2, y3);
        c[0][i] = clip16(y0 >> 7); c[1][i] = clip16(y1 >> 7); c[2][i] = clip16(y2 >> 7); c[3][i] = clip16(y3 >> 7);
    }
    const int shift2 = 20 - bit_depth;
    unsigned char *base = PLANE_PTR3(planes, jplane) + (size_t)jy * PLANE_STRIDE3(planes, jplane) + (size_t)jx * sizeof(Pixel);
    const int stride = PLANE_STRIDE3(planes, jplane);
#pragma unroll
    for (int r = 0; r < 4; r++) {   // rows
        int res[4];
        tr(c[r][0], c[r][1], c[r][2], c[r][3], 1 << (shift2 - 1), res[0], res[1], res[2], res[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) res[k] >>= shift2;
        add_row_store<4, Pixel>(base + (size_t)r * stride, res, bit_depth, true);
    }
}

template <typename Pixel, bool DST>
__global__ __launch_bounds__(256) void tu_4x4_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int njobs,
                                                     const int16_t *__restrict__ coeffs, int bit_depth)
{
    tu_4x4_body<Pixel, DST>(blockIdx.x, planes, jobs, njobs, coeffs, bit_depth);
}

// ------------------------------------------------------------------ DC-only / transform-skip / bypass (+rdpcm): one lane per row
// idct_dc :303-316, transform_skip :139-163, transform_rdpcm :114-136 (int16 wrap-around of the in-place reference
// is reproduced by truncating to int16 after the modular prefix sum).
// residual row r of the block described by job record `jraw` (kind != OHEVC_TU_PCM: those samples replace the block)
template <int LOG2N>
__device__ __forceinline__ void tu_rows_residual(const u32x4 jraw, const int r, const int16_t *__restrict__ coeffs, const int bit_depth, const int kind, int *res)
{
    constexpr int N = 1 << LOG2N;
    if (kind == OHEVC_TU_DC) {
        const int dc = (int)jraw.y >> 16, shift = 14 - bit_depth, add = shift > 0 ? 1 << (shift - 1) : 0;       // BIT_DEPTH 14: see tu_generic.hpp
        const int v = (((dc + 1) >> 1) + add) >> shift;
#pragma unroll
        for (int x = 0; x < N; x++) res[x] = v;
    } else {
        const int16_t *blk = coeffs + jraw.z;
        const bool skip = kind == OHEVC_TU_SKIP || kind == OHEVC_TU_SKIP_RDPCM_H || kind == OHEVC_TU_SKIP_RDPCM_V;
        const bool vert = kind == OHEVC_TU_SKIP_RDPCM_V || kind == OHEVC_TU_BYPASS_RDPCM_V;
        const bool horz = kind == OHEVC_TU_SKIP_RDPCM_H || kind == OHEVC_TU_BYPASS_RDPCM_H;
        const int shift = 15 - bit_depth - LOG2N;
#pragma unroll
        for (int x = 0; x < N; x++) res[x] = 0;
        const int first = vert ? 0 : r;                    // vertical rdpcm: sum rows 0..r
        for (int yy = first; yy <= r; yy++) {
            const u32x2 *rowp = reinterpret_cast<const u32x2 *>(blk + yy * N);
#pragma unroll
            for (int q = 0; q < N / 4; q++) {
                const u32x2 v = rowp[q];
                int e[4] = { (int)(short)(v.x & 0xffffu), (int)v.x >> 16, (int)(short)(v.y & 0xffffu), (int)v.y >> 16 };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int s = e[k];
                    if (skip) {
                        if (shift > 0) s = (s + (1 << (shift - 1))) >> shift;
                        else           s = (int)(short)(s << -shift);
                    }
                    res[4 * q + k] += s;
                }
            }
        }
        if (horz) {
#pragma unroll
            for (int x = 1; x < N; x++) res[x] += res[x - 1];
        }
#pragma unroll
        for (int x = 0; x < N; x++) res[x] = (int)(short)res[x];
    }
}

// a row of packed pixels to the plane (the tail of finish_row)
template <int N, typename Pixel>
__device__ __forceinline__ void store_row(unsigned char *row, const unsigned *px)
{
    constexpr int ROWDW = N * (int)sizeof(Pixel) / 4;
    constexpr int VEC   = ROWDW >= 4 ? 4 : ROWDW;
#pragma unroll
    for (int v = 0; v < ROWDW / VEC; v++) {
        if constexpr (VEC == 4) {
            u32x4 t = { px[4 * v], px[4 * v + 1], px[4 * v + 2], px[4 * v + 3] };
            *reinterpret_cast<u32x4 *>(row + 16 * v) = t;
        } else if constexpr (VEC == 2) {
            u32x2 t = { px[2 * v], px[2 * v + 1] };
            *reinterpret_cast<u32x2 *>(row + 8 * v) = t;
        } else {
            *reinterpret_cast<unsigned *>(row + 4 * v) = px[v];
        }
    }
}

template <int LOG2N, typename Pixel>
__device__ __forceinline__ void tu_rows_body(int wg, const PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int njobs,
                                             const int16_t *__restrict__ coeffs, int bit_depth, int kind)
{
    constexpr int N = 1 << LOG2N;
    const int tid = wg * 256 + threadIdx.x;
    const int job = tid >> LOG2N, r = tid & (N - 1);
    if (job >= njobs) return;
    const u32x4 jraw = reinterpret_cast<const u32x4 *>(jobs)[job];
    const int jx = jraw.x & 0xffff, jy = jraw.x >> 16, jplane = jraw.y & 0xff;
    int res[N];
    tu_rows_residual<LOG2N>(jraw, r, coeffs, bit_depth, kind, res);
    unsigned char *row = PLANE_PTR3(planes, jplane) + (size_t)(jy + r) * PLANE_STRIDE3(planes, jplane) + (size_t)jx * sizeof(Pixel);
    if (kind == OHEVC_TU_PCM) {                      // put_pcm: the samples replace the block (clip(0 + sample) == sample)
        unsigned px[N * (int)sizeof(Pixel) / 4];
        load_row<N, Pixel>(row, px, false);
        finish_row<N, Pixel>(row, px, res, bit_depth, true);
    } else {
        add_row_store<N, Pixel>(row, res, bit_depth, true);
    }
}

template <int LOG2N, typename Pixel>
__global__ __launch_bounds__(256) void tu_rows_kernel(PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int njobs,
                                                      const int16_t *__restrict__ coeffs, int bit_depth, int kind)
{
    tu_rows_body<LOG2N, Pixel>(blockIdx.x, planes, jobs, njobs, coeffs, bit_depth, kind);
}


// ------------------------------------------------------------------ cross-component prediction (RExt, 4:4:4)
// hls_transform_unit (hevc.c:1291-1365) + the tail of ff_hevc_hls_residual_coding (hevc_cabac.c:1942-1949): with
// cross_component_prediction the residual added to a chroma block is
//     (int16)(rC + ((res_scale_val * rY) >> 3))
// where rY is the luma residual of the same transform unit AFTER its inverse transform and rC the chroma block's own
// residual (0 when the block has no coded coefficients).  Behind recording tables the luma residual never exists on the
// host, so such a chroma block carries both coefficient blocks and both residual kinds (OHEVC_TU_CROSS, ohevc_hip.h) and
// this body derives the two residuals itself.  A rare tool: generic code, one wavefront per block, two blocks per
// workgroup (each uses half of the workgroup's LDS), any residual kind in direct matrix form -- exact integer arithmetic
// identical to the dedicated kernels (no partial sum can overflow int32: 32 terms of at most 2^15 * 90).
template <typename Pixel>
__device__ __forceinline__ void tu_cross_body(unsigned char *lds, int wg, const PlaneSet planes, const ohevc_tu_job *__restrict__ jobs, int njobs,
                                              int log2, const int16_t *__restrict__ coeffs, int bit_depth)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ji = wg * 2 + wave;
    if (wave >= 2 || ji >= njobs) return;                   // wave-uniform; only wave-level synchronisation below
    static_assert(4 * TuLayout<5>::WAVE_BYTES >= 2 * 3 * 32 * 32 * 2, "two blocks x three int16 tiles must fit the workgroup's LDS");
    short *tile = reinterpret_cast<short *>(lds + wave * (2 * TuLayout<5>::WAVE_BYTES));
    short *tmp = tile, *ry = tile + 1024, *rc = tile + 2048;
    const ohevc_tu_job jb = jobs[ji];
    const int N = 1 << log2, NN = N * N;
    const int kind_c = jb.reserved0 & 15, kind_y = jb.reserved0 >> 4, scale = jb.dc;
    residual_generic(kind_y, log2, coeffs + jb.reserved1, bit_depth, tmp, ry, lane);
    if (kind_c != 15) residual_generic(kind_c, log2, coeffs + jb.coeff_off, bit_depth, tmp, rc, lane);
    unsigned char *base = PLANE_PTR3(planes, jb.plane) + (size_t)jb.y * PLANE_STRIDE3(planes, jb.plane) + (size_t)jb.x * sizeof(Pixel);
    const int stride = PLANE_STRIDE3(planes, jb.plane), maxv = (1 << bit_depth) - 1;
    for (int o = lane; o < NN; o += 64) {
        const int own = kind_c != 15 ? (int)rc[o] : 0;
        const int res = (int)(short)(own + ((scale * (int)ry[o]) >> 3));            // hevc_cabac.c:1946 / hevc.c:1326 (int16 store)
        Pixel *px = reinterpret_cast<Pixel *>(base + (size_t)(o >> log2) * stride) + (o & (N - 1));
        const int v = (int)*px + res;                                                // transform_add, hevcdsp_template.c:45-111
        *px = (Pixel)(v < 0 ? 0 : v > maxv ? maxv : v);
    }
}

// one workgroup of a (size, kind) bin: the dispatch shared by the segmented launch and the level executor
template <typename Pixel>
__device__ __forceinline__ void tu_dispatch(unsigned char *lds, int wg, const PlaneSet planes, const ohevc_tu_job *__restrict__ j, int n, int log2, int kind,
                                            const int16_t *__restrict__ coeffs, int bit_depth)
{
    if (kind == OHEVC_TU_IDCT && log2 == 5)      tu_idct_add_body<5, Pixel, 16 + 128>(lds, wg, planes, j, n, coeffs, bit_depth);
    else if (kind == OHEVC_TU_IDCT && log2 == 4) tu_idct_add_body<4, Pixel, 16 + 128>(lds, wg, planes, j, n, coeffs, bit_depth);
    else if (kind == OHEVC_TU_IDCT && log2 == 3) tu_idct_add_body<3, Pixel, 1>(lds, wg, planes, j, n, coeffs, bit_depth);
    else if (kind == OHEVC_TU_IDCT)              tu_4x4_body<Pixel, false>(wg, planes, j, n, coeffs, bit_depth);
    else if (kind == OHEVC_TU_DST4)              tu_4x4_body<Pixel, true>(wg, planes, j, n, coeffs, bit_depth);
    else if (kind == OHEVC_TU_CROSS)             tu_cross_body<Pixel>(lds, wg, planes, j, n, log2, coeffs, bit_depth);
    else if (log2 == 2)                          tu_rows_body<2, Pixel>(wg, planes, j, n, coeffs, bit_depth, kind);
    else if (log2 == 3)                          tu_rows_body<3, Pixel>(wg, planes, j, n, coeffs, bit_depth, kind);
    else if (log2 == 4)                          tu_rows_body<4, Pixel>(wg, planes, j, n, coeffs, bit_depth, kind);
    else                                         tu_rows_body<5, Pixel>(wg, planes, j, n, coeffs, bit_depth, kind);
}

// ------------------------------------------------------------------ one launch for a mix of sizes and kinds
// A frame's residuals come in up to 4 sizes x 10 kinds.  Launching each (size, kind) bin separately costs a kernel
// boundary per bin (x every intra dependency level); instead the host passes a small segment table and every workgroup
// looks up which bin it serves.  All lanes of a workgroup run the same body, so nothing diverges.
constexpr int TU_MAX_SEGMENTS = 40;
struct TuSegTable {
    int nsegs;
    int first_wg[TU_MAX_SEGMENTS + 1];       // prefix sum of workgroups per segment
    int first_job[TU_MAX_SEGMENTS];
    int njobs[TU_MAX_SEGMENTS];
    unsigned char log2[TU_MAX_SEGMENTS], kind[TU_MAX_SEGMENTS];
};

template <typename Pixel>
__global__ __launch_bounds__(256) void tu_multi_kernel(PlaneSet planes, TuSegTable tab, const ohevc_tu_job *__restrict__ jobs,
                                                       const int16_t *__restrict__ coeffs, int bit_depth)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TuLayout<5>::WAVE_BYTES];
    int s = 0;
    while (s + 1 < tab.nsegs && (int)blockIdx.x >= tab.first_wg[s + 1]) s++;      // wave-uniform scan, <= 40 entries
    const int wg = blockIdx.x - tab.first_wg[s], log2 = tab.log2[s], kind = tab.kind[s], n = tab.njobs[s];
    tu_dispatch<Pixel>(lds, wg, planes, jobs + tab.first_job[s], n, log2, kind, coeffs, bit_depth);
}

// ------------------------------------------------------------------ intra dependency levels in ONE launch
// Intra pictures chain ~150 dependency levels per 1080p picture (prediction of level L reads what level L-1
// reconstructed), each only tens of blocks wide.  Launched level by level that is ~300 kernel boundaries per picture,
// and kernel boundaries (command-processor dispatch + device-wide cache maintenance) are a GPU-wide serial resource:
// pictures of different decoding threads cannot overlap.  Here the whole chain runs inside one kernel:
//   * the work is a list of PHASES (intra jobs of a level, or one (size, kind) residual bin of a level), each a run of
//     virtual workgroups; phases are grouped into STEPS (step 2L = prediction of level L, step 2L+1 = its residuals);
//   * persistent workgroups draw virtual workgroup numbers from a ticket counter, so numbers are handed out in order
//     and a holder of number i only ever waits for numbers < i, which are held by workgroups that are already running:
//     forward progress needs no assumption about residency or dispatch order;
//   * before running a virtual workgroup of step s, a workgroup waits until all of step s-1 has signalled;
//   * ONE XCD does the whole chain: the workgroup with blockIdx == leader publishes its XCC_ID, workgroups on other
//     XCDs leave at once.  All participants then share one L2, so handing samples from step to step needs no L2
//     write-back / invalidate (which costs microseconds per workgroup per step when done device-wide, measured):
//     stores are complete in L2 after s_waitcnt vmcnt(0) (the vector L1 is write-through), readers only drop their
//     L1 (buffer_inv).  The kernel boundary publishes the result to the other XCDs as usual.  Chains of different
//     pictures (decoding threads) pick different leaders and so run on different XCDs side by side.
struct LevelPhase {                  // mirrors ohevc_level_phase (include/ohevc_hip.h)
    int first_wg, step, type, first_job, njobs, log2_size, kind, reserved;
};
enum { LV_HOME = 0, LV_TICKET = 1, LV_DONE = 2 };       // layout of the sync words

template <typename Pixel>
__global__ __launch_bounds__(256) void levels_kernel(PlaneSet planes, const LevelPhase *__restrict__ phases, int nphases, int total_wgs,
                                                     unsigned *sync, const unsigned *__restrict__ need, int leader,
                                                     const ohevc_intra_job *__restrict__ intra_jobs, const ohevc_intra_cip *__restrict__ cips,
                                                     const ohevc_tu_job *__restrict__ tu_jobs, const int16_t *__restrict__ coeffs, int bit_depth)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TuLayout<5>::WAVE_BYTES];
    __shared__ IntraShared ish[4];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // in an SGPR: the compiler must SEE that it is uniform
    // Every branch around the barriers below is WAVE-uniform on purpose: a lane-divergent `if (tid == 0)` inside this loop
    // lets the compiler park lane 0 while the rest of its wavefront runs ahead to the barrier (observed: the ticket was
    // never refreshed).  Single-lane effects are expressed through the operand instead (add 1 in lane 0, 0 elsewhere).
    const unsigned my_home = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) + 1u;       // HW_REG_XCC_ID + 1
    if ((int)blockIdx.x == leader) {
        if (wave == 0) atomicMax(&sync[LV_HOME], my_home);                                   // was 0
    } else {
        unsigned home;
        while ((home = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sync[LV_HOME], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) == 0u)
            __builtin_amdgcn_s_sleep(1);
        if (home != my_home) return;
    }
    for (;;) {
        if (wave == 0) {
            const unsigned got = atomicAdd(&sync[LV_TICKET], lane == 0 ? 1u : 0u);
            s_ticket = __builtin_amdgcn_readfirstlane((int)got);          // lane 0's return value = the ticket
        }
        __syncthreads();
        const int vwg = s_ticket;
        if (vwg >= total_wgs) return;
        int lo = 0, hi = nphases - 1;                          // last phase with first_wg <= vwg (workgroup-uniform)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (phases[mid].first_wg <= vwg) lo = mid; else hi = mid - 1;
        }
        const LevelPhase ph = phases[lo];
        if (ph.step > 0) {                                     // every wavefront polls for itself (same address: one request)
            const unsigned want = need[ph.step - 1];
            while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sync[LV_DONE + ph.step - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (int)want)
                __builtin_amdgcn_s_sleep(1);
            xcd_acquire();                                     // acquire inside the XCD: forget what this CU's L1 holds
        }
        const int local = vwg - ph.first_wg;
        if (ph.type == 0) {                                    // intra prediction: one wavefront per block, four per workgroup
            const int ji = local * 4 + wave;
            if (ji < ph.njobs) intra_body<Pixel, true>(ish[wave], lane, planes, intra_jobs[ph.first_job + ji], bit_depth, cips);
        } else {
            tu_dispatch<Pixel>(lds, local, planes, tu_jobs + ph.first_job, ph.njobs, ph.log2_size, ph.kind, coeffs, bit_depth);
        }
        xcd_release();                                         // release inside the XCD: this wavefront's stores sit in the shared L2 ...
        __syncthreads();                                       // (also: everyone has read s_ticket before wave 0 rewrites it)
        if (wave == 0) atomicAdd(&sync[LV_DONE + ph.step], lane == 0 ? 1u : 0u);   // ... before the step counter says so
    }
}

// ------------------------------------------------------------------ launcher
int g_tu_variant = -1;    // set through ohevc_debug_set_tu_variant(); -1 = shipped configuration (see launch_idct)
int g_tu_pipe_wgs = 2048; // workgroups of the persistent lab forms (ohevc_debug_set_tu_pipe_workgroups)

#ifdef OHEVC_LAB
// the A/B forms of the lab build (include/ohevc_debug.h lists the bits); returns false when `variant` names none of them
template <int LOG2N, typename Pixel>
static bool launch_idct_lab(int variant, int grid, hipStream_t st, const PlaneSet &ps, const ohevc_tu_job *jobs, int njobs, const int16_t *coeffs, int bit_depth)
{
    if constexpr (LOG2N == 5) {
        if ((variant & 2048) && (variant & (4096 | 8192 | 16384 | 32768))) {      // loop forms of the tile kernel, its ablation, the traffic-only kernel
            const int ntiles = njobs / 8, rest = njobs - ntiles * 8;
            if (ntiles) {
                const int pgrid = ntiles < g_tu_pipe_wgs ? ntiles : g_tu_pipe_wgs;
                if (variant & 32768)      hipLaunchKernelGGL((tu_tile_traffic_kernel<Pixel>), dim3(ntiles), dim3(256), 0, st, ps, jobs, ntiles, coeffs, bit_depth);
                else if (variant & 16384) hipLaunchKernelGGL((tu_idct32_tile2_kernel<Pixel>), dim3(pgrid), dim3(256), 0, st, ps, jobs, ntiles, coeffs, bit_depth);
                else if (variant & 8192)  hipLaunchKernelGGL((tu_idct32_tile_kernel<Pixel, 3, true>), dim3(pgrid), dim3(256), 0, st, ps, jobs, ntiles, coeffs, bit_depth);
                else                      hipLaunchKernelGGL((tu_idct32_tile_kernel<Pixel, 3>), dim3(pgrid), dim3(256), 0, st, ps, jobs, ntiles, coeffs, bit_depth);
            }
            if (rest) hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 16 + 128>), dim3(1), dim3(256), 0, st, ps, jobs + ntiles * 8, rest, coeffs, bit_depth);
            return true;
        }
        if (variant & 256) {     // first matrix-core form; every wave loops over block pairs
            const int pairs = (njobs + 1) / 2, need = (pairs + 3) / 4, pgrid = need < g_tu_pipe_wgs ? need : g_tu_pipe_wgs;
            switch ((variant >> 9) & 3) {      // bit 9: prediction rows fetched early; bit 10: no register prefetch
            case 0: hipLaunchKernelGGL((tu_idct32_mfma_kernel<Pixel, 0>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
            case 1: hipLaunchKernelGGL((tu_idct32_mfma_kernel<Pixel, 1>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
            case 2: hipLaunchKernelGGL((tu_idct32_mfma_kernel<Pixel, 2>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
            case 3: hipLaunchKernelGGL((tu_idct32_mfma_kernel<Pixel, 3>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
            }
            return true;
        }
    }
    if (variant & 2048) return false;
    if (variant & 4) {
        const int pgrid = grid < g_tu_pipe_wgs ? grid : g_tu_pipe_wgs;
        switch (variant & 9) {
        case 0: hipLaunchKernelGGL((tu_idct_add_pipe_kernel<LOG2N, Pixel, 0>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
        case 1: hipLaunchKernelGGL((tu_idct_add_pipe_kernel<LOG2N, Pixel, 1>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
        case 8: hipLaunchKernelGGL((tu_idct_add_pipe_kernel<LOG2N, Pixel, 8>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
        case 9: hipLaunchKernelGGL((tu_idct_add_pipe_kernel<LOG2N, Pixel, 9>), dim3(pgrid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth); break;
        }
        return true;
    }
    if (variant & 96) {          // ablations: 32 = no LDS / no transform, 64 = LDS traffic kept, no transform
        if (variant & 32) hipLaunchKernelGGL((tu_ablation_kernel<LOG2N, Pixel, 0>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        else              hipLaunchKernelGGL((tu_ablation_kernel<LOG2N, Pixel, 1>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        return true;
    }
    if constexpr (LOG2N >= 4) {
        if (variant & 512) {       // workgroup-wide epilogue of the dot2 form (256-sample strips); bit 10: non-temporal coefficient loads
            if (variant & 1024) hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 512 + 1024 + 128>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
            else                hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 512 + 128>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
            return true;
        }
    }
    if (!(variant & 16) && (variant & 2)) {       // coefficients straight from HBM as int16 columns
        if (variant & 1) hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 3>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        else             hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 2>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        return true;
    }
    return false;
}
#endif

template <int LOG2N, typename Pixel>
static void launch_idct(int grid, hipStream_t st, const PlaneSet &ps, const ohevc_tu_job *jobs, int njobs, const int16_t *coeffs, int bit_depth)
{
    // Shipped configuration (A/B on MI355X, profiles/r02*_ab_tu_variants.txt, r02l_bench_ab.jsonl):
    //   32x32  matrix-core tiles of 8 blocks, one per workgroup (tu_idct32_tile1_kernel); what is left of the batch (< 8 blocks) and the
    //          segmented / level launches take the dot2 form;
    //   16x16  dot2 form, wave-private coalesced epilogue, non-accumulating chain starts, non-temporal coefficient loads;
    //   8x8    dot2 form with the prediction rows requested before the transform.
    const int variant = g_tu_variant >= 0 ? g_tu_variant : (LOG2N == 5 ? 2048 + 16 + 128 + 1024 : LOG2N == 4 ? 16 + 128 + 1024 : 1);
#ifdef OHEVC_LAB
    if (launch_idct_lab<LOG2N, Pixel>(variant, grid, st, ps, jobs, njobs, coeffs, bit_depth)) return;
#endif
    if constexpr (LOG2N == 5) {
        if (variant & 2048) {
            const int ntiles = njobs / 8, rest = njobs - ntiles * 8;
            if (ntiles) hipLaunchKernelGGL((tu_idct32_tile1_kernel<Pixel, 4>), dim3(ntiles), dim3(256), 0, st, ps, jobs, ntiles, coeffs, bit_depth);
            if (rest) hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 16 + 128 + 1024>), dim3(1), dim3(256), 0, st, ps, jobs + ntiles * 8, rest, coeffs, bit_depth);
            return;
        }
    }
    if (variant & 16) {
        if (variant & 1024)     hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 16 + 128 + 1024>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        else if (variant & 128) hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 16 + 128>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        else                    hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 16>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        return;
    }
    if (variant & 1) hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 1>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
    else             hipLaunchKernelGGL((tu_idct_add_kernel<LOG2N, Pixel, 0>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
}

template <typename Pixel>
static int launch_tu(const PlaneSet &ps, int bit_depth, int log2, int kind, const ohevc_tu_job *jobs, int njobs,
                     const int16_t *coeffs, hipStream_t st)
{
    if (kind == OHEVC_TU_IDCT && log2 >= 3) {
        const int bpw = 64 >> log2, per_wg = 4 * bpw, grid = (njobs + per_wg - 1) / per_wg;
        switch (log2) {
        case 3: launch_idct<3, Pixel>(grid, st, ps, jobs, njobs, coeffs, bit_depth); break;
        case 4: launch_idct<4, Pixel>(grid, st, ps, jobs, njobs, coeffs, bit_depth); break;
        case 5: launch_idct<5, Pixel>(grid, st, ps, jobs, njobs, coeffs, bit_depth); break;
        }
    } else if (kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4) {
        const int grid = (njobs + 255) / 256;
        if (kind == OHEVC_TU_DST4)
            hipLaunchKernelGGL((tu_4x4_kernel<Pixel, true>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
        else
            hipLaunchKernelGGL((tu_4x4_kernel<Pixel, false>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth);
    } else {
        const long long threads = (long long)njobs << log2;
        const int grid = (int)((threads + 255) / 256);
        switch (log2) {
        case 2: hipLaunchKernelGGL((tu_rows_kernel<2, Pixel>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth, kind); break;
        case 3: hipLaunchKernelGGL((tu_rows_kernel<3, Pixel>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth, kind); break;
        case 4: hipLaunchKernelGGL((tu_rows_kernel<4, Pixel>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth, kind); break;
        case 5: hipLaunchKernelGGL((tu_rows_kernel<5, Pixel>), dim3(grid), dim3(256), 0, st, ps, jobs, njobs, coeffs, bit_depth, kind); break;
        }
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

}  // namespace ohevc

// ------------------------------------------------------------------ intra prediction + its residual, one wavefront per block
// In the reference a transform block is predicted and its residual added back to back (hls_transform_unit, hevc.c:1214-1215 then
// :1260-1290).  As two launches per dependency level the residual kernel waits on a kernel boundary for samples its own wavefront just
// stored: here the wavefront that predicted block b runs the residual body for b (the batched bodies, pointed at this one job).  One
// launch per level instead of two.
// Round 1 had this kernel and reverted it for "rare stale reads between its two halves".  The cause: s_waitcnt vmcnt(0) makes the
// prediction stores complete in L2, but the CU's L1 may still hold a line of the block that ANOTHER wavefront of the CU fetched
// earlier - as neighbour samples of its own block - and stores do not update it: the residual body then adds to old samples.  The
// read-back needs the same acquire as a hand-off between wavefronts (buffer_inv sc1, the CTB executor's xcd_acquire).
namespace ohevc {
template <typename Pixel>
__global__ __launch_bounds__(64) void intra_recon_kernel(PlaneSet planes, const ohevc_intra_job *__restrict__ jobs, const ohevc_tu_job *__restrict__ residuals, int njobs,
                                                         int bit_depth, const ohevc_intra_cip *__restrict__ cips, const int16_t *__restrict__ coeffs)
{
    __shared__ IntraShared ish;
    __shared__ __attribute__((aligned(16))) unsigned char tu_lds[2 * TuLayout<5>::WAVE_BYTES];
    const ohevc_intra_job jb = jobs[blockIdx.x];
    intra_body<Pixel, false>(ish, threadIdx.x, planes, jb, bit_depth, cips);
    const ohevc_tu_job *res = residuals + blockIdx.x;
    const int kind1 = __builtin_amdgcn_readfirstlane((int)res->reserved0);        // residual kind + 1; 0: this block has none
    if (kind1 == 0) return;
    xcd_release();                                              // the prediction is in L2 ...
    xcd_acquire();                                              // ... and nothing older of it in this CU's L1
    tu_dispatch<Pixel>(tu_lds, 0, planes, res, 1, jb.log2_size, kind1 - 1, coeffs, bit_depth);
}
}  // namespace ohevc

extern "C" int ohevc_dev_intra_recon_batch(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, const ohevc_tu_job *residuals, int njobs,
                                           const ohevc_intra_cip *cip, const int16_t *coeffs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(jobs != nullptr && residuals != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0 && (reinterpret_cast<uintptr_t>(residuals) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(cip) & 15) == 0 && (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0, "arrays must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bit_depth == 8) hipLaunchKernelGGL((intra_recon_kernel<uint8_t>), dim3(njobs), dim3(64), 0, st, ps, jobs, residuals, njobs, bit_depth, cip, coeffs);
    else                hipLaunchKernelGGL((intra_recon_kernel<uint16_t>), dim3(njobs), dim3(64), 0, st, ps, jobs, residuals, njobs, bit_depth, cip, coeffs);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

#include "intra_pack.hpp"       // N lanes per N x N block: prediction + residual in registers, one store (uses the residual bodies above)

extern "C" int ohevc_dev_intra_recon_sorted(const ohevc_plane planes[3], int bit_depth, const ohevc_intra_job *jobs, const ohevc_tu_job *residuals,
                                            const int32_t count_by_size[4], const int16_t *coeffs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr && count_by_size != nullptr, "null argument");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    IntraPackSegs sg = {};
    long long total = 0;
    for (int s = 0; s < 4; s++) {
        OHEVC_REQUIRE(count_by_size[s] >= 0, "negative job count");
        const int per_wave = 16 >> s;
        sg.first_job[s] = (int)total;
        sg.njobs[s] = count_by_size[s];
        sg.first_wave[s + 1] = sg.first_wave[s] + (count_by_size[s] + per_wave - 1) / per_wave;
        total += count_by_size[s];
    }
    OHEVC_REQUIRE(total < (1ll << 30), "too many jobs");
    if (total == 0) return OHEVC_OK;
    OHEVC_REQUIRE(jobs != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0 && (reinterpret_cast<uintptr_t>(residuals) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0, "arrays must be 16-byte aligned");
    OHEVC_REQUIRE(residuals == nullptr || coeffs != nullptr, "residual records without a coefficient arena");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int waves = sg.first_wave[4];
    if (residuals != nullptr) {
        if (bit_depth == 8) hipLaunchKernelGGL((intra_pack_kernel<uint8_t, true>), dim3(waves), dim3(64), 0, st, ps, jobs, residuals, sg, bit_depth, coeffs);
        else                hipLaunchKernelGGL((intra_pack_kernel<uint16_t, true>), dim3(waves), dim3(64), 0, st, ps, jobs, residuals, sg, bit_depth, coeffs);
    } else {
        if (bit_depth == 8) hipLaunchKernelGGL((intra_pack_kernel<uint8_t, false>), dim3(waves), dim3(64), 0, st, ps, jobs, residuals, sg, bit_depth, coeffs);
        else                hipLaunchKernelGGL((intra_pack_kernel<uint16_t, false>), dim3(waves), dim3(64), 0, st, ps, jobs, residuals, sg, bit_depth, coeffs);
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

// the widest level (in wavefronts of the packed kernel) ohevc_dev_intra_chain takes: since round 4 any - a level wider than the kernel's
// one workgroup costs further passes of it (the caller decides from which width on a launch of its own is cheaper)
extern "C" int ohevc_intra_chain_max_waves(void) { return 1 << 20; }
extern "C" int ohevc_intra_chain_workgroup_waves(void) { return ohevc::kChainWaves; }
extern "C" int ohevc_intra_chain_max_levels(void) { return ohevc::kChainMaxLevels; }      // levels one launch takes (their records sit in LDS)
// diagnosis: five 64-bit counters in device memory the chain kernel's wavefront 0 adds its phase clocks to (ohevc_debug.h)
static unsigned long long *g_chain_clocks = nullptr;
extern "C" int ohevc_debug_intra_chain_clocks(int on, unsigned long long out[8])
{
    if (on && !g_chain_clocks) {
        OHEVC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_chain_clocks), 64 * sizeof(unsigned long long)));
        OHEVC_HIP_TRY(hipMemset(g_chain_clocks, 0, 64 * sizeof(unsigned long long)));
    }
    if (out && g_chain_clocks) {
        OHEVC_HIP_TRY(hipDeviceSynchronize());
        OHEVC_HIP_TRY(hipMemcpy(out, g_chain_clocks, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (!on && g_chain_clocks) { (void)hipFree(g_chain_clocks); g_chain_clocks = nullptr; }
    return OHEVC_OK;
}
// The hand-over between two levels of the chain (ohevc_debug_set_chain_handover): bit 0 = the agent-scope acquire of rounds 2-3 (buffer_inv sc1
// per level), bit 1 = wait for the level's stores in front of the barrier (s_waitcnt vmcnt(0): rounds 4-5).  0 = the barrier alone.
static int g_chain_agent_acquire = 0;
extern "C" int ohevc_debug_set_chain_handover(int mode) { const int prev = g_chain_agent_acquire; g_chain_agent_acquire = mode & 7; return prev; }      // bit 2: slots from the level records (round 5), not from descriptors

extern "C" int ohevc_dev_intra_chain(const ohevc_plane planes[3], int bit_depth, const void *base, const ohevc_intra_chain_level *levels, int nlevels,
                                    const int16_t *coeffs, void *stream)
{
    using namespace ohevc;
    static_assert(sizeof(ohevc_intra_chain_level) == sizeof(IntraChainLevel) && sizeof(IntraChainLevel) == 48, "level record layout");
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(nlevels >= 0 && nlevels <= kChainMaxLevels, "nlevels (at most ohevc_intra_chain_max_levels() per launch)");
    if (nlevels == 0) return OHEVC_OK;
    OHEVC_REQUIRE(base != nullptr && levels != nullptr && (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (reinterpret_cast<uintptr_t>(levels) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0, "arrays must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const IntraChainLevel *lv = reinterpret_cast<const IntraChainLevel *>(levels);
    // the transforms of the run's blocks first, over the whole GPU, in place (the arena is the caller's device buffer: written here)
    if (coeffs != nullptr)
        hipLaunchKernelGGL(intra_chain_residual_kernel, dim3(32, nlevels), dim3(64), 0, st, static_cast<const unsigned char *>(base), lv, nlevels, bit_depth,
                           const_cast<int16_t *>(coeffs));
#define CHAIN_LAUNCH(PIX, CLK) hipLaunchKernelGGL((intra_chain_kernel<PIX, CLK>), dim3(1), dim3(64 * kChainWaves), 0, st, ps, static_cast<const unsigned char *>(base), lv, nlevels, bit_depth, coeffs, g_chain_agent_acquire, g_chain_clocks)
    if (g_chain_clocks) { if (bit_depth == 8) CHAIN_LAUNCH(uint8_t, true); else CHAIN_LAUNCH(uint16_t, true); }
    else                { if (bit_depth == 8) CHAIN_LAUNCH(uint8_t, false); else CHAIN_LAUNCH(uint16_t, false); }
#undef CHAIN_LAUNCH
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

#include "ctb_kernels.hpp"      // the CTB executor runs tu_dispatch on its LDS tiles

extern "C" int ohevc_dev_tu_batch(const ohevc_plane planes[3], int bit_depth, int log2_size, int kind,
                                  const ohevc_tu_job *jobs, int njobs, const int16_t *coeffs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(log2_size >= 2 && log2_size <= 5, "log2_size must be 2..5");
    OHEVC_REQUIRE(kind >= 0 && kind < OHEVC_TU_NKINDS, "unknown residual kind");
    OHEVC_REQUIRE(kind != OHEVC_TU_DST4 || log2_size == 2, "DST is 4x4 only");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    if (kind == OHEVC_TU_CROSS) {                           // generic two-blocks-per-workgroup body lives in the segmented kernel
        const ohevc_tu_segment sg = { log2_size, kind, 0, njobs };
        return ohevc_dev_tu_multi(planes, bit_depth, &sg, 1, jobs, coeffs, stream);
    }
    OHEVC_REQUIRE(jobs != nullptr, "jobs");
    OHEVC_REQUIRE(kind == OHEVC_TU_DC || coeffs != nullptr, "coeffs");
    OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(jobs) & 15) == 0, "jobs must be 16-byte aligned");
    OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(coeffs) & 15) == 0, "coeffs must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return bit_depth == 8 ? launch_tu<uint8_t>(ps, bit_depth, log2_size, kind, jobs, njobs, coeffs, st)
                          : launch_tu<uint16_t>(ps, bit_depth, log2_size, kind, jobs, njobs, coeffs, st);
}

// workgroups a segment of n jobs needs (must match the per-kind launch geometry in launch_tu)
static int tu_workgroups(int log2, int kind, int n)
{
    if (kind == OHEVC_TU_IDCT && log2 >= 3) { const int per_wg = 4 * (64 >> log2); return (n + per_wg - 1) / per_wg; }
    if (kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4) return (n + 255) / 256;
    if (kind == OHEVC_TU_CROSS) return (n + 1) / 2;
    return (int)((((long long)n << log2) + 255) / 256);
}

extern "C" int ohevc_dev_tu_multi(const ohevc_plane planes[3], int bit_depth, const ohevc_tu_segment *segs, int nsegs,
                                  const ohevc_tu_job *jobs, const int16_t *coeffs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(planes != nullptr && (nsegs == 0 || segs != nullptr), "null argument");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(nsegs >= 0 && nsegs <= TU_MAX_SEGMENTS, "too many segments (max 40)");
    OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(jobs) & 15) == 0 && (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0, "jobs/coeffs must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps);
    if (rc != OHEVC_OK) return rc;
    TuSegTable tab = {};
    int wgs = 0;
    for (int i = 0; i < nsegs; i++) {
        const ohevc_tu_segment &sg = segs[i];
        OHEVC_REQUIRE(sg.log2_size >= 2 && sg.log2_size <= 5 && sg.kind < OHEVC_TU_NKINDS && sg.njobs >= 0, "bad segment");
        OHEVC_REQUIRE(sg.kind != OHEVC_TU_DST4 || sg.log2_size == 2, "DST is 4x4 only");
        OHEVC_REQUIRE(sg.kind != OHEVC_TU_CROSS || coeffs != nullptr, "coeffs");
        if (sg.njobs == 0) continue;
        const int k = tab.nsegs++;
        tab.first_wg[k] = wgs; tab.first_job[k] = sg.first_job; tab.njobs[k] = sg.njobs;
        tab.log2[k] = (unsigned char)sg.log2_size; tab.kind[k] = (unsigned char)sg.kind;
        wgs += tu_workgroups(sg.log2_size, sg.kind, sg.njobs);
    }
    tab.first_wg[tab.nsegs] = wgs;
    if (wgs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(jobs != nullptr, "jobs");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bit_depth == 8) hipLaunchKernelGGL((tu_multi_kernel<uint8_t>), dim3(wgs), dim3(256), 0, st, ps, tab, jobs, coeffs, bit_depth);
    else                hipLaunchKernelGGL((tu_multi_kernel<uint16_t>), dim3(wgs), dim3(256), 0, st, ps, tab, jobs, coeffs, bit_depth);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_level_phase_workgroups(int type, int log2_size, int kind, int njobs)
{
    if (njobs <= 0) return 0;
    return type == 0 ? (njobs + 3) / 4 : tu_workgroups(log2_size, kind, njobs);
}

extern "C" int ohevc_dev_levels(const ohevc_plane planes[3], int bit_depth, const ohevc_level_phase *phases, int nphases, int total_wgs,
                                uint32_t *sync, const uint32_t *need, const ohevc_intra_job *intra_jobs, const ohevc_intra_cip *cips,
                                const ohevc_tu_job *tu_jobs, const int16_t *coeffs, void *stream)
{
    using namespace ohevc;
    static_assert(sizeof(LevelPhase) == sizeof(ohevc_level_phase), "phase record layout");
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
    OHEVC_REQUIRE(nphases >= 0 && total_wgs >= 0, "negative count");
    if (nphases == 0 || total_wgs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(phases != nullptr && sync != nullptr && need != nullptr, "null argument");
    OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(intra_jobs) & 15) == 0 && (reinterpret_cast<uintptr_t>(tu_jobs) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0 && (reinterpret_cast<uintptr_t>(cips) & 15) == 0, "job arrays must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    // 24 workgroups per XCD are dispatched (8 XCDs, round-robin by workgroup number); the ones on the leader's XCD stay.
    // Successive launches name different leaders, so the chains of pictures in flight spread over the XCDs.
    static std::atomic<unsigned> rotation{0};
    const int grid = 192, leader = (int)(rotation.fetch_add(1) % 8u);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LevelPhase *ph = reinterpret_cast<const LevelPhase *>(phases);
    if (bit_depth == 8) hipLaunchKernelGGL((levels_kernel<uint8_t>), dim3(grid), dim3(256), 0, st, ps, ph, nphases, total_wgs, sync, need, leader, intra_jobs, cips, tu_jobs, coeffs, bit_depth);
    else                hipLaunchKernelGGL((levels_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, ps, ph, nphases, total_wgs, sync, need, leader, intra_jobs, cips, tu_jobs, coeffs, bit_depth);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}

extern "C" int ohevc_debug_set_tu_variant(int variant)
{
    int old = ohevc::g_tu_variant;
    ohevc::g_tu_variant = variant;
    return old;
}

extern "C" int ohevc_debug_set_tu_pipe_workgroups(int n)
{
    int old = ohevc::g_tu_pipe_wgs;
    if (n > 0) ohevc::g_tu_pipe_wgs = n;
    return old;
}

extern "C" int ohevc_debug_has_lab(void)
{
#ifdef OHEVC_LAB
    return 1;
#else
    return 0;
#endif
}

extern "C" int ohevc_debug_mfma_i8_probe(const void *a, const void *b, void *d, int nprobes, void *stream)
{
    using namespace ohevc;
#ifdef OHEVC_LAB
    OHEVC_REQUIRE(a && b && d && nprobes > 0, "null argument");
    hipLaunchKernelGGL(mfma_i8_probe_kernel, dim3(nprobes), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const v4i *>(a),
                       static_cast<const v4i *>(b), static_cast<v16i *>(d));
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
#else
    (void)a; (void)b; (void)d; (void)nprobes; (void)stream;
    OHEVC_REQUIRE(false, "probe kernels exist in the lab build only (make -C openhevc_amd/csrc LAB=1)");
#endif
}

extern "C" int ohevc_debug_lds_tr16_probe(const void *addr, void *out, int nprobes, void *stream)
{
    using namespace ohevc;
#ifdef OHEVC_LAB
    OHEVC_REQUIRE(addr && out && nprobes > 0, "null argument");
    hipLaunchKernelGGL(lds_tr16_probe_kernel, dim3(nprobes), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const int *>(addr),
                       static_cast<u32x2 *>(out));
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
#else
    (void)addr; (void)out; (void)nprobes; (void)stream;
    OHEVC_REQUIRE(false, "probe kernels exist in the lab build only (make -C openhevc_amd/csrc LAB=1)");
#endif
}

extern "C" const char *ohevc_tu_kernel_name(int bit_depth, int log2_size, int kind)
{
    if (kind == OHEVC_TU_IDCT && log2_size == 5) {
        const int variant = ohevc::g_tu_variant >= 0 ? ohevc::g_tu_variant : 2048;
#ifdef OHEVC_LAB
        if ((variant & 2048) && (variant & 32768)) return "tu_tile_traffic_kernel";
        if ((variant & 2048) && (variant & 16384)) return "tu_idct32_tile2_kernel";
        if ((variant & 2048) && (variant & (4096 | 8192))) return "tu_idct32_tile_kernel";
        if (variant & 256) return "tu_idct32_mfma_kernel";
#endif
        if (variant & 2048) return "tu_idct32_tile1_kernel";
    }
    if (kind == OHEVC_TU_IDCT && log2_size >= 3) return "tu_idct_add_kernel";
    if (kind == OHEVC_TU_IDCT || kind == OHEVC_TU_DST4) return "tu_4x4_kernel";
    (void)bit_depth;
    return "tu_rows_kernel";
}
