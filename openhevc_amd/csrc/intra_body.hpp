// intra_body.hpp -- the intra prediction of one block by one wavefront (shared by intra_kernels.hip and the level
// kernel in tu_kernels.hip).  See intra_kernels.hip for the mapping to hevcpred_template.c.
#pragma once
#include "common.hpp"

namespace ohevc {

// intraPredAngle and invAngle (H.265 tables 8-4 / 8-5; hevcpred_template.c:425-433) as ARITHMETIC on packed constants, not as tables in
// memory: a table look-up is a vector-memory load in the middle of the prediction, and vector memory returns in order - the wait for that one
// byte was a wait for every load issued before it, the HBM prefetches of the chain kernel included (s_waitcnt vmcnt(0) behind
// global_load_sbyte in the round-4 listing of every intra kernel).
//   |angle| by distance from the horizontal (10) / vertical (26) mode: 0 2 5 9 13 17 21 26 32, six bits each;
//   |invAngle| = 8192 / |angle| rounded: 4096 1638 910 630 482 390 315 256, thirteen bits each in two words.
__device__ __forceinline__ int intra_pred_angle(const int mode)          // mode 2 .. 34
{
    const bool vert = mode >= 18;
    const int d = mode - (vert ? 26 : 10), a = d < 0 ? -d : d;
    const int mag = (int)((0x2069544d245080ull >> (6 * a)) & 63ull);
    return (d < 0) == vert ? -mag : mag;
}
__device__ __forceinline__ int intra_inv_angle(const int mode)           // mode 11 .. 25 (the negative angles)
{
    const int d = mode - (mode >= 18 ? 26 : 10), a = d < 0 ? -d : d;      // 1 .. 8
    const unsigned long long w = a <= 4 ? 0x13b0e38ccd000ull : 0x8004ec30c1e2ull;
    return -(int)((w >> (13 * ((a - 1) & 3))) & 0x1fffull);
}

struct IntraShared {
    int top[68], left[68];        // element k of the reference arrays lives at [k + 1]  (k = -1 .. 2N-1)
    int ftop[68], fleft[68];
    int ref[100];                 // angular reference, ref[k] at [k + 32]  (k = -32 .. 2N)
};

// One wavefront predicts one block.  WAVE_SYNC = false: the wavefront is the whole workgroup (intra_kernel) and the
// LDS hand-offs use s_barrier; true: several wavefronts of one workgroup each run their own block on their own
// IntraShared (levels_kernel), so the hand-offs only have to order this wavefront's own LDS accesses.
template <typename Pixel, bool WAVE_SYNC>
__device__ __forceinline__ void intra_body(IntraShared &sh, const int lane, const PlaneSet planes, const ohevc_intra_job jb, const int bit_depth,
                                           const ohevc_intra_cip *__restrict__ cips)
{
    const int log2 = jb.log2_size, n = 1 << log2, n2 = 2 * n, mode = jb.mode;
    const int stride = PLANE_STRIDE3(planes, jb.plane);
    unsigned char *blk = PLANE_PTR3(planes, jb.plane) + (size_t)jb.y * stride + (size_t)jb.x * sizeof(Pixel);
    bool c_bl = jb.flags & OHEVC_INTRA_BOTTOM_LEFT, c_l = jb.flags & OHEVC_INTRA_LEFT, c_ul = jb.flags & OHEVC_INTRA_UP_LEFT;
    bool c_u = jb.flags & OHEVC_INTRA_UP, c_ur = jb.flags & OHEVC_INTRA_UP_RIGHT;
    const int bl_size = jb.bottom_left_size, tr_size = jb.top_right_size;
    int *t = sh.top + 1, *l = sh.left + 1;
#define INTRA_SYNC() do { if (WAVE_SYNC) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); } while (0)
#define REC(x, y) ((int)*reinterpret_cast<const Pixel *>(blk + (ptrdiff_t)(y) * stride + (ptrdiff_t)(x) * (int)sizeof(Pixel)))

    const bool cip = (jb.flags2 & OHEVC_INTRA2_CIP) && cips != nullptr;
    if (cip) {            // memset(left/top, 128, ...) of BYTES, top[-1] = 128  (:160-162)
        const int fill = sizeof(Pixel) == 2 ? 0x8080 : 128;
        t[lane] = fill; l[lane] = fill;
        if (lane == 63) { t[-1] = 128; l[-1] = 128; }
        INTRA_SYNC();
    }
    // ---- gather what is available (:164-183); samples beyond the picture replicate the last valid one
    //      Three loads per lane (row above, column left, the corner), unconditional and issued together: a lane with nothing to fetch reads
    //      sample (0, 0) of the block and drops it.  Behind their conditions each load waited for the one before it - three memory round
    //      trips at the head of every block, and a dependency level of a picture lasts as long as one block.
    {
        const int k = lane;
        const bool top_ok = k < n2 && (k < n ? c_u : c_ur), left_ok = k < n2 && (k < n ? c_l : c_bl), corner_ok = lane == 63 && c_ul;
        const int tx = k < n ? k : (k < n + tr_size ? k : n + tr_size - 1), ly = k < n ? k : (k < n + bl_size ? k : n + bl_size - 1);
        const int tv = REC(top_ok ? tx : 0, top_ok ? -1 : 0), lv = REC(left_ok ? -1 : 0, left_ok ? ly : 0), cv = REC(corner_ok ? -1 : 0, corner_ok ? -1 : 0);
        if (top_ok) t[k] = tv;
        if (left_ok) l[k] = lv;
        if (corner_ok) { l[-1] = cv; t[-1] = cv; }
    }
    INTRA_SYNC();

    // ---- constrained intra prediction (:185-249): samples of inter-coded neighbours are overwritten from the nearest
    //      intra-coded ones.  The walk is order-dependent, so one lane runs it over the LDS arrays (rare streams only).
    if (cip && (c_bl || c_l || c_ul || c_u || c_ur)) {
        if (lane == 0) {
            const ohevc_intra_cip cr = cips[jb.cip_index];
            // 65 bits each: bit 0 = the corner (k = -1), bits 1..64 = k = 0..63; held as corner + 64-bit mask (no indexing)
            unsigned long long tmask = 0, lmask = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                tmask |= (unsigned long long)(((unsigned)cr.top_bits[b] | ((unsigned)cr.top_bits[b + 1] << 8)) >> 1 & 0xff) << (8 * b);
                lmask |= (unsigned long long)(((unsigned)cr.left_bits[b] | ((unsigned)cr.left_bits[b + 1] << 8)) >> 1 & 0xff) << (8 * b);
            }
            const int tcorner = cr.top_bits[0] & 1, lcorner = cr.left_bits[0] & 1;
            auto TB = [&](int k) { return k < 0 ? tcorner : (int)((tmask >> k) & 1); };           // IS_INTRA(k, -1)
            auto LB = [&](int k) { return k < 0 ? lcorner : (int)((lmask >> k) & 1); };           // IS_INTRA(-1, k)
            const int smx = cr.size_max_x, smy = cr.size_max_y;
            int j = n + (c_bl ? bl_size : 0) - 1;
            if (c_bl || c_l || c_ul) {
                while (j > -1 && !LB(j)) j--;
                if (!LB(j)) {
                    j = 0;
                    while (j < smx && !TB(j)) j++;
                    for (int i = j; i > -1; i--) if (!TB(i - 1)) t[i - 1] = t[i];
                    l[-1] = t[-1];
                }
            } else {
                j = 0;
                while (j < smx && !TB(j)) j++;
                if (j > 0) {
                    if (cr.x0_nonzero) {
                        for (int i = j; i > -1; i--) if (!TB(i - 1)) t[i - 1] = t[i];
                    } else {
                        for (int i = j; i > 0; i--) if (!TB(i - 1)) t[i - 1] = t[i];
                        t[-1] = t[0];
                    }
                }
                l[-1] = t[-1];
            }
            l[-1] = t[-1];
            if (c_bl || c_l) {
                int a = l[-1];
                for (int i = 0; i < smy; i += 4) {
                    if (!LB(i)) { l[i] = l[i + 1] = l[i + 2] = l[i + 3] = a; } else a = l[i + 3];
                }
            }
            if (!c_l) for (int i = 0; i < n; i++) l[i] = l[-1];
            if (!c_bl) for (int i = n; i < n2; i++) l[i] = l[n - 1];
            if (cr.x0_nonzero && cr.y0_nonzero) {
                int a = l[smy - 1];
                for (int i = smy - 1; i > -1; i -= 4) {
                    if (!LB(i - 3)) { l[i - 3] = l[i - 2] = l[i - 1] = l[i] = a; } else a = l[i - 3];
                }
                if (!LB(-1)) l[-1] = l[0];
            } else if (!cr.x0_nonzero) {
                for (int i = 0; i < smy; i++) l[i] = 0;
            } else {
                int a = l[smy - 1];
                for (int i = smy - 1; i > -1; i -= 4) {
                    if (!LB(i - 3)) { l[i - 3] = l[i - 2] = l[i - 1] = l[i] = a; } else a = l[i - 3];
                }
            }
            t[-1] = l[-1];
            if (cr.y0_nonzero) {
                int a = l[-1];
                for (int i = 0; i < smx; i += 4) {
                    if (!TB(i)) { t[i] = t[i + 1] = t[i + 2] = t[i + 3] = a; } else a = t[i + 3];
                }
            }
        }
        INTRA_SYNC();
    }

    // ---- substitution of unavailable samples (:251-286); every branch is wave-uniform
    if (!c_bl) {
        if (c_l) {
            const int v = l[n - 1];
            if (lane >= n && lane < n2) l[lane] = v;
        } else if (c_ul) {
            const int v = l[-1];
            if (lane < n2) l[lane] = v;
            c_l = true;
        } else if (c_u) {
            const int v = t[0];
            if (lane < n2) l[lane] = v;
            if (lane == 63) l[-1] = v;
            c_ul = c_l = true;
        } else if (c_ur) {
            const int v = t[n];
            if (lane < n) t[lane] = v;
            if (lane < n2) l[lane] = v;
            if (lane == 63) l[-1] = v;
            c_u = c_ul = c_l = true;
        } else {
            const int v = 1 << (bit_depth - 1);
            if (lane < n2) { t[lane] = v; l[lane] = v; }
            if (lane == 63) l[-1] = v;
        }
        INTRA_SYNC();
    }
    if (!c_l) {
        const int v = l[n];
        if (lane < n) l[lane] = v;
        INTRA_SYNC();
    }
    if (!c_ul) {
        if (lane == 63) l[-1] = l[0];
        INTRA_SYNC();
    }
    if (!c_u) {
        const int v = l[-1];
        if (lane < n) t[lane] = v;
    }
    if (!c_ur) {
        INTRA_SYNC();
        const int v = t[n - 1];
        if (lane >= n && lane < n2) t[lane] = v;
    }
    INTRA_SYNC();
    if (lane == 63) t[-1] = l[-1];
    INTRA_SYNC();

    // ---- reference smoothing (:289-327)
    if (!(jb.flags & OHEVC_INTRA_NO_SMOOTHING) && mode != 1 && n != 4) {
        const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
        const int dist = dv < dh ? dv : dh, thresh = log2 == 3 ? 7 : log2 == 4 ? 1 : 0;
        if (dist > thresh) {
            int *ft = sh.ftop + 1, *fl = sh.fleft + 1;
            const int lim = 1 << (bit_depth - 5);
            int a = t[-1] + t[n2 - 1] - 2 * t[n - 1], b = l[-1] + l[n2 - 1] - 2 * l[n - 1];
            a = a < 0 ? -a : a; b = b < 0 ? -b : b;
            if ((jb.flags & OHEVC_INTRA_STRONG) && log2 == 5 && a < lim && b < lim) {
                const int t0 = t[-1], t63 = t[63], l0 = l[-1], l63 = l[63];
                if (lane < 63) {
                    ft[lane] = ((63 - lane) * t0 + (lane + 1) * t63 + 32) >> 6;
                    fl[lane] = ((63 - lane) * l0 + (lane + 1) * l63 + 32) >> 6;
                } else {
                    ft[63] = t63; fl[63] = l63; ft[-1] = t0; fl[-1] = l0;
                }
            } else {
                if (lane < n2 - 1) {
                    ft[lane] = (t[lane + 1] + 2 * t[lane] + t[lane - 1] + 2) >> 2;
                    fl[lane] = (l[lane + 1] + 2 * l[lane] + l[lane - 1] + 2) >> 2;
                } else if (lane == n2 - 1) {
                    ft[lane] = t[lane]; fl[lane] = l[lane];
                }
                if (lane == 63) ft[-1] = fl[-1] = (l[0] + 2 * l[-1] + t[0] + 2) >> 2;
                // (for N == 32, lane 63 == n2 - 1 executes both assignments above)
            }
            t = ft; l = fl;
            INTRA_SYNC();
        }
    }

    // ---- prediction
    const int maxv = (1 << bit_depth) - 1;
    const bool luma_edge = (jb.flags & OHEVC_INTRA_LUMA_EDGE) && n < 32;
    if (mode == 0) {                                   // pred_planar, :359-372
        for (int idx = lane; idx < n * n; idx += 64) {
            const int y = idx >> log2, x = idx & (n - 1);
            const int v = ((n - 1 - x) * l[y] + (x + 1) * t[n] + (n - 1 - y) * t[x] + (y + 1) * l[n] + n) >> (log2 + 1);
            *reinterpret_cast<Pixel *>(blk + (size_t)y * stride + (size_t)x * sizeof(Pixel)) = (Pixel)v;
        }
    } else if (mode == 1) {                            // pred_dc, :388-417
        int part = 0;
        if (lane < n) part = l[lane] + t[lane];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
        const int dc = (part + n) >> (log2 + 1);
        for (int idx = lane; idx < n * n; idx += 64) {
            const int y = idx >> log2, x = idx & (n - 1);
            int v = dc;
            if (luma_edge) {
                if (x == 0 && y == 0) v = (l[0] + 2 * dc + t[0] + 2) >> 2;
                else if (y == 0)      v = (t[x] + 3 * dc + 2) >> 2;
                else if (x == 0)      v = (l[y] + 3 * dc + 2) >> 2;
            }
            *reinterpret_cast<Pixel *>(blk + (size_t)y * stride + (size_t)x * sizeof(Pixel)) = (Pixel)v;
        }
    } else {                                           // pred_angular, :419-510
        const int angle = intra_pred_angle(mode), last = (n * angle) >> 5;
        const bool vertical = mode >= 18;
        const int *mainr = vertical ? t : l, *sider = vertical ? l : t;
        int *ref = sh.ref + 32;
        for (int k = lane; k <= n2; k += 64) ref[k] = mainr[k - 1];
        if (angle < 0 && last < -1) {
            const int inv = intra_inv_angle(mode);
            const int k = -1 - lane;                   // k = -1 .. last
            if (k >= last) ref[k] = sider[-1 + ((k * inv + 128) >> 8)];
        }
        INTRA_SYNC();
        for (int idx = lane; idx < n * n; idx += 64) {
            const int y = idx >> log2, x = idx & (n - 1);
            const int a = vertical ? y : x, b = vertical ? x : y;   // a: minor axis (steps the angle), b: along the reference
            const int pos = (a + 1) * angle, i2 = pos >> 5, fact = pos & 31;
            int v = fact ? ((32 - fact) * ref[b + i2 + 1] + fact * ref[b + i2 + 2] + 16) >> 5 : ref[b + i2 + 1];
            if (luma_edge) {
                if (mode == 26 && x == 0) { v = t[0] + ((l[y] - l[-1]) >> 1); v = v < 0 ? 0 : v > maxv ? maxv : v; }
                if (mode == 10 && y == 0) { v = l[0] + ((t[x] - t[-1]) >> 1); v = v < 0 ? 0 : v > maxv ? maxv : v; }
            }
            *reinterpret_cast<Pixel *>(blk + (size_t)y * stride + (size_t)x * sizeof(Pixel)) = (Pixel)v;
        }
    }
#undef REC
#undef INTRA_SYNC
}


}  // namespace ohevc
