// mc_kernels.hip -- luma (8-tap) / chroma (4-tap) motion compensation with uni / bi / weighted prediction.
//
// Replaces luma_mc_uni/bi + chroma_mc_uni/bi (hevc.c:1641-1949), the ten put_hevc_{qpel,epel}* table families
// (hevcdsp_template.c:610-1609) and vdsp.emulated_edge_mc (videodsp_template.c:26-100):
//   * one wavefront per prediction block, walking it in 16x16 tiles;
//   * the (w+T-1) x (h+T-1) reference window is gathered into LDS with CLAMPED picture coordinates -- the same
//     samples the reference's edge emulation would have replicated -- so no padded copy is ever made;
//   * horizontal pass -> int16 LDS tile (the reference's tmp_array, :763-776), vertical pass from LDS;
//   * bi-prediction evaluates both references inside the job, so the reference's int16 tmp[64*64] hand-off
//     between put_hevc_qpel and put_hevc_qpel_bi (hevc.c:1761-1764) stays in registers.
// The four (mx,my) cases of the reference collapse into one exact formulation:
//   tmp  = mx ? (sum fh*src) >> (BD-8) : src << (14-BD)            (14-bit intermediate, fits int16)
//   v14  = my ? (sum fv*tmp) >> 6      : tmp                       ((S << (14-BD)) >> 6 == S >> (BD-8) exactly)
#include "common.hpp"

namespace ohevc {

// ff_hevc_qpel_filters / ff_hevc_epel_filters (libavcodec/hevcdsp.c:1028-1042): the standard's interpolation taps
__device__ const signed char kLumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};
__device__ const signed char kChromaTaps[8][4] = {
    { 0, 64, 0, 0 },
    { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 },
    { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 },
};

constexpr int MC_TILE = 16;
constexpr int MC_WIN  = MC_TILE + 7;        // window rows/cols for the 8-tap case
constexpr int MC_WINP = MC_WIN + 1;         // padded LDS row (int16)

struct McShared {
    short win[MC_WIN][MC_WINP];
    short tmp[MC_WIN][MC_TILE];
};

template <typename Pixel>
__device__ __forceinline__ void mc_tile_ref(McShared &sh, const ohevc_plane &ref, int sx, int sy, int mx, int my,
                                            bool luma, int tw, int th, int bit_depth, int lane, int *v)
{
    const int taps = luma ? 8 : 4, before = luma ? 3 : 1;
    const int ww = tw + taps - 1, wh = th + taps - 1;
    const unsigned char *base = static_cast<const unsigned char *>(ref.data);
    const int xmax = ref.width - 1, ymax = ref.height - 1;
    // ---- gather window, coordinates clamped to the picture (== emulated_edge_mc's replication)
    for (int idx = lane; idx < ww * wh; idx += 64) {
        const int wy = idx / ww, wx = idx - wy * ww;
        int x = sx + wx - before, y = sy + wy - before;
        x = x < 0 ? 0 : x > xmax ? xmax : x;
        y = y < 0 ? 0 : y > ymax ? ymax : y;
        const Pixel p = *reinterpret_cast<const Pixel *>(base + (size_t)y * ref.stride + (size_t)x * sizeof(Pixel));
        sh.win[wy][wx] = (short)p;
    }
    __syncthreads();
    // ---- horizontal pass over all window rows
    const signed char *fh = luma ? kLumaTaps[mx] : kChromaTaps[mx];
    for (int idx = lane; idx < tw * wh; idx += 64) {
        const int r = idx / tw, x = idx - r * tw;
        int s;
        if (mx) {
            s = 0;
            for (int k = 0; k < taps; k++) s += fh[k] * (int)(unsigned short)sh.win[r][x + k];
            s >>= bit_depth - 8;
        } else {
            s = (int)(unsigned short)sh.win[r][x + before] << (14 - bit_depth);
        }
        sh.tmp[r][x] = (short)s;
    }
    __syncthreads();
    // ---- vertical pass: up to 4 samples per lane
    const signed char *fv = luma ? kLumaTaps[my] : kChromaTaps[my];
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
        const int idx = lane + 64 * k4;
        int s = 0;
        if (idx < tw * th) {
            const int y = idx / tw, x = idx - y * tw;
            if (my) {
                for (int k = 0; k < taps; k++) s += fv[k] * (int)sh.tmp[y + k][x];
                s >>= 6;
            } else {
                s = sh.tmp[y + before][x];
            }
        }
        v[k4] = s;
    }
    __syncthreads();
}

template <typename Pixel>
__global__ __launch_bounds__(64) void mc_kernel(PlaneSet dst, const ohevc_plane *__restrict__ refs,
                                                const ohevc_mc_job *__restrict__ jobs, int njobs, int bit_depth)
{
    __shared__ McShared sh;
    const int lane = threadIdx.x;
    const ohevc_mc_job jb = jobs[blockIdx.x];
    const bool luma = jb.plane == 0, bi = jb.flags & OHEVC_MC_BI, weighted = jb.flags & OHEVC_MC_WEIGHTED;
    const ohevc_plane ref0 = refs[3 * jb.ref0 + jb.plane];
    const ohevc_plane ref1 = refs[3 * (bi ? jb.ref1 : jb.ref0) + jb.plane];
    unsigned char *dbase = PLANE_PTR3(dst, jb.plane);
    const int dstride = PLANE_STRIDE3(dst, jb.plane);
    const int maxv = (1 << bit_depth) - 1;

    for (int ty = 0; ty < jb.h; ty += MC_TILE)
        for (int tx = 0; tx < jb.w; tx += MC_TILE) {
            const int tw = jb.w - tx < MC_TILE ? jb.w - tx : MC_TILE;
            const int th = jb.h - ty < MC_TILE ? jb.h - ty : MC_TILE;
            int v0[4], v1[4];
            mc_tile_ref<Pixel>(sh, ref0, jb.sx0 + tx, jb.sy0 + ty, jb.mx0, jb.my0, luma, tw, th, bit_depth, lane, v0);
            if (bi)
                mc_tile_ref<Pixel>(sh, ref1, jb.sx1 + tx, jb.sy1 + ty, jb.mx1, jb.my1, luma, tw, th, bit_depth, lane, v1);
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                const int idx = lane + 64 * k4;
                if (idx >= tw * th) continue;
                const int y = idx / tw, x = idx - y * tw;
                int out;
                if (!bi && !weighted) {             // put_hevc_*_uni_*: hevcdsp_template.c:626-640,796-943
                    const int shift = 14 - bit_depth;
                    out = (v0[k4] + (1 << (shift - 1))) >> shift;
                } else if (bi && !weighted) {       // put_hevc_*_bi_*: :642-666,822-983
                    const int shift = 15 - bit_depth;
                    out = (v1[k4] + v0[k4] + (1 << (shift - 1))) >> shift;
                } else if (!bi) {                   // put_hevc_*_uni_w_*: :668-690,985-1134
                    const int shift = jb.denom + 14 - bit_depth;
                    out = ((v0[k4] * jb.wx0 + (1 << (shift - 1))) >> shift) + jb.ox0 * (1 << (bit_depth - 8));
                } else {                            // put_hevc_*_bi_w_*: :692-716,1012-1174
                    const int log2wd = jb.denom + 14 - bit_depth;
                    const int o0 = jb.ox0 * (1 << (bit_depth - 8)), o1 = jb.ox1 * (1 << (bit_depth - 8));
                    out = (v1[k4] * jb.wx1 + v0[k4] * jb.wx0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
                }
                out = out < 0 ? 0 : out > maxv ? maxv : out;
                *reinterpret_cast<Pixel *>(dbase + (size_t)(jb.y + ty + y) * dstride + (size_t)(jb.x + tx + x) * sizeof(Pixel)) = (Pixel)out;
            }
        }
}

}  // namespace ohevc

extern "C" int ohevc_dev_mc_batch(const ohevc_plane dst[3], const ohevc_plane *refs, int n_ref_slots, int bit_depth,
                                  const ohevc_mc_job *jobs, int njobs, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(dst != nullptr, "dst");
    OHEVC_REQUIRE(bit_depth >= 8 && bit_depth <= 12, "bit_depth must be 8..12");
    OHEVC_REQUIRE(njobs >= 0, "njobs");
    if (njobs == 0) return OHEVC_OK;
    OHEVC_REQUIRE(refs != nullptr && n_ref_slots > 0, "refs");
    OHEVC_REQUIRE(jobs != nullptr && (reinterpret_cast<uintptr_t>(jobs) & 15) == 0, "jobs must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(dst, ps, bit_depth > 8 ? 2 : 1);
    if (rc != OHEVC_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bit_depth == 8)
        hipLaunchKernelGGL((mc_kernel<uint8_t>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth);
    else
        hipLaunchKernelGGL((mc_kernel<uint16_t>), dim3(njobs), dim3(64), 0, st, ps, refs, jobs, njobs, bit_depth);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}
