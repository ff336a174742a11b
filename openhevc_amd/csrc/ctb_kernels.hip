// ctb_kernels.hip -- every intra-coded block of a picture in ONE launch: coding-tree blocks as tasks, their samples in LDS.
//
// What it replaces in the executor: the chain of dependency levels (prediction launch + residual launch per level; ~150 levels in a
// 1080p picture of flat random syntax).  Measured in round 2 (profiles/r02n_*): those ~300 launches are 1.1 ms of kernel time and,
// worse, ~1.5 ms of launch work on the host that the runtime serialises across decoding threads, and the pictures of a GOP wait for
// each other's chains.  The chain exists because intra_pred (hevcpred_template.c:30-357) reads the reconstructed row above / column
// left of its block: block k + 1 of a CTB needs block k's prediction + residual.  Inside one CTB that hand-off needs no memory at
// all if the CTB's samples live in LDS while its blocks are reconstructed in decoding order; between CTBs only the classic
// wavefront remains (a CTB needs its left, above-left, above and above-right neighbours: hls_decode_entry_wpp's 2-CTB lag,
// hevc.c:2779, pthread_slice.c:238-262).
//
//   * task = one CTB that contains intra-coded blocks; tasks are numbered in raster order and handed out through a ticket counter to
//     persistent single-wave workgroups, so a task only ever waits for tasks with smaller numbers - held by waves that are already
//     running: no assumption about residency or dispatch order;
//   * a wave loads its CTB (all three planes) plus the row above (with the above-right extension) and the column left into LDS,
//     walks the CTB's operations in decoding order - intra prediction (intra_body.hpp, the same code as intra_kernel's, reading and
//     writing the LDS tile) and the residual that follows it (any kind, tu_generic.hpp) - and writes the CTB back;
//   * CTB-to-CTB hand-off through the shared L2 of ONE XCD, as in the level kernel: the workgroups of one XCD class (blockIdx % 8)
//     stay, the first of them claims its XCC id as home and the others check theirs; stores are complete in L2 after
//     s_waitcnt vmcnt(0), readers drop their L1 (buffer_inv sc1) after seeing the flag.  Successive launches prefer different classes,
//     so the chains of pictures in flight spread over the XCDs.
#include <atomic>
#include "common.hpp"
#include "intra_body.hpp"
#include "tu_generic.hpp"

namespace ohevc {

struct CtbTask {                    // mirrors ohevc_ctb_task (include/ohevc_hip.h)
    unsigned short cx, cy;
    unsigned first_op, nops;
    int dep[4];
    unsigned reserved;
};
struct CtbParams {
    int log2_ctb, hshift, vshift, bit_depth;
    int ntasks, preferred;
};
enum { CTB_HOME = 0, CTB_TICKET = 1, CTB_DONE = 2 };       // layout of the sync words

// LDS tile of one colour plane: rows -1 .. H - 1, columns -1 .. W + EXT - 1 of the CTB; column 0 sits at byte 4 of a row (dword
// aligned), column -1 right below it
struct TileGeom {
    int off, stride;                // byte offset inside the tile area, bytes per row
    int x0, y0, w, h, ext;          // CTB origin / size / above-right extension in samples of this plane
};

template <typename Pixel, bool FULLCHROMA>
struct CtbLds {
    static constexpr int PXB = (int)sizeof(Pixel);
    static constexpr int LUMA = 65 * (4 + 96 * PXB);
    static constexpr int CHROMA = FULLCHROMA ? LUMA : 65 * (4 + 64 * PXB);       // 4:2:2 keeps the full height
    static constexpr int TILES = LUMA + 2 * CHROMA;
};

template <typename Pixel, bool FULLCHROMA>
__global__ __launch_bounds__(64) void ctb_kernel(PlaneSet planes, CtbParams prm, const CtbTask *__restrict__ tasks, const unsigned *__restrict__ ops,
                                                 const ohevc_intra_job *__restrict__ intra_jobs, const ohevc_intra_cip *__restrict__ cips,
                                                 const ohevc_tu_job *__restrict__ tu_jobs, const int16_t *__restrict__ coeffs, unsigned *sync)
{
    constexpr int PXB = (int)sizeof(Pixel);
    __shared__ __attribute__((aligned(16))) unsigned char tiles[CtbLds<Pixel, FULLCHROMA>::TILES];
    __shared__ IntraShared ish;
    __shared__ __attribute__((aligned(16))) short scratch[3 * 1024];          // tmp / luma residual / own residual of tu_generic.hpp
    __shared__ short dc_slot[8];
    const int lane = threadIdx.x;
    if ((int)(blockIdx.x & 7u) != prm.preferred) return;
    {   // one XCD does the whole chain: whoever of the preferred class comes first names it
        const unsigned mine = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) + 1u;       // HW_REG_XCC_ID + 1
        unsigned home = 0;
        if (lane == 0) {
            home = atomicCAS(&sync[CTB_HOME], 0u, mine);
            if (home == 0u) home = mine;
        }
        home = (unsigned)__builtin_amdgcn_readfirstlane((int)home);
        if (home != mine) return;
    }
    const int S = 1 << prm.log2_ctb, bd = prm.bit_depth;
    TileGeom tg[3];
    {
        int off = 0;
        for (int p = 0; p < 3; p++) {
            const int hs = p ? prm.hshift : 0, vs = p ? prm.vshift : 0;
            tg[p].w = S >> hs; tg[p].h = S >> vs;
            tg[p].ext = tg[p].w < 32 ? tg[p].w : 32;
            tg[p].stride = 4 + (tg[p].w + tg[p].ext) * PXB;
            tg[p].off = off;
            off += (tg[p].h + 1) * tg[p].stride;
        }
    }
    for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&sync[CTB_TICKET], 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= (unsigned)prm.ntasks) return;
        const CtbTask task = tasks[t];
        // ---- wait for the neighbours this CTB reads from (smaller ticket numbers: their holders are running)
        bool waited = false;
        for (int d = 0; d < 4; d++) {
            const int dep = task.dep[d];
            if (dep < 0) continue;
            while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sync[CTB_DONE + dep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
                __builtin_amdgcn_s_sleep(1);
            waited = true;
        }
        if (waited) xcd_acquire();                              // forget what this CU's L1 holds of the neighbours' samples
        // ---- the CTB, its row above and its column left -> LDS
        for (int p = 0; p < 3; p++) {
            TileGeom &g = tg[p];
            g.x0 = task.cx * g.w; g.y0 = task.cy * g.h;
            const int pw = planes.width[p], ph = planes.height[p], pstride = planes.stride[p];
            const unsigned char *src = planes.data[p];
            unsigned char *tile = tiles + g.off;
            const int dw_per_row = (g.w + g.ext) * PXB / 4, rows = g.h + 1;
            for (int i = lane; i < rows * dw_per_row; i += 64) {
                const int r = i / dw_per_row, dcol = i - r * dw_per_row;
                const int y = g.y0 - 1 + r, xb = g.x0 * PXB + dcol * 4;          // byte column inside the plane row
                if (y >= 0 && y < ph && xb < pw * PXB)
                    *reinterpret_cast<unsigned *>(tile + r * g.stride + 4 + dcol * 4) = *reinterpret_cast<const unsigned *>(src + (size_t)y * pstride + xb);
            }
            if (g.x0 > 0)
                for (int r = lane; r < rows; r += 64) {
                    const int y = g.y0 - 1 + r;
                    if (y >= 0 && y < ph)
                        *reinterpret_cast<Pixel *>(tile + r * g.stride + 4 - PXB) = *reinterpret_cast<const Pixel *>(src + (size_t)y * pstride + (size_t)(g.x0 - 1) * PXB);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the kernels' plane view of the tiles: sample (x, y) of plane p at tile + (y - y0 + 1) * stride + 4 + (x - x0) * PXB
        PlaneSet lp;
        for (int p = 0; p < 3; p++) {
            lp.data[p] = tiles + tg[p].off + (ptrdiff_t)(1 - tg[p].y0) * tg[p].stride + 4 - (ptrdiff_t)tg[p].x0 * PXB;
            lp.stride[p] = tg[p].stride;
            lp.width[p] = planes.width[p]; lp.height[p] = planes.height[p];
        }
        // ---- the CTB's operations in decoding order
        for (unsigned k = 0; k < task.nops; k++) {
            const unsigned op = ops[task.first_op + k];
            const unsigned idx = op & 0x1ffffffu;
            if (!(op >> 31)) {
                intra_body<Pixel, true>(ish, lane, lp, intra_jobs[idx], bd, cips);
            } else {
                const int log2 = (int)((op >> 29) & 3u) + 2, kind = (int)((op >> 25) & 15u), N = 1 << log2, NN = N * N;
                const ohevc_tu_job jb = tu_jobs[idx];
                short *tmp = scratch, *ry = scratch + 1024, *rc = scratch + 2048;
                int scale = 0;
                bool have_own = true;
                if (kind == OHEVC_TU_CROSS) {                   // hevc.c:1291-1365: own residual + (res_scale_val * luma residual) >> 3
                    const int kind_c = jb.reserved0 & 15, kind_y = jb.reserved0 >> 4;
                    scale = jb.dc;
                    residual_generic(kind_y, log2, coeffs + jb.reserved1, bd, tmp, ry, lane);
                    have_own = kind_c != 15;
                    if (have_own) residual_generic(kind_c, log2, coeffs + jb.coeff_off, bd, tmp, rc, lane);
                } else if (kind == OHEVC_TU_DC) {               // the coefficient travels in the job
                    if (lane == 0) dc_slot[0] = jb.dc;
                    CROSS_SYNC();
                    residual_generic(kind, log2, dc_slot, bd, tmp, rc, lane);
                } else if (kind == OHEVC_TU_PCM) {              // put_pcm: the samples replace the block
                    for (int o = lane; o < NN; o += 64) rc[o] = (coeffs + jb.coeff_off)[o];
                    CROSS_SYNC();
                } else {
                    residual_generic(kind, log2, coeffs + jb.coeff_off, bd, tmp, rc, lane);
                }
                const int stride = PLANE_STRIDE3(lp, jb.plane), maxv = (1 << bd) - 1;
                unsigned char *base = PLANE_PTR3(lp, jb.plane) + (ptrdiff_t)jb.y * stride + (ptrdiff_t)jb.x * PXB;
                for (int o = lane; o < NN; o += 64) {
                    Pixel *px = reinterpret_cast<Pixel *>(base + (ptrdiff_t)(o >> log2) * stride) + (o & (N - 1));
                    int res = have_own ? (int)rc[o] : 0;
                    if (kind == OHEVC_TU_CROSS) res = (int)(short)(res + ((scale * (int)ry[o]) >> 3));
                    const int v = (kind == OHEVC_TU_PCM ? 0 : (int)*px) + res;          // transform_add, hevcdsp_template.c:45-111
                    *px = (Pixel)(v < 0 ? 0 : v > maxv ? maxv : v);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- write the CTB back (the part of it inside the picture)
        for (int p = 0; p < 3; p++) {
            const TileGeom &g = tg[p];
            const int pw = planes.width[p], ph = planes.height[p], pstride = planes.stride[p];
            unsigned char *dst = planes.data[p];
            const unsigned char *tile = tiles + g.off;
            const int dw_per_row = g.w * PXB / 4;
            for (int i = lane; i < g.h * dw_per_row; i += 64) {
                const int r = i / dw_per_row, dcol = i - r * dw_per_row;
                const int y = g.y0 + r, xb = g.x0 * PXB + dcol * 4;
                if (y < ph && xb < pw * PXB)
                    *reinterpret_cast<unsigned *>(dst + (size_t)y * pstride + xb) = *reinterpret_cast<const unsigned *>(tile + (r + 1) * g.stride + 4 + dcol * 4);
            }
        }
        xcd_release();                                          // this wave's stores sit in the XCD's L2 ...
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&sync[CTB_DONE + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before the flag says so
    }
}

}  // namespace ohevc

extern "C" int ohevc_dev_ctbs(const ohevc_plane planes[3], int bit_depth, int chroma_format_idc, int log2_ctb_size, const ohevc_ctb_task *tasks, int ntasks,
                              const uint32_t *ops, const ohevc_intra_job *intra_jobs, const ohevc_intra_cip *cips, const ohevc_tu_job *tu_jobs,
                              const int16_t *coeffs, uint32_t *sync, void *stream)
{
    using namespace ohevc;
    static_assert(sizeof(CtbTask) == sizeof(ohevc_ctb_task) && sizeof(CtbTask) == 32, "task record layout");
    OHEVC_REQUIRE(planes != nullptr, "planes");
    OHEVC_REQUIRE(bit_depth >= 8 && bit_depth <= 12, "bit_depth must be 8..12");
    OHEVC_REQUIRE(chroma_format_idc >= 1 && chroma_format_idc <= 3, "chroma_format_idc must be 1..3");
    OHEVC_REQUIRE(log2_ctb_size >= 4 && log2_ctb_size <= 6, "log2_ctb_size must be 4..6");
    OHEVC_REQUIRE(ntasks >= 0, "ntasks");
    if (ntasks == 0) return OHEVC_OK;
    OHEVC_REQUIRE(tasks != nullptr && ops != nullptr && sync != nullptr, "null argument");
    OHEVC_REQUIRE((reinterpret_cast<uintptr_t>(tasks) & 15) == 0 && (reinterpret_cast<uintptr_t>(intra_jobs) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(tu_jobs) & 15) == 0 && (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(cips) & 15) == 0, "job arrays must be 16-byte aligned");
    PlaneSet ps;
    int rc = make_plane_set(planes, ps, 4);
    if (rc != OHEVC_OK) return rc;
    for (int i = 0; i < 3; i++) OHEVC_REQUIRE(planes[i].data != nullptr, "all three planes are needed");
    static std::atomic<unsigned> rotation{0};
    CtbParams prm;
    prm.log2_ctb = log2_ctb_size; prm.bit_depth = bit_depth;
    prm.hshift = chroma_format_idc == 1 || chroma_format_idc == 2; prm.vshift = chroma_format_idc == 1;
    prm.ntasks = ntasks; prm.preferred = (int)(rotation.fetch_add(1) % 8u);
    // 8 XCD classes x `per` single-wave workgroups; only the preferred class stays.  More waves than tasks can run side by side buy nothing.
    const int per = ntasks < 64 ? ntasks : 64;
    const dim3 grid(8 * per), block(64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const CtbTask *tk = reinterpret_cast<const CtbTask *>(tasks);
    const bool full = chroma_format_idc == 3;
    if (bit_depth == 8) {
        if (full) hipLaunchKernelGGL((ctb_kernel<uint8_t, true>), grid, block, 0, st, ps, prm, tk, ops, intra_jobs, cips, tu_jobs, coeffs, sync);
        else      hipLaunchKernelGGL((ctb_kernel<uint8_t, false>), grid, block, 0, st, ps, prm, tk, ops, intra_jobs, cips, tu_jobs, coeffs, sync);
    } else {
        if (full) hipLaunchKernelGGL((ctb_kernel<uint16_t, true>), grid, block, 0, st, ps, prm, tk, ops, intra_jobs, cips, tu_jobs, coeffs, sync);
        else      hipLaunchKernelGGL((ctb_kernel<uint16_t, false>), grid, block, 0, st, ps, prm, tk, ops, intra_jobs, cips, tu_jobs, coeffs, sync);
    }
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}
