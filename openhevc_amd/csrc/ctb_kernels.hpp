// ctb_kernels.hpp -- (included at the end of tu_kernels.hip: it runs that file's residual bodies) every intra-coded block of a picture in ONE launch: coding-tree blocks as tasks, their samples in LDS.
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
#pragma once

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
enum { CTB_HOME = 0, CTB_TICKET = 1, CTB_DONE = 2 };       // layout of the sync words: home XCC, ticket counter, one done flag per task,
                                                            // then one progress word per task (diagnosis: OHEVC_CTB_DEBUG, ctx.hip)
constexpr unsigned kCtbSpinLimit = 1u << 22;                // polls (of >= 64 cycles each) before a wait gives up: a protocol error must not hang the device

// LDS tile of one colour plane: rows -1 .. H - 1, columns -1 .. W + EXT - 1 of the CTB; column 0 sits at byte 16 of a row and rows are a
// multiple of 16 bytes apart (the residual bodies move 8- / 16-byte pieces of naturally aligned blocks), column -1 right below it
struct TileGeom {
    int off, stride;                // byte offset inside the tile area, bytes per row
    int x0, y0, w, h, ext;          // CTB origin / size / above-right extension in samples of this plane
};

template <typename Pixel, bool FULLCHROMA>
struct CtbLds {
    static constexpr int PXB = (int)sizeof(Pixel);
    static constexpr int LUMA = 65 * (16 + 96 * PXB);
    static constexpr int CHROMA = FULLCHROMA ? LUMA : 65 * (16 + 64 * PXB);      // 4:2:2 keeps the full height
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
    __shared__ __attribute__((aligned(16))) unsigned char tu_lds[2 * TuLayout<5>::WAVE_BYTES];      // what one wave of the residual bodies needs
    const int lane = threadIdx.x;
    if ((int)(blockIdx.x & 7u) != prm.preferred) return;
    {   // one XCD does the whole chain: whoever of the preferred class comes first names it
        const unsigned mine = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) + 1u;       // HW_REG_XCC_ID + 1
        // (every lane makes the same attempt: whichever wins, all of them end up with the same answer -- see the note on branches below)
        unsigned home = atomicCAS(&sync[CTB_HOME], 0u, mine);
        if (home == 0u) home = mine;
        home = (unsigned)__builtin_amdgcn_readfirstlane((int)(home & 0x7fffffffu));
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
            tg[p].stride = 16 + (tg[p].w + tg[p].ext) * PXB;
            tg[p].off = off;
            off += (tg[p].h + 1) * tg[p].stride;
        }
    }
    // Every branch of this loop is WAVE-uniform on purpose, and single-lane effects are expressed through operands (lane 0 adds 1 to the
    // ticket, the others 0; all lanes store the same flag).  With an `if (lane == 0)` around the flag store the compiler - for which lanes
    // are independent threads - is free to let the other 63 lanes run ahead into the next task's wait loop and park lane 0's store behind
    // it: observed on the device (profiles/r02r_ctb_debug.txt: every wave past its write-back, no flag ever set, the kernel never ends).
    for (;;) {
        unsigned t = atomicAdd(&sync[CTB_TICKET], lane == 0 ? 1u : 0u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);          // lane 0's return value = this wave's ticket
        if (t >= (unsigned)prm.ntasks) return;
        CtbTask task = tasks[t];
        task.nops = (unsigned)__builtin_amdgcn_readfirstlane((int)task.nops);
        task.first_op = (unsigned)__builtin_amdgcn_readfirstlane((int)task.first_op);
        unsigned *state = sync + CTB_DONE + prm.ntasks + t;
        __hip_atomic_store(state, 0x100u | (blockIdx.x << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- the CTB's own samples -> LDS.  They are what earlier launches left (inter prediction, residuals of inter blocks): nothing of
        //      this launch writes them but this task, so they can travel while the neighbours are still at work
        for (int p = 0; p < 3; p++) {
            TileGeom &g = tg[p];
            g.x0 = task.cx * g.w; g.y0 = task.cy * g.h;
            const int pw = planes.width[p], ph = planes.height[p], pstride = planes.stride[p];
            const unsigned char *src = planes.data[p];
            unsigned char *tile = tiles + g.off;
            const int dw_per_row = g.w * PXB / 4, total = g.h * dw_per_row;
            // eight requests in flight per lane (a load-store-load-store loop pays one memory latency per dword: measured, 30 us per CTB)
            for (int i0 = lane; i0 < total; i0 += 64 * 8) {
                unsigned v[8];
                int dstoff[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + 64 * u;
                    const int r = i / dw_per_row, dcol = i - r * dw_per_row;
                    const int y = g.y0 + r, xb = g.x0 * PXB + dcol * 4;              // byte column inside the plane row
                    const bool ok = i < total && y < ph && xb < pw * PXB;
                    dstoff[u] = ok ? (r + 1) * g.stride + 16 + dcol * 4 : -1;
                    v[u] = ok ? *reinterpret_cast<const unsigned *>(src + (size_t)y * pstride + xb) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (dstoff[u] >= 0) *reinterpret_cast<unsigned *>(tile + dstoff[u]) = v[u];
            }
        }
        // ---- wait for the neighbours this CTB reads from (smaller ticket numbers: their holders are running)
        bool waited = false;
        for (int d = 0; d < 4; d++) {
            const int dep = __builtin_amdgcn_readfirstlane(task.dep[d]);
            if (dep < 0) continue;
            unsigned spins = 0;
            while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sync[CTB_DONE + dep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kCtbSpinLimit) {                  // never on a healthy run; the picture is then wrong, and the home word says so
                    atomicOr(&sync[CTB_HOME], 0x80000000u);
                    break;
                }
            }
            waited = true;
        }
        if (waited) xcd_acquire();                              // forget what this CU's L1 holds of the neighbours' samples
        __hip_atomic_store(state, 0x200u | (blockIdx.x << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- the row above (with its above-right extension) and the column left: the neighbours' samples
        for (int p = 0; p < 3; p++) {
            const TileGeom &g = tg[p];
            const int pw = planes.width[p], ph = planes.height[p], pstride = planes.stride[p];
            const unsigned char *src = planes.data[p];
            unsigned char *tile = tiles + g.off;
            const int dw_top = (g.w + g.ext) * PXB / 4;
            if (g.y0 > 0)
                for (int dcol = lane; dcol < dw_top; dcol += 64) {
                    const int xb = g.x0 * PXB + dcol * 4;
                    if (xb < pw * PXB) *reinterpret_cast<unsigned *>(tile + 16 + dcol * 4) = *reinterpret_cast<const unsigned *>(src + (size_t)(g.y0 - 1) * pstride + xb);
                }
            if (g.x0 > 0)
                for (int r = lane; r < g.h + 1; r += 64) {
                    const int y = g.y0 - 1 + r;
                    if (y >= 0 && y < ph)
                        *reinterpret_cast<Pixel *>(tile + r * g.stride + 16 - PXB) = *reinterpret_cast<const Pixel *>(src + (size_t)y * pstride + (size_t)(g.x0 - 1) * PXB);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __hip_atomic_store(state, 0x300u | (blockIdx.x << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the kernels' plane view of the tiles: sample (x, y) of plane p at tile + (y - y0 + 1) * stride + 16 + (x - x0) * PXB
        PlaneSet lp;
        for (int p = 0; p < 3; p++) {
            lp.data[p] = tiles + tg[p].off + (ptrdiff_t)(1 - tg[p].y0) * tg[p].stride + 16 - (ptrdiff_t)tg[p].x0 * PXB;
            lp.stride[p] = tg[p].stride;
            lp.width[p] = planes.width[p]; lp.height[p] = planes.height[p];
        }
        // ---- the CTB's operations in decoding order
        for (unsigned k = 0; k < task.nops; k++) {
            const unsigned op = (unsigned)__builtin_amdgcn_readfirstlane((int)ops[task.first_op + k]);
            const unsigned idx = op & 0x1ffffffu;
            if (!(op >> 31)) {
                intra_body<Pixel, true>(ish, lane, lp, intra_jobs[idx], bd, cips);
            } else {
                // the residual of the block just predicted: the same bodies the batched launches run (one block, this wave), on the LDS view
                const int log2 = (int)((op >> 29) & 3u) + 2, kind = (int)((op >> 25) & 15u);
                tu_dispatch<Pixel>(tu_lds, 0, lp, tu_jobs + idx, 1, log2, kind, coeffs, bd);
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
                    *reinterpret_cast<unsigned *>(dst + (size_t)y * pstride + xb) = *reinterpret_cast<const unsigned *>(tile + (r + 1) * g.stride + 16 + dcol * 4);
            }
        }
        xcd_release();                                          // this wave's stores sit in the XCD's L2 ...
        __builtin_amdgcn_wave_barrier();
        __hip_atomic_store(&sync[CTB_DONE + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                    // ... before the flag says so
        __hip_atomic_store(state, 0x500u | (blockIdx.x << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    OHEVC_REQUIRE(OHEVC_BIT_DEPTH_OK(bit_depth), "bit_depth must be 8..12 or 14");
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
