/* ohevc_debug.h -- tuning/A-B switches of libohevc_hip.so.  NOT part of the drop-in boundary: nothing a host
 * integration needs lives here; bench/profiling scripts use it to compare kernel variants inside one process. */
#ifndef OHEVC_DEBUG_H
#define OHEVC_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Selects the variant of the 8x8..32x32 IDCT+add kernel used by ohevc_dev_tu_batch (results are identical):
 * bit 0: prefetch the prediction row before the transform; bit 1: read coefficients straight from HBM as int16
 * columns instead of staging 16-byte chunks through LDS; bit 2: persistent waves that prefetch the next blocks'
 * coefficients into registers while transforming the current ones; bit 3: cap that kernel at 5 waves/SIMD; bit 4:
 * LDS-transposed epilogue (full 64/128-byte row segments per wave instruction).  Bits 5/6 select ABLATION kernels that
 * keep the memory traffic but drop the transform (bit 5: also drop the LDS passes) -- their output is NOT correct;
 * they exist only to locate the bottleneck.  Returns the previous value; -1 restores the shipped default (coalesced epilogue for 16x16/32x32, prefetch for 8x8). */
int ohevc_debug_set_tu_variant(int variant);
/* 1 when the library is the LAB build (make -C openhevc_amd/csrc LAB=1 -> libohevc_hip_lab.so): the product library carries only the
 * shipped kernels and their direct alternatives -- dot2 forms 0 / 1 / 16 / 16+128 / 16+128+1024 (bit 10: non-temporal coefficient
 * loads) and, for 32x32, bit 11 (2048): the matrix-core tile kernel, one workgroup per 8 blocks (shipped).  Everything else below
 * (persistent / software-pipelined forms, ablations, probes, the loop forms of the tile kernel: bits 2, 5, 6, 8, 9, 12-15 with 11) exists
 * in the lab build only; the product library ignores those bits. */
int ohevc_debug_has_lab(void);
/* bit 8 of the variant (256): 32x32 blocks run the matrix-core form (tu_idct32_mfma_kernel: v_mfma_i32_32x32x32_i8 on the high and low
 * byte planes of the int16 inputs, both passes in registers); its grid is the persistent kernel's (ohevc_debug_set_tu_pipe_workgroups).
 * ohevc_debug_mfma_i8_probe runs one such MFMA per probe on raw lane data (a, b: nprobes x 64 lanes x 16 bytes, d: nprobes x 64 x 16
 * int32, all device memory): tools/probe_mfma_layout.py recovers the fragment layout the kernel relies on from it. */
int ohevc_debug_mfma_i8_probe(const void *a, const void *b, void *d, int nprobes, void *stream);
/* variant bits 9 / 10 (with bit 8): fetch the prediction rows before the transform / no register prefetch of the next block pair.
 * ohevc_debug_lds_tr16_probe: one ds_read_b64_tr_b16 per lane at byte address addr[probe * 64 + lane] of a 2048-byte LDS image holding
 * int16 i at index i; out: nprobes x 64 lanes x 8 bytes (device memory). */
int ohevc_debug_lds_tr16_probe(const void *addr, void *out, int nprobes, void *stream);
/* bit 2 of the variant selects the persistent, software-pipelined kernel; this sets its grid size (workgroups). */
int ohevc_debug_set_tu_pipe_workgroups(int n);
/* motion-compensation kernel: 1 = first scalar kernel, 2 = packed-pair dot-product kernel, 3 = LDS tiles with both reference windows
 * staged before the first barrier (mc3), 4 = the matrix-core form (mc4) for tile batches and mc3's
 * four-jobs-per-wavefront form for the small-block entry point, 5 = mc4 for both, 6 (shipped since round 3) = mc4 for tile batches and mc4q
 * - four blocks of at most 8x8 per matrix-core tile - for the small-block entry point.  3, 4 and 5 hand tiles with reference samples above
 * the bit depth's range to the exact redo kernel; 1 and 2 are exact for samples that fit the bit depth.  Env OHEVC_MC_VARIANT sets
 * the initial value (A/B of whole-decoder runs).  Lab build only (results are NOT pictures): 102 / 103 = mc4q_kernel's traffic-only /
 * arithmetic-only twin, 104 = the kernel itself again. */
int ohevc_debug_set_mc_variant(int variant);
/* SAO kernel: 0 = shipped (wide form: 16 bytes of one row per lane, no LDS, position rules as byte masks; blocks it cannot take fall
 * back to the LDS-window form); bit 1 = never take the wide form (A/B); bit 0 = the edge classes split a block into its interior (short form: no border / restore predicate can
 * apply there) and its outer ring (full form), enumerated so that whole wavefronts take one form.  Same results (CPU emulation and
 * tests).  Which XCD takes which block of the list: default = runs of 16 consecutive list entries per XCD; bit 2 (4) = list order (workgroup
 * number = list index, i.e. round-robin over the XCDs); bit 3 (8) = one contiguous eighth of the list per XCD (profiles/r03l_*, r03m_*);
 * bit 4 (16) = interior edge-class blocks through the general loop too instead of the short form sao_edge_plain (round 5). */
int ohevc_debug_set_sao_variant(int variant);
/* Band SAO events on samples ABOVE the bit depth's range since the last reset (all streams of the current device; waits for them).  The
 * reference's sao_band_filter reads past its 32-entry offset table for such a sample (hevcdsp_template.c:340-365; constrained intra
 * prediction above 8 bit produces them): its output for that stream is whatever lay on its stack.  The kernels wrap the band index.
 * 0 = every sample the stream put through the band filter was in range: the stream is comparable with the reference.  -1 on error. */
long ohevc_debug_sao_band_above_range(int reset);
/* ctx executor for the intra-coded blocks of a picture: 2 (shipped) = ONE ohevc_dev_ctbs launch, coding-tree blocks as tasks with their
 * samples in LDS (pictures whose intra jobs name no CTB size fall back to 0); 0 issues one prediction launch and one residual launch
 * per dependency level; 1 runs all levels of a picture inside one ohevc_dev_levels launch (persistent ticketed workgroups on one XCD, in-kernel
 * step barriers).  Same results.  Measured on MI355X with the real decoder (1080p, profiles/r01n_level_executor_ab.txt):
 * a step costs about as much as a kernel boundary (both are a chain of L2 round trips), so mode 1 only saves host-side
 * launch work and is currently the slower one.  Returns the old mode. */
int ohevc_debug_set_level_launch(int mode);
/* Profiling aid for the HOST side only: contexts created while this is on need no device and produce NO pixels -- every
 * ohevc_rec_* call does its normal work, the frame-end executor just drops the recorded jobs.  It exists to time the
 * recording cost of the table slots (tools/profile_recording.py); never a fallback: pictures stay unwritten. */
/* 1 (default): in the levels form a block's residual rides with its intra job (ohevc_dev_intra_recon_batch: one launch per level); 0: two
 * launches per level.  Env OHEVC_FUSE_INTRA sets the initial value.  Returns the previous setting. */
int ohevc_debug_set_fuse_intra(int on);
long ohevc_debug_parked_total(void);          /* frame ends parked so far by ohevc_frame_end_deferred, process-wide */
int ohevc_debug_set_park_frames(int on);     /* process default of OHEVC_OPT_PARK_FRAMES (0); returns the previous value */
int ohevc_debug_set_record_only(int on);      /* 2: record the forms a context WITH a device records (filter maps instead of per-edge jobs) and drop them */
/* 1 (default): the intra blocks of a dependency level go through the packed kernel (ohevc_dev_intra_recon_sorted: N lanes per block,
 * residual added in registers); 0: one wavefront per block (ohevc_dev_intra_recon_batch).  Pictures with constrained intra prediction
 * always take the latter.  Returns the previous setting.  Environment: OHEVC_INTRA_PACK. */
int ohevc_debug_set_intra_pack(int on);
/* 1: runs of consecutive narrow dependency levels (at most 8 wavefronts each) go out as one ohevc_dev_intra_chain launch (measured: fewer launches,
 * not faster - DESIGN.md 5g; default 0);
 * 0: one launch per level.  Environment: OHEVC_INTRA_CHAIN. */
int ohevc_debug_set_intra_chain(int on);
/* A/B of ohevc_tables_derive_filters: 1 (default) the deblocking maps travel and the device derives the edges (ohevc_dev_deblock_maps);
 * 0 the host derives one job per edge (the only form record-only contexts and the filter-lag emulation of 16x16 CTBs have).
 * Returns the previous setting. */
int ohevc_debug_set_filters_on_device(int on);
/* 1 (default): the boundary strengths are derived on the device as well (ohevc_dev_boundary_strengths; the front end records its
 * ff_hevc_deblocking_boundary_strengths calls instead of making them); 0: the reference's own host arrays travel.  Environment: OHEVC_DEVICE_BS. */
int ohevc_debug_set_bs_on_device(int on);

/* Inspection of a record-only context, for HOST-LOGIC TESTS without a GPU: before a record-only context drops what was
 * recorded, it hands itself to this callback -- stage 0 from ohevc_frame_reconstruct (motion compensation, residual and intra
 * levels), stage 1 from ohevc_frame_end (in-loop filters) -- and the accessors below expose the job arrays exactly as the
 * executor would stage them (slice-thread recorders already merged).  tests/ uses it to run the recorded jobs through the CPU
 * oracle on the decoder's own host frames (oracle/sw_exec.c) and compare whole pictures with the untouched decoder: recording
 * slots, pointer registry, dependency levels, filter flags.  The product never installs a sink and computes nothing on the CPU. */
struct ohevc_ctx;
typedef void (*ohevc_debug_sink)(void *user, struct ohevc_ctx *ctx, int stage);
void ohevc_debug_set_frame_sink(ohevc_debug_sink fn, void *user);
int ohevc_debug_target(struct ohevc_ctx *ctx, int *slot, int *width, int *height, int *chroma_format_idc, int *bit_depth);
int ohevc_debug_mc(struct ohevc_ctx *ctx, int small, const struct ohevc_mc_job **jobs, int *n);
int ohevc_debug_level_count(struct ohevc_ctx *ctx);
int ohevc_debug_level_intra(struct ohevc_ctx *ctx, int level, const struct ohevc_intra_job **jobs, int *n);
int ohevc_debug_level_tu(struct ohevc_ctx *ctx, int level, int log2_size, int kind, const struct ohevc_tu_job **jobs, int *n);
/* the intra work of the CTB executor (level launch mode 2): tasks in raster order, their operation words (ohevc_dev_ctbs) and the job arrays they index */
int ohevc_debug_ctbs(struct ohevc_ctx *ctx, const struct ohevc_ctb_task **tasks, int *ntasks, const uint32_t **ops, const struct ohevc_intra_job **intra_jobs,
                     const struct ohevc_tu_job **tu_jobs, int *log2_ctb_size);
/* 1: the jobs of every dependency level are staged back to front (the kernel emulator runs the workgroups of a launch one after the other in
 * launch order: an order that hides a dependency the level computation missed; the jobs of a level are independent, so any order must do) */
int ohevc_debug_set_reverse_levels(int on);
/* a frame with at least this many recorded dependency levels is issued on the context's long-chain stream (highest stream priority: a hardware
 * queue pool of its own); 0: never (rounds 1-4).  Default 96. */
int ohevc_debug_set_long_chain_levels(int levels);
/* 2 (default): the long-chain streams of a store's contexts alternate between the highest and the lowest stream priority (two hardware-queue
 * pools); 1: the highest only */
int ohevc_debug_set_long_chain_pools(int n);
/* 0: every recorded coefficient block crosses the bus whole, as in rounds 1-4 (A/B of the compact upload; default 1) */
int ohevc_debug_set_compact_coeffs(int on);
int ohevc_debug_arena(struct ohevc_ctx *ctx, const int16_t **coeffs, const struct ohevc_intra_cip **cips);
int ohevc_debug_filters(struct ohevc_ctx *ctx, const struct ohevc_dbk_job **vertical, int *n_vertical, const struct ohevc_dbk_job **horizontal,
                        int *n_horizontal, const struct ohevc_sao_job **sao, int *n_sao, struct ohevc_sao_bypass *bypass /* HOST map */);
/* block until the frame that reconstructs picture `slot` has ended (frame threads: another context of the store) */
int ohevc_debug_wait_picture(struct ohevc_ctx *ctx, int slot);
/* Where the intra chain kernel (ohevc_dev_intra_chain) spends a level: on != 0 switches the counters on (they add up over all launches of
 * the process); out (may be NULL) receives {clocks waiting for stores + barrier, clocks issuing the level's loads and prefetches, clocks in
 * the level's arithmetic incl. the wait for its samples, clocks in further passes of wide levels, levels} of wavefront 0; on == 0 frees them.
 * Shader clock (s_memtime).  Diagnosis only. */
int ohevc_debug_intra_chain_clocks(int on, unsigned long long out[64]);      /* [5..7]: inside the issue phase - after the sample loads, after the residual prefetch, after the level record; round 6, every wavefront: [8 + s] clocks in the arithmetic of steps whose blocks are of size class s (4x4 .. 32x32), [12 + s] of which waiting for the five samples, [16 + s] such steps; [20 + n] levels of n wavefronts (n = 16: more); [40 + s] clocks of the SLOWEST wavefront's arithmetic being of class s is not recorded - see tools/diag_chain_clocks.py */
int ohevc_debug_set_chain_handover(int mode);   /* hand-over between two levels of the intra chain kernel: 0 the workgroup barrier alone (default since round 6), 2 wait for the level's stores first (rounds 4-5), 1 / 3 agent-scope acquire behind it (rounds 2-3); + 4: a slot's job from the level records instead of the prologue's descriptors (A/B); returns the previous mode */
/* SHVC up-sampling kernel: 0 = the tile form (shipped: a workgroup per 64 x 32 output tile, both passes through LDS, dot instructions),
 * 1 = the round-2 strip form (a thread per column strip).  Returns the previous value.  Environment: OHEVC_UPSAMPLE_VARIANT. */
int ohevc_debug_set_upsample_variant(int variant);

/* Device pictures come in batches (ctx.hip: PicStore::spare; OHEVC_PICTURE_BATCH=0 switches it off): how many batch allocations the picture
 * store of `ctx` has made so far.  A test hook: two picture sizes taking turns (the two layers of an SHVC stream share a store) must not make
 * one batch per picture. */
int ohevc_debug_picture_batches(struct ohevc_ctx *ctx);
/* Deblocking from the maps (ohevc_dev_deblock_maps): 0 = a lane per 4-line luma segment with packed 16-bit arithmetic (shipped, up to 10 bit),
 * 1 = a lane per line (rounds 2-3; what deeper pictures and unaligned planes take anyway).  Returns the previous value.
 * Environment: OHEVC_DEBLOCK_VARIANT. */
int ohevc_debug_set_deblock_variant(int variant);
long long ohevc_debug_deblock_segment_launches(void);       /* launches that took form 0 so far (tests: which form ran) */
/* The file-system rendezvous of the native transport's RCCL wire (ohevc_frames.h) without RCCL: rank 0 hands the 128 bytes in `id` to the
 * other ranks (who receive them in `id`), through `path`, with the nonce handshake that keeps a file of an earlier run from being accepted.
 * Tests only. */
int ohevc_debug_frames_rendezvous(const char *path, int rank, int world, int timeout_s, unsigned char id[128]);

#ifdef __cplusplus
}
#endif
#endif
