/* ohevc_debug.h -- tuning/A-B switches of libohevc_hip.so.  NOT part of the drop-in boundary: nothing a host
 * integration needs lives here; bench/profiling scripts use it to compare kernel variants inside one process. */
#ifndef OHEVC_DEBUG_H
#define OHEVC_DEBUG_H
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
/* bit 2 of the variant selects the persistent, software-pipelined kernel; this sets its grid size (workgroups). */
int ohevc_debug_set_tu_pipe_workgroups(int n);
/* motion-compensation kernel: 1 = first scalar kernel, 2 = packed-pair dot-product kernel, 3 = 2 + both reference
 * windows staged before the first barrier (shipped) */
int ohevc_debug_set_mc_variant(int variant);
/* ctx executor for intra dependency levels: 0 (shipped) issues one prediction launch and one residual launch per level;
 * 1 runs all levels of a picture inside one ohevc_dev_levels launch (persistent ticketed workgroups on one XCD, in-kernel
 * step barriers).  Same results.  Measured on MI355X with the real decoder (1080p, profiles/r01n_level_executor_ab.txt):
 * a step costs about as much as a kernel boundary (both are a chain of L2 round trips), so mode 1 only saves host-side
 * launch work and is currently the slower one.  Returns the old mode. */
int ohevc_debug_set_level_launch(int mode);
/* Profiling aid for the HOST side only: contexts created while this is on need no device and produce NO pixels -- every
 * ohevc_rec_* call does its normal work, the frame-end executor just drops the recorded jobs.  It exists to time the
 * recording cost of the table slots (tools/profile_recording.py); never a fallback: pictures stay unwritten. */
int ohevc_debug_set_record_only(int on);
#ifdef __cplusplus
}
#endif
#endif
