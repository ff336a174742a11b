/*
 * oracle/sse_stubs.c -- TEST / BASELINE INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * What it takes to link the reference decoder AS SHIPPED ON x86 - ARCH_X86 1, the SSE2 / SSSE3 / SSE4.1 / SSE4.2 intrinsics of
 * libavcodec/x86/hevc_{idct,mc,sao,intra_pred,il_pred}_sse.c wired in by libavcodec/x86/hevcdsp_init.c:403-640 and
 * x86/hevcpred_init.c:31-41 - in an image that has no yasm: the link is short by exactly the symbols the .asm files
 * would have provided.
 *
 *   - eight HEVC deblocking functions (x86/hevc_deblock.asm; assigned at x86/hevcdsp_init.c:441-442,471-472,533-534,556-557).  They are
 *     forwarded to the reference's own C implementation of the same slots (hevcdsp_template.c:1629-1787): hevcdsp.c is compiled a second
 *     time with ARCH_X86 0 and its init symbol renamed (ohx86_c_dsp_init), and the forwarders call through that table.  So: in
 *     libopenhevc_sse.so DEBLOCKING RUNS AS C.  Every report that quotes this build says so.
 *   - nine symbols of code an HEVC stream never reaches (half-pel / quarter-pel DSP of other codecs, FFT / MDCT / DCT-32 of the audio
 *     decoders, the deinterlacer): empty init functions, and bodies that abort if anything ever calls them.
 *
 * Compiled against the reference's headers (like ref_shim.c); no reference source is copied.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <stddef.h>

#include "libavcodec/hevcdsp.h"

/* hevcdsp.c, ARCH_X86 0, -Dff_hevc_dsp_init=ohx86_c_dsp_init (oracle/Makefile) */
void ohx86_c_dsp_init(HEVCDSPContext *c, int bit_depth);

static HEVCDSPContext c8, c10;
static void __attribute__((constructor)) fill_c_tables(void)
{
    ohx86_c_dsp_init(&c8, 8);
    ohx86_c_dsp_init(&c10, 10);
}

#define LUMA(dir, depth, tab)                                                                                                  \
    void ff_hevc_##dir##_loop_filter_luma_##depth##_ssse3(uint8_t *pix, ptrdiff_t stride, int *beta, int *tc, uint8_t *no_p,   \
                                                          uint8_t *no_q)                                                       \
    {                                                                                                                          \
        tab.hevc_##dir##_loop_filter_luma(pix, stride, beta, tc, no_p, no_q);                                                  \
    }
#define CHROMA(dir, depth, tab)                                                                                                \
    void ff_hevc_##dir##_loop_filter_chroma_##depth##_sse2(uint8_t *pix, ptrdiff_t stride, int *tc, uint8_t *no_p, uint8_t *no_q) \
    {                                                                                                                          \
        tab.hevc_##dir##_loop_filter_chroma(pix, stride, tc, no_p, no_q);                                                      \
    }
LUMA(h, 8, c8) LUMA(v, 8, c8) LUMA(h, 10, c10) LUMA(v, 10, c10)
CHROMA(h, 8, c8) CHROMA(v, 8, c8) CHROMA(h, 10, c10) CHROMA(v, 10, c10)

/* ---- never reached by an HEVC stream ---- */
static void unreachable(const char *what)
{
    fprintf(stderr, "oracle/sse_stubs.c: %s was called - the SSE baseline build has no yasm objects\n", what);
    abort();
}
void ff_hpeldsp_init_x86(void *c, int flags) { (void)c; (void)flags; }       /* hpeldsp.c:369: keeps the C table */
void ff_qpeldsp_init_x86(void *c) { (void)c; }                               /* qpeldsp.c:811 */
void ff_fft_permute_sse(void *s, void *z) { (void)s; (void)z; unreachable("ff_fft_permute_sse"); }
void ff_fft_calc_sse(void *s, void *z) { (void)s; (void)z; unreachable("ff_fft_calc_sse"); }
void ff_imdct_calc_sse(void *s, float *o, const float *i) { (void)s; (void)o; (void)i; unreachable("ff_imdct_calc_sse"); }
void ff_imdct_half_sse(void *s, float *o, const float *i) { (void)s; (void)o; (void)i; unreachable("ff_imdct_half_sse"); }
void ff_dct32_float_sse2(float *o, const float *i) { (void)o; (void)i; unreachable("ff_dct32_float_sse2"); }
void ff_deinterlace_line_mmx(uint8_t *dst, const uint8_t *a, const uint8_t *b, const uint8_t *c, const uint8_t *d, const uint8_t *e, int size)
{
    (void)dst; (void)a; (void)b; (void)c; (void)d; (void)e; (void)size; unreachable("ff_deinterlace_line_mmx");
}
void ff_deinterlace_line_inplace_mmx(const uint8_t *a, const uint8_t *b, const uint8_t *c, const uint8_t *d, const uint8_t *e, int size)
{
    (void)a; (void)b; (void)c; (void)d; (void)e; (void)size; unreachable("ff_deinterlace_line_inplace_mmx");
}
