/*
 * oracle/synth_hooks.h -- TEST INFRASTRUCTURE ONLY (stream synthesiser).
 *
 * Force-included (gcc -include) when the reference's libavcodec/hevc.c and libavcodec/hevc_cabac.c are compiled, in
 * place and unmodified, for _ref/libopenhevc_gen.so.  It takes the include guard of the reference's
 * cabac_functions.h so that those two translation units get OUR five primitives instead of the arithmetic *decoder*:
 * every bin the reference's slice-data parser asks for is drawn from a seeded random source and, at the same time,
 * arithmetic-ENCODED (synth_gen.c).  The parser itself stays the reference's, so the bin sequence is legal syntax by
 * construction and the resulting bytes decode, on the untouched decoder, to the same syntax elements.
 */
#ifndef OH_SYNTH_HOOKS_H
#define OH_SYNTH_HOOKS_H

#include <stdint.h>
#include "libavcodec/cabac.h"

#define AVCODEC_CABAC_FUNCTIONS_H          /* keeps libavcodec/cabac_functions.h (:27-198) out */

int            ohsyn_bin(CABACContext *c, uint8_t *state);           /* replaces get_cabac            (:125) */
int            ohsyn_bypass(CABACContext *c);                        /* replaces get_cabac_bypass     (:130) */
int            ohsyn_terminate(CABACContext *c);                     /* replaces get_cabac_terminate  (:168) */
const uint8_t *ohsyn_skip_bytes(CABACContext *c, int n);             /* replaces skip_bytes           (:182) */
void           ohsyn_init_decoder(CABACContext *c, const uint8_t *buf, int buf_size);

static inline int get_cabac(CABACContext *c, uint8_t *const state) { return ohsyn_bin(c, state); }
static inline int get_cabac_bypass(CABACContext *c) { return ohsyn_bypass(c); }
/* reference semantics (:148-161): the bin selects between +val (1) and -val (0) */
static inline int get_cabac_bypass_sign(CABACContext *c, int val) { return ohsyn_bypass(c) ? val : -val; }
static inline int get_cabac_terminate(CABACContext *c) { return ohsyn_terminate(c); }
static inline const uint8_t *skip_bytes(CABACContext *c, int n) { return ohsyn_skip_bytes(c, n); }

#define ff_init_cabac_decoder ohsyn_init_decoder

#ifdef OHSYN_RENAME_CALLERS
/* hevc.c only: the two terminate-coded flags whose value the synthesiser must decide itself
 * (end_of_slice_segment_flag hevc.c:2582, pcm_flag hevc.c:2415) */
#define ff_hevc_end_of_slice_flag_decode ohsyn_end_of_slice_flag
#define ff_hevc_pcm_flag_decode          ohsyn_pcm_flag
/* SHVC enhancement layers: a motion vector into the inter-layer reference picture must be zero (H.265 F.8.5.3 / H.8.1.4; the reference
 * only resamples the CTBs a ZERO vector reaches, hevc.c:2077-2099, hevc_filter.c:1377-1430).  All its predictors are zero by construction;
 * these two wrappers make the coded difference zero as well (hevc.c:2027,2034,2046,2056). */
#define ff_hevc_ref_idx_lx_decode        ohsyn_ref_idx_lx
#define ff_hevc_hls_mvd_coding           ohsyn_mvd_coding
#endif

#endif /* OH_SYNTH_HOOKS_H */
