/*
 * ref_sse_shim.c -- TEST/BENCH INFRASTRUCTURE ONLY.  Call-through into the REFERENCE's x86 SSE4
 * intrinsics kernels (libavcodec/x86/hevc_idct_sse.c:504-745,891-916), compiled unmodified from
 * /root/reference by oracle/Makefile into oracle/_ref/libhevcref_sse.so.  Used only as the
 * "reference SIMD path" CPU baseline timed beside the GPU number (bench.py cpu_baseline) and to
 * cross-check C == SSE.  The yasm deblock kernels cannot be assembled here (no yasm) -- not included.
 */
#include <pthread.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>
#include "libavutil/mem.h"
#include "libavcodec/x86/hevcdsp.h"

typedef void (*idct_fn)(int16_t *coeffs, int col_limit);
typedef void (*add_fn)(uint8_t *dst, int16_t *coeffs, ptrdiff_t stride);

static idct_fn pick_idct(int bd, int log2)
{
    static const idct_fn t8[4]  = { ff_hevc_transform_4x4_8_sse4,  ff_hevc_transform_8x8_8_sse4,  ff_hevc_transform_16x16_8_sse4,  ff_hevc_transform_32x32_8_sse4 };
    static const idct_fn t10[4] = { ff_hevc_transform_4x4_10_sse4, ff_hevc_transform_8x8_10_sse4, ff_hevc_transform_16x16_10_sse4, ff_hevc_transform_32x32_10_sse4 };
    return bd == 8 ? t8[log2 - 2] : bd == 10 ? t10[log2 - 2] : NULL;
}
static add_fn pick_add(int bd, int log2)
{
    static const add_fn t8[4]  = { ff_hevc_transform_4x4_add_8_sse4,  ff_hevc_transform_8x8_add_8_sse4,  ff_hevc_transform_16x16_add_8_sse4,  ff_hevc_transform_32x32_add_8_sse4 };
    static const add_fn t10[4] = { ff_hevc_transform_4x4_add_10_sse4, ff_hevc_transform_8x8_add_10_sse4, ff_hevc_transform_16x16_add_10_sse4, ff_hevc_transform_32x32_add_10_sse4 };
    return bd == 8 ? t8[log2 - 2] : bd == 10 ? t10[log2 - 2] : NULL;
}

int ohsse_available(void) { return 1; }

/* IDCT + add of n dense blocks, same contract as ohref_tu_batch(kind = OH_TU_IDCT) */
int ohsse_idct_add_batch(int bd, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                         ptrdiff_t stride, const int32_t *xy)
{
    idct_fn f = pick_idct(bd, log2);
    add_fn  g = pick_add(bd, log2);
    int nn = 1 << (2 * log2), ps = bd > 8 ? 2 : 1;
    DECLARE_ALIGNED(32, int16_t, tmp)[32 * 32];
    if (!f || !g) return -1;
    for (int i = 0; i < n; i++) {
        memcpy(tmp, coeffs + (size_t)i * nn, nn * sizeof(int16_t));
        f(tmp, 1 << log2);
        g(plane + (ptrdiff_t)xy[2 * i + 1] * stride + xy[2 * i] * ps, tmp, stride);
    }
    return 0;
}

struct mt_arg { int bd, log2, n; const int16_t *coeffs; uint8_t *plane; ptrdiff_t stride; const int32_t *xy; };
static void *mt_worker(void *p)
{
    struct mt_arg *a = p;
    ohsse_idct_add_batch(a->bd, a->log2, a->n, a->coeffs, a->plane, a->stride, a->xy);
    return NULL;
}
int ohsse_idct_add_batch_mt(int bd, int log2, int n, const int16_t *coeffs, uint8_t *plane,
                            ptrdiff_t stride, const int32_t *xy, int threads)
{
    pthread_t th[256];
    struct mt_arg a[256];
    int nn = 1 << (2 * log2);
    if (!pick_idct(bd, log2)) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    for (int t = 0; t < threads; t++) {
        int lo = (int)((long long)n * t / threads), hi = (int)((long long)n * (t + 1) / threads);
        a[t] = (struct mt_arg){ bd, log2, hi - lo, coeffs + (size_t)lo * nn, plane, stride, xy + 2 * lo };
        pthread_create(&th[t], NULL, mt_worker, &a[t]);
    }
    for (int t = 0; t < threads; t++)
        pthread_join(th[t], NULL);
    return 0;
}
