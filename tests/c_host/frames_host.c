/* frames_host.c -- a plain C host (no Python, no torch, no gloo) of the frame-parallel decoder: one process per rank, the reference's
 * decoder with the product's hooks (oracle/_ref/libopenhevc_hip*.so, decoder_harness.c API) and the NATIVE transport of
 * include/ohevc_frames.h between the processes.  What an application built on openHEVC's C API would do (libOpenHevcInit /
 * libOpenHevcDecode, gpac/modules/openhevc_dec/openHevcWrapper.c:47-155) after linking the drop-in.
 *
 *   frames_host <decoder .so> <product .so> <stream file> <rank> <world> <wire: rccl|sockets> <rendezvous> <device> <out file> [max bands]
 *
 * stream file : uint32 count, count x uint32 sizes, then the access units back to back (tests write it from tests/golden/streams.npz).
 * out file    : one line per picture THIS rank reconstructed: "<output position> <fnv64 of plane 0> <plane 1> <plane 2>", then a
 *               "stats ..." line of the transport's counters.  The test merges the ranks' files and compares with the single-process decoder.
 * Build: gcc frames_host.c -I../../include -ldl
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ohevc_frames.h"

#define FAIL(...) do { fprintf(stderr, "frames_host[%d]: ", rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

static uint64_t fnv64(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv)
{
    int rank = argc > 4 ? atoi(argv[4]) : 0;
    if (argc != 10 && argc != 11) FAIL("usage: frames_host <decoder.so> <product.so> <stream> <rank> <world> <rccl|sockets> <rendezvous> <device> <out> [max bands]");
    const int world = atoi(argv[5]), device = atoi(argv[8]);
    const int wire = strcmp(argv[6], "rccl") == 0 ? OHEVC_FRAMES_WIRE_RCCL : OHEVC_FRAMES_WIRE_SOCKETS;

    /* the product first (the decoder's DT_NEEDED names the same file: one loaded instance), then the decoder */
    void *prod = dlopen(argv[2], RTLD_NOW | RTLD_GLOBAL);
    if (!prod) FAIL("dlopen %s: %s", argv[2], dlerror());
    void *dec = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!dec) FAIL("dlopen %s: %s", argv[1], dlerror());
#define SYM(lib, name) name##_fn name = (name##_fn)dlsym(lib, #name); if (!name) FAIL("missing symbol %s", #name)
    typedef int (*ohevc_frames_transport_create_fn)(ohevc_frames_transport **, int, int, int, int, const char *, int);
    typedef const ohhip_frames_mode *(*ohevc_frames_transport_mode_fn)(ohevc_frames_transport *);
    typedef int (*ohevc_frames_transport_finish_fn)(ohevc_frames_transport *);
    typedef int (*ohevc_frames_transport_set_bands_fn)(ohevc_frames_transport *, int);
    typedef void (*ohevc_frames_transport_destroy_fn)(ohevc_frames_transport *);
    typedef int (*ohevc_frames_transport_stats_fn)(ohevc_frames_transport *, ohevc_frames_stats *);
    typedef const char *(*ohevc_last_error_fn)(void);
    typedef void *(*ohdec_open_ex_fn)(int, int, int);
    typedef int (*ohdec_frames_mode_fn)(void *, const void *);
    typedef int (*ohdec_decode_fn)(void *, const uint8_t *, int, int64_t);
    typedef int (*ohdec_flush_fn)(void *);
    typedef int (*ohdec_frame_is_local_fn)(void *);
    typedef int (*ohdec_frame_info_fn)(void *, int *, int *, int *, int *, int *);
    typedef int (*ohdec_frame_copy_fn)(void *, int, uint8_t *);
    typedef void (*ohdec_close_fn)(void *);
    SYM(prod, ohevc_frames_transport_create); SYM(prod, ohevc_frames_transport_mode); SYM(prod, ohevc_frames_transport_finish);
    SYM(prod, ohevc_frames_transport_set_bands); SYM(prod, ohevc_frames_transport_destroy); SYM(prod, ohevc_frames_transport_stats); SYM(prod, ohevc_last_error);
    SYM(dec, ohdec_open_ex); SYM(dec, ohdec_frames_mode); SYM(dec, ohdec_decode); SYM(dec, ohdec_flush); SYM(dec, ohdec_frame_is_local);
    SYM(dec, ohdec_frame_info); SYM(dec, ohdec_frame_copy); SYM(dec, ohdec_close);

    FILE *f = fopen(argv[3], "rb");
    if (!f) FAIL("cannot open %s", argv[3]);
    uint32_t count = 0;
    if (fread(&count, 4, 1, f) != 1 || count == 0 || count > 100000) FAIL("bad stream file");
    uint32_t *sizes = malloc(4 * (size_t)count);
    if (fread(sizes, 4, count, f) != count) FAIL("bad stream file");

    if (device >= 0) { char buf[16]; snprintf(buf, sizeof(buf), "%d", device); setenv("OHHIP_DEVICE", buf, 1); }
    ohevc_frames_transport *t = NULL;
    if (ohevc_frames_transport_create(&t, rank, world, device < 0 ? 0 : device, wire, argv[7], 60) != OHEVC_OK) FAIL("transport: %s", ohevc_last_error());
    if (argc == 11 && ohevc_frames_transport_set_bands(t, atoi(argv[10])) != OHEVC_OK) FAIL("set_bands: %s", ohevc_last_error());
    void *d = ohdec_open_ex(1, 1, 0);
    if (!d) FAIL("ohdec_open failed");
    if (world > 1 && ohdec_frames_mode(d, ohevc_frames_transport_mode(t)) != 0) FAIL("decoder has no frames mode");

    FILE *out = fopen(argv[9], "w");
    if (!out) FAIL("cannot write %s", argv[9]);
    int pos = 0, rc = 0;
    uint8_t *plane = NULL;
    size_t plane_cap = 0;
    for (uint32_t i = 0; i <= count && rc >= 0; i++) {
        int got;
        if (i < count) {
            uint8_t *au = malloc(sizes[i]);
            if (fread(au, 1, sizes[i], f) != sizes[i]) FAIL("short stream file");
            got = ohdec_decode(d, au, (int)sizes[i], (int64_t)i + 1);
            free(au);
        } else {
            got = ohdec_flush(d);
            if (got > 0) i--;                                   /* keep draining */
        }
        if (got < 0) { rc = got; break; }
        if (got > 0) {
            int w, h, bd, cw, ch;
            if (ohdec_frame_info(d, &w, &h, &bd, &cw, &ch) != 0) FAIL("no frame");
            if (ohdec_frame_is_local(d)) {
                uint64_t hsh[3];
                for (int c = 0; c < 3; c++) {
                    const int pw = c ? -((-w) >> cw) : w, ph = c ? -((-h) >> ch) : h;
                    const size_t n = (size_t)pw * ph * (bd > 8 ? 2 : 1);
                    if (n > plane_cap) { plane = realloc(plane, n); plane_cap = n; }
                    ohdec_frame_copy(d, c, plane);
                    hsh[c] = fnv64(plane, n);
                }
                fprintf(out, "%d %016llx %016llx %016llx\n", pos, (unsigned long long)hsh[0], (unsigned long long)hsh[1], (unsigned long long)hsh[2]);
            }
            pos++;
        }
    }
    if (rc < 0) fprintf(stderr, "frames_host[%d]: decode error %d: %s\n", rank, rc, ohevc_last_error());
    const int fin = ohevc_frames_transport_finish(t);
    ohevc_frames_stats st;
    ohevc_frames_transport_stats(t, &st);
    fprintf(out, "stats pictures %d published %lld subscribed %lld awaited_motion %lld awaited_planes %lld released %lld failed %lld bytes %lld wire_ranks %lld bands_imported %lld\n", pos, st.published,
            st.subscribed, st.awaited_motion, st.awaited_planes, st.released, st.failed, st.bytes, st.wire_ranks, st.bands_imported);
    fclose(out);
    if (world > 1) ohdec_frames_mode(d, NULL);
    ohdec_close(d);
    ohevc_frames_transport_destroy(t);
    free(plane); free(sizes); fclose(f);
    return rc < 0 ? 2 : fin != OHEVC_OK ? 3 : 0;
}
