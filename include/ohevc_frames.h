/*
 * ohevc_frames.h -- frame-parallel decoding over several processes, one per GPU: the protocol's callback table and a
 * native transport for it.
 *
 * The reference's frame threads (pthread_frame.c:479-513) give every thread one picture, share the DPB in host memory and wait on the
 * rows of the pictures they predict from (hevc_await_progress, hevc.c:1951-1958) and on their motion fields (hevc_mvs.c).  Across
 * processes the owner of a picture - decoding-order index % world - parses its slice data and reconstructs it on its GPU; what later
 * pictures need from it travels to everyone: the sample planes and the motion field HEVCFrame.tab_mvf.  integration/hip_frames.h is the
 * decoder-side half (which hooks call the four callbacks when); this header is the application-side half:
 *
 *   ohhip_frames_mode          the callback table the hooks drive (any transport can fill it);
 *   ohevc_frames_transport_*   a transport in C inside libohevc_hip.so - no Python, no torch, no gloo - over one of two wires:
 *       OHEVC_FRAMES_WIRE_RCCL     ncclBroadcast over xGMI, one GPU per process: planes AND motion fields travel as device memory in one
 *                                  group call per picture on a stream of their own (librccl is loaded when the transport is created);
 *       OHEVC_FRAMES_WIRE_SOCKETS  TCP connections between the processes, host-staged: for machines (and tests) where the ranks
 *                                  share one GPU or have none - RCCL refuses two ranks on one device.
 *   Collectives are issued in decoding order on every rank (publish by the owner, subscribe by the others: exactly one of the two per
 *   exchanged picture), asynchronously; a rank blocks only in await_motion / await_planes, the two waits of the reference's frame threads.
 */
#ifndef OHEVC_FRAMES_H
#define OHEVC_FRAMES_H
#include <stddef.h>
#include "ohevc_ctx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ohhip_frames_mode {
    size_t struct_size;      /* sizeof(ohhip_frames_mode) as the caller was compiled: FIRST, checked by ohhip_backend_frames_mode (0 or larger than the
                              * library's: refused); fields beyond it are taken as NULL / 0; new fields are only ever appended */
    int rank, world;
    void *user;
    /* owner: picture `index` is complete (device work drained): planes in picture-store slot `slot` of ctx, motion field at mvf.
     * Must not keep the pointers after returning (copy or send synchronously).
     * failed != 0: the owner could not decode / reconstruct the picture.  The transport still issues the picture's collectives - every
     * rank issues exactly one publish or subscribe per exchanged picture, or the others' receives never complete - with an error mark
     * that makes await_motion / await_planes of the subscribers return nonzero at once (payload undefined; mvf may be NULL). */
    int (*publish)(void *user, int index, ohevc_ctx *ctx, int slot, const void *mvf, size_t mvf_bytes, int failed);
    /* everyone else: start receiving picture `index` from rank index % world; must not block */
    int (*subscribe)(void *user, int index, ohevc_ctx *ctx, int slot, size_t mvf_bytes);
    /* block until the motion field of remote picture `index` has arrived and copy it to mvf */
    int (*await_motion)(void *user, int index, void *mvf, size_t mvf_bytes);
    /* block until the planes of remote picture `index` have arrived and put them into picture-store slot `slot` (ohevc_pic_import) */
    int (*await_planes)(void *user, int index, ohevc_ctx *ctx, int slot);
    /* the decoder dropped the buffer of remote picture `index` (its DPB entry was recycled) or will never look at it again: wait for what
     * is still in flight for it and free the staging memory.  May be NULL. */
    int (*release)(void *user, int index);
    /* like await_planes, but only rows 0 .. last_luma_row of the picture have to be in the store when it returns (last_luma_row < 0 or beyond the
     * picture: all of it) - the wait of hevc_await_progress (hevc.c:1951-1958) for the rows a picture's motion vectors reach, at the granularity
     * of the transport's bands.  Calls for one picture may come with growing row numbers.  Returns < 0 on failure, 0 when the rows asked for
     * are in, 1 when that happened to complete the picture (the transport may forget it then: do not ask again).  May be NULL: the hooks then
     * use await_planes. */
    int (*await_rows)(void *user, int index, ohevc_ctx *ctx, int slot, int last_luma_row);
    /* 0: a picture's owner is its decoding-order index % world and exchanged pictures cross the wire (any stream).  1: ownership per IDR SEGMENT -
     * segment number % world, a segment = an IDR or BLA picture and everything up to the next one: nothing after such a picture predicts from
     * anything before it, so a segment needs nothing from the other ranks and NOTHING is exchanged; the ranks decode different segments at the
     * same time.  For streams with regular IDR / BLA pictures (closed GOPs, splices); CRA pictures do not open a segment (their RASL pictures
     * predict across them), so an open-GOP stream without IDR / BLA pictures stays on one rank: use per-picture ownership there. */
    int segment_ownership;
} ohhip_frames_mode;

enum { OHEVC_FRAMES_WIRE_RCCL = 0, OHEVC_FRAMES_WIRE_SOCKETS = 1 };

typedef struct ohevc_frames_transport ohevc_frames_transport;

/* rank / world: this process and the number of processes; device: the GPU this process decodes on (RCCL: one per rank).
 * rendezvous: RCCL: the path of a file (on storage all ranks see) through which rank 0 hands its ncclUniqueId to the others - every rank
 * first leaves a nonce in <path>.ready.<rank> and only accepts an id file that carries it, so a file left behind by a crashed run is never
 * mistaken for this run's; all of these files are removed once the communicator exists; sockets: "host:port" - all ranks run on `host`,
 * rank r listens on port + r of that address only, and a peer must present the run's token (OHEVC_FRAMES_TOKEN, or derived from this string).
 * timeout_s: how long a rank waits for its peers (rendezvous, a connection, a message) before it gives up with an error. */
int  ohevc_frames_transport_create(ohevc_frames_transport **out, int rank, int world, int device, int wire, const char *rendezvous, int timeout_s);
/* A picture crosses the wire as its motion field followed by bands of whole CTU rows (at most 8, the default; 1 = whole pictures): the owner
 * exports band b + 1 while band b is on the wire, a subscriber imports band b while band b + 1 arrives, and await_rows returns as soon as the
 * bands a dependent picture reaches are in.  Every rank must use the same value; only between pictures (nothing in flight). */
int  ohevc_frames_transport_set_bands(ohevc_frames_transport *t, int max_bands);
/* ohhip_frames_mode.segment_ownership of this transport's callback table (above); before the first picture, the same on every rank */
int  ohevc_frames_transport_set_ownership(ohevc_frames_transport *t, int per_idr_segment);
/* the callback table to hand to ohhip_set_frames_mode (valid until the transport is destroyed) */
const ohhip_frames_mode *ohevc_frames_transport_mode(ohevc_frames_transport *t);
/* every collective this rank issued has completed (call on all ranks after the last picture, before destroying) */
int  ohevc_frames_transport_finish(ohevc_frames_transport *t);
void ohevc_frames_transport_destroy(ohevc_frames_transport *t);
/* One picture through the wire, called by EVERY rank: `root` sends the picture in its src_slot and the mvf_bytes at mvf_in; every rank, the
 * root included, receives into its dst_slot (same geometry) and mvf_out.  Exactly the steps of publish on the owner and subscribe + await on
 * the others; a start-up check of the wire - and with world 1, where the decoder never exchanges a picture, what executes the RCCL calls
 * (ncclCommInitRank, one ncclGroup of four ncclBroadcast on device memory) on a single GPU (tests/test_dist_gpu.py). */
int  ohevc_frames_transport_selftest(ohevc_frames_transport *t, ohevc_ctx *ctx, int src_slot, int dst_slot, int root, const void *mvf_in, void *mvf_out,
                                     size_t mvf_bytes);
/* wire_ranks: the size of the communicator as the wire itself reports it - ncclCommCount of the RCCL communicator; 1 + the connected peers
 * of the sockets wire (not the `world` argument echoed back) */
typedef struct ohevc_frames_stats { long long published, subscribed, awaited_motion, awaited_planes, released, failed, bytes, wire_ranks, bands_imported; } ohevc_frames_stats;
int  ohevc_frames_transport_stats(ohevc_frames_transport *t, ohevc_frames_stats *out);

#ifdef __cplusplus
}
#endif
#endif
