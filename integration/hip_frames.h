/*
 * integration/hip_frames.h -- frame-parallel decoding over several processes, one per GPU.
 *
 * The reference's frame threads (pthread_frame.c) give every thread one picture and share the DPB in host memory; a thread waits
 * on the row progress of the pictures it predicts from (hevc_await_progress, hevc.c:1951-1958) and on their motion fields
 * (hevc_mvs.c).  Across processes the same structure becomes:
 *
 *   - every process reads the whole stream and parses every slice HEADER (DPB management, POC, reference picture sets and lists
 *     stay identical everywhere), but only the OWNER of a picture - decoding-order index % world - parses its slice data and
 *     reconstructs it; the others skip the slice data (avctx->execute / execute2, the application-owned dispatch points that
 *     hls_slice_data calls at hevc.c:3082-3089, are replaced by ohhip_frames_install);
 *   - what later pictures need from a picture travels from its owner to everyone: the sample planes (device memory,
 *     ohevc_pic_export / ohevc_pic_import; RCCL broadcast over xGMI) and the motion field HEVCFrame.tab_mvf (host memory), the two
 *     things the reference's frame threads share through the DPB;
 *   - a process waits for a remote picture's motion field when it starts a picture that may name it (ff_hevc_frame_rps) and for
 *     its planes right before it launches that picture's device work (frame end): the two waits of the reference's frame threads.
 *
 * The transport is the application's (openhevc_amd/dist.py: torch.distributed).  All processes must call the hooks with the same
 * stream; collectives are issued in decoding order on every process (publish by the owner, subscribe by the others, exactly one
 * of the two per exchanged picture).  Pictures nothing can reference (sub-layer non-reference pictures of the highest temporal
 * sub-layer) are not exchanged.  One decoding thread per process.
 */
#ifndef OHHIP_FRAMES_H
#define OHHIP_FRAMES_H
#include <stddef.h>
#include "ohevc_ctx.h"

typedef struct ohhip_frames_mode {
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
} ohhip_frames_mode;

/* switch the mode on (m != NULL) or off; call before the first picture */
int  ohhip_set_frames_mode(const ohhip_frames_mode *m);
/* replace avctx->execute / execute2 by versions that skip the slice data of remote pictures (call after avcodec_open2) */
struct AVCodecContext;
void ohhip_frames_install(struct AVCodecContext *avctx);
/* The decoder gave up on the picture it was decoding (avcodec_decode_video2 returned an error after hevc_frame_start): the open frame is
 * aborted and, in frames mode, published as failed so that no other process waits for it.  In openHEVC proper the call belongs on the
 * error return of hevc_decode_frame (hevc.c:4138-4144). */
int  ohdec_backend_frame_failed(void);
/* 1 if the picture in the host frame whose luma plane is data0 was reconstructed by this process (its samples are valid here) */
int  ohhip_frames_is_local(const unsigned char *data0);
#endif
