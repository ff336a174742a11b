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
 * The transport is the application's: include/ohevc_frames.h has one in C (RCCL broadcast of planes and motion fields over xGMI; a sockets
 * wire for ranks that share a GPU), openhevc_amd/dist.py one on torch.distributed for the Python tests.  All processes must call the hooks with the same
 * stream; collectives are issued in decoding order on every process (publish by the owner, subscribe by the others, exactly one
 * of the two per exchanged picture).  Pictures nothing can reference (sub-layer non-reference pictures of the highest temporal
 * sub-layer) are not exchanged.  One decoding thread per process.
 */
#ifndef OHHIP_FRAMES_H
#define OHHIP_FRAMES_H
#include <stddef.h>
#include "ohevc_ctx.h"
#include "ohevc_frames.h"

/* the callback table (and a native transport that fills it: ohevc_frames_transport_*) live in the product's public header; the per-decoder
 * entry points - ohhip_backend_frames_mode (switch the mode on before the first picture), ohhip_backend_frames_install (replace
 * avctx->execute / execute2 by versions that skip the slice data of remote pictures, after avcodec_open2), ohhip_backend_frame_failed (the
 * decoder gave up on the picture it was decoding: the open frame is aborted and published as failed so that no other process waits for it;
 * in openHEVC proper the call belongs on the error return of hevc_decode_frame, hevc.c:4138-4144) and ohhip_backend_frame_is_local - in
 * hip_backend.h */
#include "hip_backend.h"
#endif
