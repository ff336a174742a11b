# integration/renames.mk -- the patch points of INTEGRATION.md, applied at COMPILE time to libavcodec/hevc.c only.
#
# /root/reference is read-only, so instead of editing hevc.c each call site below is renamed by the preprocessor when that one
# translation unit is compiled; the wrappers of the same names live in integration/hip_hooks.c and call the reference's own
# function plus the libohevc_hip.so hook.  A maintainer with write access makes the same calls by hand.
#
#   call site in hevc.c                                   wrapper                      INTEGRATION.md
#   ff_hevc_dsp_init / _pred_init / ff_videodsp_init      ohhip_*_init                 section 1-2 (set_sps, hevc.c:421-423)
#   ff_hevc_set_new_ref      (hevc.c:3245)                ohhip_set_new_ref            section 3: alloc_frame + hevc_frame_start
#   ff_hevc_frame_rps        (hevc.c:3250)                ohhip_frame_rps              section 3: generate_missing_ref
#   ff_thread_report_progress(.., INT_MAX) (hevc.c:4027)  ohhip_report_progress        section 3: frame end (frame threads)
#   av_pix_fmt_desc_get      (hevc.c:4148)                ohhip_pix_fmt_desc_get       section 3: frame end before the MD5 check
#   ff_thread_await_progress (hevc.c:1951-1958)           ohhip_await_progress         section 3: no reconstructed-row waits
#   ff_hevc_cabac_init       (hevc.c:2666,2785,2873)      ohhip_cabac_init             section 2b: bind slice workers
#   ff_hevc_log2_res_scale_abs / _res_scale_sign_flag     ohhip_*                      section 2b: cross-component prediction
#   ff_hevc_hls_filters / ff_hevc_hls_filter              ohhip_hls_filter(s)          section 3: filter drivers in bulk
#   ff_hevc_deblocking_boundary_strengths (hevc.c:1578,1607,2400,2484)  ohhip_deblocking_boundary_strengths   section 3: boundary strengths on the device
#   ff_upsample_block        (hevc.c:2082,2097)           ohhip_upsample_block         section 2b: SHVC at ratio 1 (the reference copies with memcpy)
HIPRENAMES := -Dff_hevc_dsp_init=ohhip_hevc_dsp_init -Dff_hevc_pred_init=ohhip_hevc_pred_init \
              -Dff_videodsp_init=ohhip_videodsp_init -Dff_hevc_set_new_ref=ohhip_set_new_ref \
              -Dff_hevc_frame_rps=ohhip_frame_rps -Dav_pix_fmt_desc_get=ohhip_pix_fmt_desc_get \
              -Dff_thread_report_progress=ohhip_report_progress -Dff_thread_await_progress=ohhip_await_progress \
              -Dff_hevc_cabac_init=ohhip_cabac_init -Dff_hevc_log2_res_scale_abs=ohhip_log2_res_scale_abs \
              -Dff_hevc_res_scale_sign_flag=ohhip_res_scale_sign_flag \
              -Dff_hevc_hls_filters=ohhip_hls_filters -Dff_hevc_hls_filter=ohhip_hls_filter \
              -Dff_hevc_deblocking_boundary_strengths=ohhip_deblocking_boundary_strengths \
              -Dff_upsample_block=ohhip_upsample_block
