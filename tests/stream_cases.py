"""Synthetic-stream configurations shared by the CPU and GPU stream tests and tests/golden/make_streams.py."""
from oracle.pystream import DENSE_QP22

# name -> oracle.pystream.StreamParams keyword arguments.  Small pictures: the point is syntax coverage.
CASES = {
    "intra_8b": dict(gop="intra", nframes=2, seed=101),
    "intra_10b_ctb16": dict(gop="intra", nframes=2, seed=102, bit_depth=10, log2_ctb=4, log2_max_tb=4, width=136, height=72),
    "ldp_8b": dict(gop="lowdelay_p", nframes=4, seed=103),
    "ldb_8b": dict(gop="lowdelay_b", nframes=5, seed=104),
    "ldb_10b": dict(gop="lowdelay_b", nframes=4, seed=105, bit_depth=10),
    "ra_8b_ctb64": dict(gop="random_access", nframes=9, seed=106, width=416, height=240, log2_ctb=6),
    "ra_10b_odd": dict(gop="random_access", nframes=9, seed=107, bit_depth=10, width=208, height=120, log2_ctb=6),
    "weighted": dict(gop="lowdelay_b", nframes=4, seed=108, cu_qp_delta_depth=1, weighted_pred=1, weighted_bipred=1),
    "weighted_p_10b": dict(gop="lowdelay_p", nframes=4, seed=109, weighted_pred=1, bit_depth=10, cu_qp_delta_depth=2, log2_ctb=6),
    "pcm": dict(gop="lowdelay_b", nframes=4, seed=110, pcm=7),
    "pcm_10b": dict(gop="lowdelay_p", nframes=3, seed=111, pcm=9, bit_depth=10, pcm_log2_max=4),
    "cip": dict(gop="lowdelay_b", nframes=4, seed=112, constrained_intra=1),
    "wpp": dict(gop="lowdelay_b", nframes=4, seed=113, wpp=1),
    "tiles": dict(gop="lowdelay_b", nframes=4, seed=114, tiles=(2, 2)),
    "tiles_nolf": dict(gop="lowdelay_b", nframes=3, seed=115, tiles=(3, 2), loop_filter_across_tiles=0, width=256, height=128),
    "slices": dict(gop="lowdelay_b", nframes=4, seed=116, slices_per_picture=3),
    "slices_dep_wpp": dict(gop="lowdelay_b", nframes=4, seed=117, slices_per_picture=3, wpp=1, dependent_slices=1),
    "slices_nolf": dict(gop="lowdelay_b", nframes=3, seed=118, slices_per_picture=4, loop_filter_across_slices=0),
    "rext": dict(gop="lowdelay_b", nframes=4, seed=119, rext=1),
    "dense_residual": dict(gop="lowdelay_b", nframes=3, seed=120, init_qp=44,
                           probs=dict(rqt_root_cbf=0.9, cbf_luma=0.9, cbf_chroma=0.8, sig_coeff=0.7, greater1=0.6,
                                      greater2=0.6, last_x=0.8, last_y=0.8, skip=0.1)),
    "no_tools": dict(gop="lowdelay_p", nframes=3, seed=121, sao=0, deblock_control=0, tmvp=0, amp=0, sign_hiding=0,
                     transform_skip=0, strong_intra_smoothing=0),
    # RExt chroma formats (hevc.c:1291-1405: the 4:2:2 second chroma block, 4:4:4 4x4 chroma TUs and chroma smoothing)
    "fmt422_8b": dict(gop="lowdelay_b", nframes=4, seed=123, rext=1, chroma_format=2),
    "fmt422_10b_ra": dict(gop="random_access", nframes=5, seed=124, rext=1, chroma_format=2, bit_depth=10, width=208, height=120,
                          log2_ctb=6, pcm=9, weighted_bipred=1),
    "fmt444_8b": dict(gop="lowdelay_b", nframes=4, seed=125, rext=1, chroma_format=3, constrained_intra=1),
    "fmt444_10b_ctb16": dict(gop="lowdelay_p", nframes=3, seed=126, rext=1, chroma_format=3, bit_depth=10, log2_ctb=4, log2_max_tb=4,
                             width=136, height=72),
    # lossless coding units and PCM outside the loop filters (restore_tqb_pixels, hevc_filter.c:163-193)
    "tqb": dict(gop="lowdelay_b", nframes=4, seed=127, transquant_bypass=1, probs=dict(transquant_bypass=0.3)),
    "tqb_rext_422": dict(gop="lowdelay_b", nframes=3, seed=128, transquant_bypass=1, rext=1, chroma_format=2, bit_depth=10,
                         probs=dict(transquant_bypass=0.3)),
    "pcm_nolf": dict(gop="lowdelay_b", nframes=4, seed=129, pcm=7, pcm_loop_filter_disabled=1, pcm_prob=0.15),
    "pcm_nolf_ctb16": dict(gop="lowdelay_p", nframes=3, seed=130, pcm=8, pcm_loop_filter_disabled=1, pcm_prob=0.15, log2_ctb=4,
                           log2_max_tb=4, pcm_log2_max=4, transquant_bypass=1),
    # PPS range extension: cross-component prediction (hevc.c:1291-1365), larger transform-skip blocks, SAO offset scaling, 12 bit
    "cross_444_8b": dict(gop="lowdelay_b", nframes=4, seed=131, rext=1, chroma_format=3, cross_component=1,
                         probs=dict(cbf_luma=0.8, cbf_chroma=0.5, rqt_root_cbf=0.8)),
    "cross_444_10b_tqb": dict(gop="lowdelay_p", nframes=3, seed=132, rext=1, chroma_format=3, cross_component=1, bit_depth=10,
                              transquant_bypass=1, log2_max_ts=4, probs=dict(cbf_luma=0.8, transquant_bypass=0.2, transform_skip=0.4)),
    "rext_12b_sao_scale": dict(gop="lowdelay_b", nframes=3, seed=133, rext=1, bit_depth=12, sao_offset_scale=(2, 1), log2_max_ts=5,
                               probs=dict(transform_skip=0.4)),
    # SPS range extension: intra_smoothing_disabled (hevcpred_template.c:289) -- reference samples unfiltered, boundary smoothing kept
    "rext_no_intra_smoothing": dict(gop="intra", nframes=2, seed=901, rext=1, intra_smoothing_disabled=1),
    "rext_no_intra_smoothing_444_10b": dict(gop="lowdelay_b", nframes=3, seed=902, rext=1, intra_smoothing_disabled=1, bit_depth=10,
                                            chroma_format=3),
    "small_blocks": dict(gop="lowdelay_b", nframes=3, seed=122, log2_ctb=4, log2_max_tb=4,
                         probs=dict(split_cu=0.8, split_transform=0.7)),
    # BIT_DEPTH 14 (hevcdsp.c:1060-1062, 1284-1286; hevc_ps.c:1685-1689): no rounding term in the final MC shifts, DC add of 0
    "ra_14b_weighted": dict(gop="random_access", nframes=5, seed=131, bit_depth=14, rext=1, weighted_pred=1, weighted_bipred=1),
    "fmt444_14b_cip_cross": dict(gop="lowdelay_b", nframes=4, seed=132, bit_depth=14, rext=1, chroma_format=3, constrained_intra=1,
                                 cross_component=1),
    # leaf pictures of the hierarchy as sub-layer non-reference pictures (TRAIL_N): the frame-parallel decoder does not exchange them
    "ra_8b_nonref_leaves": dict(gop="random_access", nframes=9, seed=133, nonref_leaves=1, width=208, height=120),
    # ... and a leaf that stays in the reference picture set of the next picture without being used by it (a Foll entry): the frame-parallel
    # decoder must neither wait for it nor trip over it
    # BASELINE config 1's geometry and residual density: BQMall's 832x480, qp22-like statistics (~45 KB per picture at this size)
    "bqmall_geometry_dense_qp22": dict(gop="random_access", nframes=5, seed=135, width=832, height=480, log2_ctb=6, **DENSE_QP22),
    "ra_8b_foll_leaf": dict(gop="random_access", nframes=9, seed=134, nonref_leaves=1, foll_leaves=1, width=208, height=120),
}
