"""Two-layer (SHVC) stream cases: name -> (base-layer StreamParams kwargs, enhancement-layer kwargs, cross_layer_phase_alignment_flag).
Used by tests/golden/make_shvc_streams.py (fixtures), tests/test_shvc_stream_cpu.py and tests/test_shvc_stream_gpu.py.  The enhancement
layer repeats the base layer's GOP plan (oracle/pystream.py: enhancement_plan), so gop / nframes / gop_size / seed are given once."""


def _pair(common, bl, el, phase_align=0):
    b = dict(common, **bl)
    e = dict(common, **dict(dict(tmvp=0), **el))
    return b, e, phase_align


SHVC_CASES = {
    # spatial x2 (UpsamplInf.idx X2: the fixed-phase slots), low-delay P
    "x2_ldp": _pair(dict(gop="lowdelay_p", nframes=3, seed=5), dict(width=96, height=64), dict(width=192, height=128)),
    # hierarchical B pictures in both layers: the inter-layer picture sits in the middle of L0 and at the end of L1
    "x2_ra": _pair(dict(gop="random_access", nframes=9, gop_size=8, seed=7), dict(width=96, height=64), dict(width=192, height=128)),
    # x1.5 (idx X1_5)
    "x1_5_ldb": _pair(dict(gop="lowdelay_b", nframes=5, seed=8), dict(width=128, height=96), dict(width=192, height=144)),
    # sizes that are not multiples of the CTB
    "x2_odd": _pair(dict(gop="lowdelay_b", nframes=4, seed=9), dict(width=104, height=72), dict(width=208, height=144)),
    # a ratio with no fixed-pattern slot (idx DEFAULT: the general 16-phase filter)
    "ratio_5_3": _pair(dict(gop="lowdelay_p", nframes=4, seed=10), dict(width=96, height=64), dict(width=160, height=112)),
    # central-position alignment
    "x2_phase": _pair(dict(gop="lowdelay_p", nframes=4, seed=11), dict(width=96, height=64), dict(width=192, height=128), 1),
    # 16x16 CTBs below, 64x64 above
    "x2_ctb64": _pair(dict(gop="lowdelay_b", nframes=4, seed=12), dict(width=128, height=128, log2_ctb=4, log2_max_tb=4),
                      dict(width=256, height=256, log2_ctb=6)),
    # quality (SNR) scalability: ratio 1, the reference copies with memcpy
    "snr": _pair(dict(gop="lowdelay_b", nframes=4, seed=13, width=128, height=96), dict(), dict(init_qp=24)),
    # ... and with cross_layer_phase_alignment_flag set: still a copy (the reference tests the scale alone, hevc.c:486-487) - the two-layer
    # fuzzer's first real find (the back end resampled with the phase offsets)
    "snr_phase": _pair(dict(gop="lowdelay_p", nframes=3, seed=18, width=136, height=88), dict(), dict(init_qp=26), 1),
    # ratio 1 with wavefront substreams: under slice threads the copy is asked for by a row worker, not by the picture's own thread
    "snr_wpp": _pair(dict(gop="lowdelay_b", nframes=3, seed=19, width=192, height=160, wpp=1), dict(), dict(init_qp=25)),
    # wavefront substreams in both layers
    "x2_wpp": _pair(dict(gop="lowdelay_b", nframes=4, seed=14, wpp=1), dict(width=128, height=96), dict(width=256, height=192)),
    # several slices per picture (the base layer at least as many as the enhancement layer: set_refindex_data, hevc_refs.c:373-394,
    # reads the base-layer picture's reference list of the SAME slice index)
    "x2_slices": _pair(dict(gop="lowdelay_b", nframes=4, seed=15), dict(width=128, height=96, slices_per_picture=3),
                       dict(width=256, height=192, slices_per_picture=2)),
    # coding tools above the inter-layer prediction: weighted prediction, CU-level QP, PCM, lossless CUs, tiles below
    "x2_tools": _pair(dict(gop="lowdelay_p", nframes=4, seed=16), dict(width=128, height=96, tiles=(2, 2)),
                      dict(width=256, height=192, weighted_pred=1, cu_qp_delta_depth=1, pcm=7, transquant_bypass=1)),
    # low-QP statistics (dense residuals) on top of the inter-layer prediction, x1.5 with phase alignment off
    "x1_5_dense": _pair(dict(gop="random_access", nframes=5, gop_size=4, seed=17), dict(width=128, height=96),
                        dict(width=192, height=144, init_qp=22,
                             probs=dict(pred_mode=0.08, skip=0.2, rqt_root_cbf=0.92, cbf_luma=0.9, cbf_chroma=0.6, sig_coeff=0.6, greater1=0.55))),
    # temporal motion vector prediction in the enhancement layer: the collocated picture can be the inter-layer picture, whose motion
    # field is the base layer's, scaled CTB by CTB on demand (hevc_mvs.c:256, ff_upscale_mv_block hevc_filter.c:1312-1368)
    "x2_tmvp": _pair(dict(gop="random_access", nframes=6, gop_size=4, seed=21), dict(width=128, height=96), dict(width=256, height=192, tmvp=1)),
    "x1_5_tmvp": _pair(dict(gop="lowdelay_b", nframes=5, seed=23), dict(width=128, height=96), dict(width=192, height=144, tmvp=1)),
}
