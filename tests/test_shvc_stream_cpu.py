"""SHVC end to end (VERDICT round 3 "missing" 4), CPU tier.

Two-layer streams exist nowhere in this environment, so the synthesiser writes them (oracle/pystream.py: VPS extension, enhancement-layer
SPS / PPS / slice headers as the reference's SHVC draft parses them, hevc_ps.c:714-1095,1556-1725,2381-2385, hevc.c:728-831; slice data
by the reference's own parser with vectors into the inter-layer picture held at zero, oracle/synth_gen.c).  The reference side is driven
the way its public API does it (gpac/modules/openhevc_dec/openHevcWrapper.c:47-156): two decoders, decoder-id 0 and 1, BL_avcontext, the
base-layer picture handed over between the two avcodec_decode_video2 calls.

Pinned here: the fixtures (tests/golden/shvc_streams.npz) decode to the recorded pictures on the untouched C decoder and on the reference
as shipped on x86; the generator reproduces the fixtures byte for byte; an enhancement layer of skipped CUs IS the oracle's resampling of
the base layer (ties the bitstream level to the kernel-level oracle of the 13 upsample_* slots); and the whole hooked decoder over the
emulated device code (the -m gpu twin: tests/test_shvc_stream_gpu.py)."""
import os

import numpy as np
import pytest

from oracle import pystream as ps
from oracle import pyoracle as po
from shvc_cases import SHVC_CASES
from shvc_exec import check_both_layers, check_reference_md5_verdict, load_shvc, open_close_layer_pairs

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so")
needs_c = pytest.mark.skipif(not ps.have("c"), reason="oracle/_ref/libopenhevc_c.so not built (needs /root/reference)")
needs_gen = pytest.mark.skipif(not (ps.have("c") and ps.have("gen")), reason="oracle/_ref decoder libraries not built")


@needs_c
@pytest.mark.parametrize("name", sorted(SHVC_CASES))
def test_shvc_fixture_decodes_to_recorded_md5(name):
    check_both_layers("c", name)


@pytest.mark.skipif(not ps.have("sse"), reason="oracle/_ref/libopenhevc_sse.so not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(SHVC_CASES))
def test_shvc_fixture_on_the_reference_sse_decoder(name):
    """libavcodec/x86/hevc_il_pred_sse.c (the x2 / x1.5 slots in SSE4, x86/hevcdsp_init.c) against the C slots, through whole streams."""
    check_both_layers("sse", name)


@needs_c
def test_shvc_slice_threads_on_the_reference():
    check_both_layers("c", "x2_wpp", threads=4, thread_type=2)


@needs_gen
@pytest.mark.parametrize("name", ["x2_ldp", "x1_5_ldb", "snr", "x2_tmvp"])
def test_shvc_generator_reproduces_the_fixture(name):
    kb, ke, pa = SHVC_CASES[name]
    aus, gen_bl, gen_el = ps.generate_shvc(ps.StreamParams(**kb), ps.StreamParams(**ke), pa)
    want, _, _ = load_shvc(name)
    assert aus == want
    ref_bl, ref_el = ps.decode_stream_shvc("c", aus)
    for got, ref in ((gen_bl, ref_bl), (gen_el, ref_el)):
        assert len(got) == len(ref)
        for fa, fb in zip(got, ref):
            for x, y in zip(fa, fb):
                assert np.array_equal(x, y)


@needs_gen
@pytest.mark.parametrize("geom", [((96, 64), (192, 128), 0), ((128, 96), (192, 144), 0), ((96, 64), (160, 112), 0), ((96, 64), (192, 128), 1)],
                         ids=["x2", "x1_5", "ratio_5_3", "x2_phase_aligned"])
def test_skipped_enhancement_layer_is_the_resampled_base_layer(geom):
    """Every CU of the enhancement layer's first picture is skipped (merge, no residual), deblocking finds no edge, SAO is off: the picture
    the reference decodes is the inter-layer reference picture itself - and must equal the kernel-level oracle's resampling
    (oracle/hevc_oracle.c, pinned to the reference's slots by tests/test_oracle_vs_reference.py) of the decoded base-layer picture."""
    (bw, bh), (ew, eh), pa = geom
    pb = ps.StreamParams(width=bw, height=bh, gop="lowdelay_p", nframes=2, seed=31)
    pe = ps.StreamParams(width=ew, height=eh, gop="lowdelay_p", nframes=2, seed=31, tmvp=0, sao=0, probs=dict(skip=1.0, split_cu=0.3))
    aus, _, _ = ps.generate_shvc(pb, pe, pa)
    bl, el = ps.decode_stream_shvc("c", aus)
    up = po.shvc_params(bw, bh, ew, eh, (0, 0, 0, 0), phase_align=pa)
    want = [np.zeros((eh, ew), np.uint8), np.zeros((eh // 2, ew // 2), np.uint8), np.zeros((eh // 2, ew // 2), np.uint8)]
    po.shvc_upsample_frame(ORACLE, 8, want, ew, eh, [np.ascontiguousarray(x) for x in bl[0]], bw, bh, (0, 0, 0, 0), up, block_slots=1)
    for pl in range(3):
        assert np.array_equal(want[pl], el[0][pl]), f"plane {pl}: {int(np.count_nonzero(want[pl] != el[0][pl]))} samples differ"


@needs_gen
def test_shvc_reference_md5_check_on_both_layers():
    check_reference_md5_verdict("c")


REFLIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libhevcref.so")


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref/libhevcref.so not built (needs /root/reference)")
def test_geometries_where_the_reference_is_not_a_function_of_the_stream():
    """The reference sizes a CTB's source window a column / a row short for some ratios (hevc_filter.c:1194-1210,1262-1283) and then filters
    the base-layer frame buffer's edge or scratch rows other slot calls left behind.  The detector (three runs of the
    reference's sequence: replicated / zero border, scratch refilled with different values in front of every plane of every CTB) must flag the
    two geometries the fuzzer stumbled over, and none of the committed fixtures' - their digests are functions of the streams."""
    assert po.shvc_reference_not_a_function_of_its_inputs(REFLIB, 264, 288, 112, 144, 1, 6)[0] > 0          # luma columns from the frame edge
    bad = po.shvc_reference_not_a_function_of_its_inputs(REFLIB, 104, 280, 48, 152, 0, 5)
    assert bad[0] == 0 and bad[1] > 0 and bad[2] > 0                                                        # chroma rows from the scratch buffer
    for name, (kb, ke, pa) in SHVC_CASES.items():
        if (kb["width"], kb["height"]) == (ke["width"], ke["height"]):
            continue                                                                                        # ratio 1: a copy, no slot
        assert not any(po.shvc_reference_not_a_function_of_its_inputs(REFLIB, ke["width"], ke["height"], kb["width"], kb["height"], pa,
                                                                      ke.get("log2_ctb", 5))), name


# ---------------------------------------------------------------- the hooked decoder over the emulated device code
def _emu():
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    if not ps.have("hipemu") and ps.have("hip"):
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(here, "hipemu")], capture_output=True)
        subprocess.run(["make", "-s", "-C", os.path.join(os.path.dirname(here), "oracle"), "hipemu"], capture_output=True)
    if not ps.have("hipemu"):
        pytest.skip("oracle/_ref/libopenhevc_hipemu.so not built (needs the reference tree once)")


@pytest.mark.parametrize("name", sorted(SHVC_CASES))
def test_emu_shvc_both_layers(name):
    """Parsing, recording, the shared picture store, the device-side resampling of the inter-layer picture and every kernel, on the host."""
    _emu()
    # CPU-suite time: the emulator is ~1000x slower than the device, which runs every stream (tests/test_shvc_stream_gpu.py)
    if name in ("x2_ctb64", "x2_wpp", "snr_wpp", "x2_odd", "x1_5_dense", "x1_5_tmvp", "x2_ra", "x2_slices", "snr", "x2_phase"):
        pytest.skip("runs on the device only (CPU-suite time; x2_wpp / snr_wpp run under slice threads below)")
    check_both_layers("hipemu", name)


@pytest.mark.parametrize("name", ["x2_wpp", "snr_wpp"])
def test_emu_shvc_slice_threads(name):
    """The row workers of one picture record into one context; at ratio 1 it is a worker that asks for the device-side copy."""
    _emu()
    check_both_layers("hipemu", name, threads=4, thread_type=2)


def test_emu_shvc_decoder_pairs_leave_nothing_behind():
    _emu()
    open_close_layer_pairs("hipemu", ps._load("hipemu"), rounds=4)


def test_emu_shvc_reference_md5_check_on_both_layers():
    _emu()
    if not ps.have("gen"):
        pytest.skip("the stream is generated on the spot: needs oracle/_ref/libopenhevc_gen.so")
    check_reference_md5_verdict("hipemu")


def test_emu_shvc_two_unrelated_back_ends_fail_loudly(monkeypatch):
    """The integration mistake: the enhancement-layer decoder's back end does not name the base layer's (two picture stores).  The resampling
    slot call cannot find the base-layer picture: the enhancement-layer picture fails at its frame end - no CPU fallback, no garbage."""
    _emu()
    monkeypatch.setenv("OHDEC_SHVC_SEPARATE_STORES", "1")
    with pytest.raises(RuntimeError, match="decode error"):
        check_both_layers("hipemu", "x2_ldp")
