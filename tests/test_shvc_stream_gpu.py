"""SHVC end to end on the device: two-layer streams (tests/golden/shvc_streams.npz; tests/test_shvc_stream_cpu.py says how they are made
and pinned) through two instances of the reference's decoder with the gfx950 back end - the enhancement-layer decoder's back end shares the
base layer's picture store (integration/hip_backend.h: ohhip_options.base_layer), the inter-layer reference picture is resampled on the
device from the base-layer picture where it lies.  Both layers must be the untouched reference decoder's pictures."""
import numpy as np
import pytest

from oracle import pystream as ps
from shvc_cases import SHVC_CASES
from shvc_exec import check_both_layers, check_reference_md5_verdict, load_shvc, open_close_layer_pairs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(SHVC_CASES))
def test_shvc_both_layers_hip_backend(name):
    assert ps.have("hip"), "oracle/_ref/libopenhevc_hip.so missing: run __graft_entry__.build() where /root/reference exists"
    bl, el = check_both_layers("hip", name)          # pinned by the committed digests of the untouched decoder ...
    if ps.have("c"):                                 # ... and sample-exact against it when it is present
        aus, _, _ = load_shvc(name)
        ref_bl, ref_el = ps.decode_stream_shvc("c", aus)
        for got, ref in ((bl, ref_bl), (el, ref_el)):
            for fa, fb in zip(got, ref):
                for x, y in zip(fa, fb):
                    assert np.array_equal(x, y)


@pytest.mark.parametrize("name", ["x2_wpp", "snr_wpp"])
def test_shvc_slice_threads_hip_backend(name):
    check_both_layers("hip", name, threads=4, thread_type=2)


@pytest.mark.parametrize("name", ["x2_ra", "x1_5_dense", "snr"])
def test_shvc_deferred_copy_back(name, monkeypatch):
    """The pictures reach the host when the application takes them: the enhancement layer must not depend on the base-layer picture's HOST
    copy at any point (it reads the device picture)."""
    monkeypatch.setenv("OHHIP_DEFER_DOWNLOAD", "1")
    check_both_layers("hip", name)


def test_shvc_decoder_pairs_leave_nothing_behind():
    open_close_layer_pairs("hip", ps._load("hip"))


def test_shvc_reference_md5_check_on_both_layers():
    assert ps.have("gen"), "oracle/_ref/libopenhevc_gen.so missing (the stream is generated on the spot)"
    check_reference_md5_verdict("hip")
