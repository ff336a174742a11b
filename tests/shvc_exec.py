"""Shared by tests/test_shvc_stream_cpu.py (reference decoders, emulated device) and tests/test_shvc_stream_gpu.py (the device): the
committed two-layer fixtures and the checks run on them."""
import os

import numpy as np

from oracle import pystream as ps
from shvc_cases import SHVC_CASES
from test_stream_cpu import frames_md5

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shvc_streams.npz")


def load_shvc(name):
    """(access units, MD5s of the base-layer planes, MD5s of the enhancement-layer planes) as tests/golden/make_shvc_streams.py recorded
    them from the untouched reference decoder."""
    z = np.load(GOLDEN)
    data = z[name + ".data"].tobytes()
    aus, o = [], 0
    for n in z[name + ".sizes"]:
        aus.append(data[o:o + int(n)])
        o += int(n)
    return aus, [str(m) for m in z[name + ".md5_bl"]], [str(m) for m in z[name + ".md5_el"]]


def check_both_layers(kind, name, threads=1, thread_type=1):
    aus, md5_bl, md5_el = load_shvc(name)
    bl, el = ps.decode_stream_shvc(kind, aus, threads, thread_type)
    n = SHVC_CASES[name][0]["nframes"]
    assert len(bl) == n and len(el) == n
    assert frames_md5(bl) == md5_bl, f"{name}: base layer through '{kind}' differs from the reference"
    assert frames_md5(el) == md5_el, f"{name}: enhancement layer through '{kind}' differs from the reference"
    return bl, el


def open_close_layer_pairs(kind, product, rounds=8):
    """The wrapper's pair of decoders (openHevcWrapper.c:47-108) opened, used for one access unit and closed again: the enhancement layer's
    back end gives its pictures back to the store it shares, no back end stays alive."""
    aus, _, _ = load_shvc("x2_ldp")
    for _ in range(rounds):
        bl = ps.Decoder(kind)
        el = ps.Decoder(kind, decoder_id=1, base=bl)
        bl.decode(aus[0], 1)
        el.take_base(bl)
        el.decode(aus[0], 1)
        el.close()
        bl.close()
    live = product.ohhip_backend_live_count() if hasattr(product, "ohhip_backend_live_count") else 0
    assert live == 0, f"{live} back ends alive after closing every decoder"


def check_reference_md5_verdict(kind):
    """A two-layer stream with a decoded-picture-hash SEI per layer and access unit: the reference's own verification (`decode-checksum`,
    hevc.c:4146-4162) must say "Correct MD5" for every plane of both layers - with the gfx950 back end, for pictures it never computed."""
    pb = ps.StreamParams(width=128, height=96, gop="random_access", nframes=6, gop_size=4, seed=41, md5_sei=1)
    pe = ps.StreamParams(width=256, height=192, gop="random_access", nframes=6, gop_size=4, seed=41, tmvp=1)
    aus, _, _ = ps.generate_shvc(pb, pe)
    bl, el, (ok, bad) = ps.decode_stream_shvc(kind, aus, checksum=True)
    assert len(bl) == 6 and len(el) == 6
    assert (ok, bad) == (36, 0), f"'{kind}': {ok} planes verified, {bad} rejected by the reference's MD5 check (36 / 0 expected)"
