"""oracle/pystream.py -- TEST INFRASTRUCTURE ONLY: synthetic HEVC Annex-B streams + the reference decoder, from Python.

SURVEY.md 8f-2: no HEVC stream exists here, so streams are synthesised.  This file writes the parts that are plain bit
fields -- VPS/SPS/PPS (as parsed by the reference's hevc_ps.c:1097-1233, :1520-2150, :2151-2400) and slice segment
headers (hevc.c:520-1050, short-term RPS hevc_ps.c:72-192, pred_weight_table hevc.c:438-518) -- plans the GOP, and
assembles/escapes NAL units.  Slice DATA comes from `_ref/libopenhevc_gen.so` (oracle/synth_gen.c): the reference's
own parser driven by a seeded random bin source with an arithmetic encoder attached.

`Decoder("c")` is the untouched reference decoder (the bitstream-level oracle), `Decoder("hip")` the same front-end
with its tables filled by libohevc_hip.so (integration/hip_hooks.c), `Decoder("gen")` the generator.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {"c": "_ref/libopenhevc_c.so", "gen": "_ref/libopenhevc_gen.so", "hip": "_ref/libopenhevc_hip.so",
         "null": "_ref/libopenhevc_null.so", "sse": "_ref/libopenhevc_sse.so", "null_nobs": "_ref/libopenhevc_null_nobs.so", "hipemu": "_ref/libopenhevc_hipemu.so",
         "hipemu_asan": "_ref/libopenhevc_hipemu_asan.so", "hipemu_tsan": "_ref/libopenhevc_hipemu_tsan.so"}
_loaded = {}


def lib_path(kind: str) -> str:
    return os.path.join(_HERE, _LIBS[kind])


def have(kind: str) -> bool:
    return os.path.exists(lib_path(kind))


def _load(kind: str):
    if kind in _loaded:
        return _loaded[kind]
    if kind == "hip":
        import torch  # noqa: F401  (same reason as openhevc_amd/lib.py: torch's HIP runtime must be first)
    L = C.CDLL(lib_path(kind), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    L.ohdec_open.restype = C.c_void_p
    L.ohdec_open.argtypes = [C.c_int, C.c_int]
    L.ohdec_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int64]
    L.ohdec_flush.argtypes = [C.c_void_p]
    L.ohdec_frame_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
    L.ohdec_frame_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ohdec_close.argtypes = [C.c_void_p]
    L.ohdec_open_ex.restype = C.c_void_p
    L.ohdec_open_ex.argtypes = [C.c_int, C.c_int, C.c_int]
    L.ohdec_md5_results.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if hasattr(L, "ohdec_open_layer"):
        L.ohdec_open_layer.restype = C.c_void_p
        L.ohdec_open_layer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ohdec_take_base_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.ohdec_set_active_layer.argtypes = [C.c_void_p, C.c_int]
    if kind == "gen":
        L.ohsyn_reset.argtypes = [C.c_uint64]
        L.ohsyn_set_probs.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float]
        L.ohsyn_begin_au.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.ohsyn_slice_payload.argtypes = [C.c_int, C.POINTER(C.POINTER(C.c_uint8))]
        L.ohsyn_slice_substreams.argtypes = [C.c_int, C.POINTER(C.POINTER(C.c_uint32))]
        L.ohsyn_table_range_lps.restype = C.POINTER(C.c_uint8)
        L.ohsyn_table_trans_lps.restype = C.POINTER(C.c_uint8)
        L.ohsyn_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _loaded[kind] = L
    return L


def _product_lib():
    """libohevc_hip.so as the GPU-backed decoder sees it (same file => same loaded instance as its DT_NEEDED entry)."""
    if "product" not in _loaded:
        _loaded["product"] = C.CDLL(os.path.join(os.path.dirname(_HERE), "openhevc_amd", "libohevc_hip.so"), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    return _loaded["product"]


def _software_executor():
    """oracle/libohsw.so (sw_exec.c + the oracle): executes RECORDED jobs on the decoder's host frames.  CPU tests only."""
    if "ohsw" not in _loaded:
        L = C.CDLL(os.path.join(_HERE, "libohsw.so"), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        L.ohsw_error.restype = C.c_int
        _loaded["ohsw"] = L
    return _loaded["ohsw"]


def _configure_hip_backend():
    """OHHIP_SW_EXEC=1 (host-logic tests without a GPU): libohevc_hip.so records only and hands every frame's jobs to the software
    executor; otherwise no sink, and record-only mode only for OHHIP_RECORD_ONLY (host-side profiling).  The GPU-backed decoder
    itself (integration/hip_hooks.c) knows nothing of the oracle: this is the only place the two meet."""
    prod = _product_lib()
    prod.ohevc_debug_set_frame_sink.argtypes = [C.c_void_p, C.c_void_p]
    if os.environ.get("OHHIP_SW_EXEC"):
        sw = _software_executor()
        prod.ohevc_debug_set_record_only(1)
        prod.ohevc_debug_set_frame_sink(C.cast(sw.ohsw_sink, C.c_void_p), None)
        return sw
    prod.ohevc_debug_set_frame_sink(None, None)
    ro = os.environ.get("OHHIP_RECORD_ONLY")
    prod.ohevc_debug_set_record_only((2 if ro == "2" else 1) if ro else 0)
    return None


class Decoder:
    """One decoder instance.  decode(au) -> picture (list of 3 numpy planes) or None; flush() -> remaining pictures."""

    def __init__(self, kind: str = "c", threads: int = 1, thread_type: int = 1, checksum: bool = False, pipelined: bool = False,
                 decoder_id: int = 0, base: "Optional[Decoder]" = None, options: Optional[dict] = None):
        """decoder_id / base: SHVC, the way openHevcWrapper.c:47-108 opens its two decoders - the enhancement-layer decoder has
        decoder_id 1 and `base` = the base-layer decoder (its BL_avcontext); see take_base()."""
        self.kind = kind
        self.L = _load(kind)
        self.sw = _configure_hip_backend() if kind == "hip" else None
        if options:         # per-instance ohhip_options of this decoder (integration/hip_backend.h): level_launch, device_filters
            self.L.ohdec_set_next_options(int(options.get("level_launch", -2)), int(options.get("device_filters", -2)))
        if decoder_id or base is not None:
            self.h = self.L.ohdec_open_layer(threads, thread_type, 1 if checksum else 0, -1, decoder_id, base.h if base is not None else None)
        else:
            self.h = self.L.ohdec_open_ex(threads, thread_type, 1 if checksum else 0)
        if not self.h:
            raise RuntimeError(f"ohdec_open failed for {kind}")
        if pipelined:       # the application takes every picture one call late (decoder_harness.c): device work overlaps the next parse
            self.L.ohdec_set_pipelined.argtypes = [C.c_void_p, C.c_int]
            self.L.ohdec_set_pipelined(self.h, 1)

    def take_base(self, bl: "Decoder"):
        """SHVC: hand this (enhancement-layer) decoder the picture the base-layer decoder has just reconstructed - what
        libOpenHevcDecode does between its two avcodec_decode_video2 calls (openHevcWrapper.c:131-132)."""
        self.L.ohdec_take_base_frame(self.h, bl.h)

    def set_active_layer(self, layer: int):
        self.L.ohdec_set_active_layer(self.h, layer)

    def md5_results(self):
        """(planes whose decoded-picture-hash SEI matched, planes that did not) since the decoder was opened: the reference's own
        check, hevc.c:4146-4162, counted through its log lines."""
        ok, bad = C.c_int(), C.c_int()
        self.L.ohdec_md5_results(self.h, C.byref(ok), C.byref(bad))
        return ok.value, bad.value

    def frames_mode(self, mode):
        """Frame-parallel decoding over processes (integration/hip_frames.h): `mode` = the ohhip_frames_mode of an
        openhevc_amd.dist.FrameExchange (None switches it off).  Right after opening; one decoding thread."""
        self.L.ohdec_frames_mode.argtypes = [C.c_void_p, C.c_void_p]
        # `mode`: a ctypes structure (FrameExchange.mode) or the address of one (the native transport's, ohevc_frames_transport_mode)
        arg = None if mode is None else C.c_void_p(mode) if isinstance(mode, int) else C.byref(mode)
        if self.L.ohdec_frames_mode(self.h, arg) != 0:
            raise RuntimeError(f"decoder '{self.kind}' has no frames mode")

    def frame_is_local(self):
        """Whether the picture decode() / flush() just returned was reconstructed by this process."""
        self.L.ohdec_frame_is_local.argtypes = [C.c_void_p]
        return bool(self.L.ohdec_frame_is_local(self.h))

    def product_lib(self):
        """The HIP library this decoder is linked against (ctypes handle), for openhevc_amd.dist.FrameExchange."""
        if self.kind.startswith("hipemu"):      # hipemu, hipemu_asan, hipemu_tsan: the emulator build the decoder of that name is linked against
            return C.CDLL(os.path.join(os.path.dirname(_HERE), "tests", "hipemu", "libohevc_hip_emu" + self.kind[len("hipemu"):] + ".so"), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        return _product_lib()

    def _check_sw(self):
        if self.sw is not None and self.sw.ohsw_error():
            raise RuntimeError("software executor reported an error")

    def _fetch(self):
        w, h, bd, cw, ch = (C.c_int() for _ in range(5))
        if self.L.ohdec_frame_info(self.h, w, h, bd, cw, ch) != 0:
            raise RuntimeError("no frame")
        dt = np.uint16 if bd.value > 8 else np.uint8
        planes = []
        for c in range(3):
            pw = w.value if c == 0 else -((-w.value) >> cw.value)
            ph = h.value if c == 0 else -((-h.value) >> ch.value)
            a = np.empty((ph, pw), dtype=dt)
            self.L.ohdec_frame_copy(self.h, c, a.ctypes.data_as(C.c_void_p))
            planes.append(a)
        return planes

    def decode(self, au: bytes, pts: int = 0):
        r = self.L.ohdec_decode(self.h, au, len(au), pts)
        if r < 0:
            raise RuntimeError(f"decode error {r} ({self.kind})")
        self._check_sw()
        return self._fetch() if r else None

    def flush_one(self):
        """One more picture from the draining decoder, or None when it is empty."""
        r = self.L.ohdec_flush(self.h)
        if r < 0:
            raise RuntimeError(f"flush error {r}")
        self._check_sw()
        return self._fetch() if r else None

    def flush(self):
        out = []
        while True:
            r = self.L.ohdec_flush(self.h)
            if r < 0:
                raise RuntimeError(f"flush error {r}")
            self._check_sw()
            if not r:
                return out
            out.append(self._fetch())

    def close(self):
        if self.h:
            self.L.ohdec_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def decode_stream(kind: str, aus: Sequence[bytes], threads: int = 1, thread_type: int = 1, checksum: bool = False, pipelined: bool = False):
    """Decode a list of access units, return all output pictures in output order.  checksum=True turns the reference's
    `decode-checksum` option on (libOpenHevcSetCheckMD5, openHevcWrapper.c:429-440) and returns (pictures, (ok, bad)) with the
    number of planes whose decoded-picture-hash SEI matched / mismatched in the decoder's own check."""
    out = []
    with Decoder(kind, threads, thread_type, checksum, pipelined) as d:
        for i, au in enumerate(aus):
            f = d.decode(au, i + 1)
            if f is not None:
                out.append(f)
        out += d.flush()
        res = d.md5_results() if checksum else None
    return (out, res) if checksum else out


# ------------------------------------------------------------------------------------------------ bit writing
class Bits:
    def __init__(self):
        self.b: List[int] = []

    def u(self, n: int, v: int):
        assert 0 <= v < (1 << n) if n else v == 0, (n, v)
        for k in range(n - 1, -1, -1):
            self.b.append((v >> k) & 1)

    def ue(self, v: int):
        assert v >= 0
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v: int):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def trailing(self):
        """rbsp_trailing_bits / byte_alignment(): a one then zeros."""
        self.b.append(1)
        while len(self.b) & 7:
            self.b.append(0)

    def bytes(self) -> bytes:
        assert len(self.b) % 8 == 0
        a = np.packbits(np.array(self.b, dtype=np.uint8))
        return a.tobytes()


def escape(rbsp: bytes) -> bytes:
    """emulation prevention (7.4.2): 00 00 0x (x <= 3) -> 00 00 03 0x"""
    out = bytearray()
    z = 0
    for b in rbsp:
        if z >= 2 and b <= 3:
            out.append(3)
            z = 0
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


def nal(nal_type: int, rbsp: bytes, tid: int = 0, layer: int = 0) -> bytes:
    hdr = bytes([((nal_type << 1) & 0x7E) | (layer >> 5), ((layer & 31) << 3) | (1 + tid)])   # forbidden_zero, type(6), nuh_layer_id(6), tid_plus1(3)
    return b"\x00\x00\x00\x01" + hdr + escape(rbsp)


NAL_TRAIL_N, NAL_TRAIL_R, NAL_IDR_W_RADL, NAL_CRA = 0, 1, 19, 21
NAL_VPS, NAL_SPS, NAL_PPS = 32, 33, 34
NAL_SEI_SUFFIX = 40
SLICE_B, SLICE_P, SLICE_I = 0, 1, 2


@dataclass
class StreamParams:
    width: int = 192
    height: int = 128
    bit_depth: int = 8
    chroma_format: int = 1       # chroma_format_idc: 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 (2 and 3 need rext=1: RExt profiles)
    log2_ctb: int = 5
    log2_min_cb: int = 3
    log2_min_tb: int = 2
    log2_max_tb: int = 5
    tu_depth_inter: int = 2
    tu_depth_intra: int = 2
    amp: int = 1
    sao: int = 1
    pcm: int = 0                 # 0 off, else pcm sample bit depth
    pcm_log2_min: int = 3
    pcm_log2_max: int = 5
    pcm_loop_filter_disabled: int = 0    # with pcm: PCM CUs skip deblocking and SAO (restore_tqb_pixels, hevc_filter.c:163-193)
    transquant_bypass: int = 0           # PPS transquant_bypass_enable_flag: lossless CUs (same restore path)
    strong_intra_smoothing: int = 1
    tmvp: int = 1
    sign_hiding: int = 1
    cabac_init_present: int = 1
    init_qp: int = 30
    slice_qp_delta_range: int = 6
    constrained_intra: int = 0
    transform_skip: int = 1
    cu_qp_delta_depth: int = -1  # -1 off, else diff_cu_qp_delta_depth
    cb_qp_offset: int = 1
    cr_qp_offset: int = -1
    weighted_pred: int = 0
    weighted_bipred: int = 0
    tiles: Optional[Tuple[int, int]] = None      # (columns, rows), uniform spacing
    wpp: int = 0
    deblock_control: int = 1
    deblock_override: int = 1
    pps_beta_div2: int = 1
    pps_tc_div2: int = -1
    loop_filter_across_slices: int = 1
    loop_filter_across_tiles: int = 1
    log2_parallel_merge_level: int = 2
    max_merge_cand: int = 5
    slices_per_picture: int = 1
    dependent_slices: int = 0
    cross_component: int = 0     # PPS range extension: cross_component_prediction_enabled_flag (4:4:4 + rext only)
    log2_max_ts: int = 2         # PPS range extension: log2_max_transform_skip_block_size (rext + transform_skip)
    sao_offset_scale: Tuple[int, int] = (0, 0)   # PPS range extension: log2_sao_offset_scale_{luma,chroma} <= bit_depth - 10
    rext: int = 0                # range-extension SPS flags (implicit/explicit rdpcm, ts rotation/context, rice adaptation)
    nonref_leaves: int = 0       # random_access: pictures nothing references are coded as sub-layer non-reference pictures (TRAIL_N)
    foll_leaves: int = 0         # ... and the picture decoded right after such a leaf keeps it in its RPS, not used by the current picture
                                 # (a Foll set entry: H.265 8.3.2 only bars sub-layer non-reference pictures from the Curr sets)
    intra_smoothing_disabled: int = 0   # SPS range extension: no [1 2 1] / strong filtering of the intra reference samples (rext only)
    gop: str = "lowdelay_b"      # intra | lowdelay_p | lowdelay_b | random_access
    gop_size: int = 8
    nframes: int = 4
    seed: int = 1
    md5_sei: int = 0             # append a decoded-picture-hash SEI (MD5 of the generator's own reconstruction) to every access unit
    probs: dict = field(default_factory=dict)    # overrides of the per-syntax-element bin probabilities
    bypass_prob: float = 0.5
    pcm_prob: float = 0.05

    @property
    def ctb(self):
        return 1 << self.log2_ctb

    @property
    def ctb_w(self):
        return (self.width + self.ctb - 1) >> self.log2_ctb

    @property
    def ctb_h(self):
        return (self.height + self.ctb - 1) >> self.log2_ctb


# Syntax statistics of a low-QP (qp22-like) encode: most CUs carry a residual, dense significance maps, large coefficient levels -
# 200-250 KB per 1080p picture (the default distributions give ~35 KB, the encoder-like "natural" set ~8 KB).  BASELINE config 1 (BQMall
# 832x480 qp22) lives in this regime: the residual kernels and the coefficient upload dominate instead of the launch chain.
DENSE_QP22 = dict(init_qp=22, probs=dict(pred_mode=0.08, skip=0.2, merge_flag=0.5, split_cu=0.6, rqt_root_cbf=0.92, cbf_luma=0.9, cbf_chroma=0.6,
                                         split_transform=0.5, sig_coeff=0.6, sig_group=0.75, greater1=0.55, greater2=0.45, last_x=0.75, last_y=0.75))

# context index ranges (elem_offset[], hevc_cabac.c:98-155) and the default P(bin = 1) that shapes the synthetic syntax
CTX = {
    "sao_merge": (0, 1, 0.3), "sao_type": (1, 2, 0.7), "split_cu": (2, 5, 0.42), "transquant_bypass": (5, 6, 0.1),
    "skip": (6, 9, 0.3), "cu_qp_delta": (9, 12, 0.3), "pred_mode": (12, 13, 0.25), "part_mode": (13, 17, 0.5),
    "prev_intra_luma": (17, 18, 0.5), "intra_chroma": (18, 20, 0.5), "merge_flag": (20, 21, 0.5),
    "merge_idx": (21, 22, 0.5), "inter_pred_idc": (22, 27, 0.5), "ref_idx": (27, 31, 0.5),
    "mvd_gt0": (31, 33, 0.6), "mvd_gt1": (33, 35, 0.5), "mvp_flag": (35, 36, 0.5), "rqt_root_cbf": (36, 37, 0.6),
    "split_transform": (37, 40, 0.4), "cbf_luma": (40, 42, 0.6), "cbf_chroma": (42, 46, 0.4),
    "transform_skip": (46, 48, 0.2), "rdpcm_flag": (48, 50, 0.5), "rdpcm_dir": (50, 52, 0.5),
    "last_x": (52, 70, 0.6), "last_y": (70, 88, 0.6), "sig_group": (88, 92, 0.5), "sig_coeff": (92, 136, 0.45),
    "greater1": (136, 160, 0.35), "greater2": (160, 166, 0.35), "res_scale": (166, 176, 0.5),
    "chroma_qp_offset": (176, 178, 0.5),
}
HEVC_CONTEXTS = 199   # hevc.h (COM16_C806_EMT == 0: 183 used; the array is sized by the header)


def prob_table(overrides: dict) -> np.ndarray:
    p = np.full(256, 0.5, dtype=np.float32)
    for name, (a, b, v) in CTX.items():
        p[a:b] = overrides.get(name, v)
    return p


# ------------------------------------------------------------------------------------------------ parameter sets
def _ptl(b: Bits, p: StreamParams):
    profile = 4 if p.rext else (2 if p.bit_depth > 8 else 1)
    b.u(2, 0)
    b.u(1, 0)
    b.u(5, profile)
    for i in range(32):
        b.u(1, 1 if i == profile or (profile == 1 and i == 2) else 0)
    b.u(1, 1)   # progressive_source
    b.u(1, 0)   # interlaced_source
    b.u(1, 0)   # non_packed
    b.u(1, 1)   # frame_only
    b.u(16, 0)
    b.u(16, 0)
    b.u(12, 0)
    b.u(8, 153)  # level 5.1


def _dpb(p: StreamParams):
    if p.gop == "random_access":
        return 6, 4
    return 4, 0


def write_vps(p: StreamParams, el: Optional[StreamParams] = None, phase_align: int = 0) -> bytes:
    """`el`: a second (spatial enhancement) layer - vps_max_layers_minus1 = 1 and the vps_extension() of the SHVC draft the
    reference parses (parse_vps_extension, hevc_ps.c:714-1095, with the macro set of hevc_defs.h)."""
    b = Bits()
    b.u(4, 0)
    b.u(2, 3)
    b.u(6, 1 if el else 0)
    b.u(3, 0)
    b.u(1, 1)
    b.u(16, 0xFFFF)   # vps_extension_offset / reserved (read, not used: hevc_ps.c:1136-1138)
    _ptl(b, p)
    dpb, reorder = _dpb(p)
    b.u(1, 1)
    b.ue(dpb - 1)
    b.ue(reorder)
    b.ue(0)
    b.u(6, 1 if el else 0)    # vps_max_layer_id
    b.ue(1 if el else 0)      # vps_num_layer_sets_minus1
    if el:
        b.u(1, 1)             # layer_id_included_flag[1][0..1]
        b.u(1, 1)
    b.u(1, 0)    # timing info
    b.u(1, 1 if el else 0)    # extension
    if el:
        while len(b.b) & 7:   # vps_extension_alignment_bit_equal_to_one (align_get_bits, hevc_ps.c:1215)
            b.u(1, 1)
        _vps_extension(b, p, el, phase_align)
    b.trailing()
    return nal(NAL_VPS, b.bytes())


def _vps_extension(b: Bits, p: StreamParams, el: StreamParams, phase_align: int):
    """Two layers, layer 1 depends on layer 0 (hevc_ps.c:714-1095, in parse order)."""
    b.u(1, 0)                 # avc_base_layer_flag
    b.u(1, 0)                 # splitting_flag
    for i in range(16):       # scalability_mask: dependency (spatial / quality) scalability only
        b.u(1, 1 if i == 1 else 0)
    b.u(3, 0)                 # dimension_id_len_minus1[0]
    b.u(1, 0)                 # vps_nuh_layer_id_present_flag
    b.u(1, 1)                 # dimension_id[1][0]
    b.u(4, 0)                 # view_id_len_minus1
    b.u(1, 0)                 # view_id_val[0] (one view)
    b.u(1, 1)                 # direct_dependency_flag[1][0]
    b.u(1, 0)                 # vps_sub_layers_max_minus1_present_flag
    b.u(1, 0)                 # max_tid_ref_present_flag
    b.u(1, 1)                 # all_ref_layers_active_flag
    b.u(10, 1)                # vps_number_layer_sets_minus1 (must repeat the VPS's)
    b.u(6, 0)                 # vps_num_profile_tier_level_minus1
    b.u(1, 0)                 # more_output_layer_sets_than_default_flag
    b.u(1, 0)                 # default_one_target_output_layer_flag (two output layer sets)
    b.u(1, 0)                 # profile_level_tier_idx[1]
    b.u(1, 0)                 # alt_output_layer_flag
    b.u(1, 0)                 # rep_format_idx_present_flag: one rep_format() per layer, layer i uses format i
    for q in (p, el):         # rep_format() (parseRepFormat, hevc_ps.c:411-465)
        b.u(1, 1)             # chroma_and_bit_depth_vps_present_flag
        b.u(16, q.width)
        b.u(16, q.height)
        b.u(2, q.chroma_format)
        if q.chroma_format == 3:
            b.u(1, 0)
        b.u(4, q.bit_depth - 8)
        b.u(4, q.bit_depth - 8)
    b.u(1, 1)                 # max_one_active_ref_layer_flag
    b.u(1, phase_align)       # cross_layer_phase_alignment_flag
    b.u(1, 0)                 # sub_layer_flag_info_present_flag[1]
    dpb, reorder = _dpb(el)
    b.ue(dpb - 1)             # max_vps_dec_pic_buffering_minus1[1][0..1][0]: one value per layer of the set
    b.ue(dpb - 1)
    b.ue(reorder)             # max_vps_num_reorder_pics[1][0]
    b.ue(0)                   # max_vps_latency_increase_plus1[1][0]
    b.ue(0)                   # direct_dep_type_len_minus2
    b.u(1, 1)                 # default_direct_dependency_type_flag
    b.u(2, 2)                 # default_direct_dependency_type: sample and motion prediction
    b.u(1, 0)                 # single_layer_for_non_irap_flag
    b.u(1, 0)                 # higher_layer_irap_skip_flag
    b.u(1, 0)                 # vps_vui_present_flag


def write_sps(p: StreamParams, layer: int = 0, sps_id: int = 0) -> bytes:
    """layer > 0: the enhancement-layer form of the SHVC draft the reference parses (hevc_ps.c:1556-1662,1695-1725): no sub-layer /
    profile fields, update_rep_format_flag = 0 - size, chroma format and bit depth come from the VPS's rep_format() of the layer."""
    assert p.width % (1 << p.log2_min_cb) == 0 and p.height % (1 << p.log2_min_cb) == 0
    b = Bits()
    b.u(4, 0)
    if layer == 0:
        b.u(3, 0)
        b.u(1, 1)
        _ptl(b, p)
    b.ue(sps_id)                  # sps_id
    if layer == 0:
        b.ue(p.chroma_format)         # chroma_format_idc
        if p.chroma_format == 3:
            b.u(1, 0)                 # separate_colour_plane_flag
        b.ue(p.width)
        b.ue(p.height)
    else:
        assert p.chroma_format == 1 and not p.rext
        b.u(1, 0)                 # update_rep_format_flag
    b.u(1, 0)                     # conformance window
    if layer == 0:
        b.ue(p.bit_depth - 8)
        b.ue(p.bit_depth - 8)
    b.ue(4)                       # log2_max_poc_lsb = 8
    dpb, reorder = _dpb(p)
    b.u(1, 1)
    b.ue(dpb - 1)
    b.ue(reorder)
    b.ue(0)
    b.ue(p.log2_min_cb - 3)
    b.ue(p.log2_ctb - p.log2_min_cb)
    b.ue(p.log2_min_tb - 2)
    b.ue(p.log2_max_tb - p.log2_min_tb)
    b.ue(p.tu_depth_inter)
    b.ue(p.tu_depth_intra)
    b.u(1, 0)                     # scaling lists
    b.u(1, p.amp)
    b.u(1, p.sao)
    b.u(1, 1 if p.pcm else 0)
    if p.pcm:
        b.u(4, p.pcm - 1)
        b.u(4, p.pcm - 1)
        b.ue(p.pcm_log2_min - 3)
        b.ue(p.pcm_log2_max - p.pcm_log2_min)
        b.u(1, p.pcm_loop_filter_disabled)
    b.ue(0)                       # num_short_term_ref_pic_sets: every slice header carries its own
    b.u(1, 0)                     # long-term refs
    b.u(1, p.tmvp)
    b.u(1, p.strong_intra_smoothing)
    b.u(1, 0)                     # vui
    if p.rext:
        b.u(1, 1)                 # sps_extension_flag
        b.u(1, 1)                 # range extension (hevc_ps.c: one flag + 7 bits)
        b.u(7, 0)
        b.u(1, 1)                 # transform_skip_rotation_enabled
        b.u(1, 1)                 # transform_skip_context_enabled
        b.u(1, 1)                 # implicit_rdpcm_enabled
        b.u(1, 1)                 # explicit_rdpcm_enabled
        b.u(1, 0)                 # extended_precision_processing
        b.u(1, p.intra_smoothing_disabled)
        b.u(1, 0)                 # high_precision_offsets
        b.u(1, 1)                 # persistent_rice_adaptation
        b.u(1, 0)                 # cabac_bypass_alignment
    else:
        b.u(1, 0)
    b.trailing()
    return nal(NAL_SPS, b.bytes(), layer=layer)


def write_pps(p: StreamParams, layer: int = 0, pps_id: int = 0, sps_id: int = 0) -> bytes:
    b = Bits()
    b.ue(pps_id)
    b.ue(sps_id)
    b.u(1, p.dependent_slices)
    b.u(1, 0)                     # output_flag_present
    b.u(3, 0)                     # extra slice header bits
    b.u(1, p.sign_hiding)
    b.u(1, p.cabac_init_present)
    b.ue(0)
    b.ue(0)
    b.se(p.init_qp - 26)
    b.u(1, p.constrained_intra)
    b.u(1, p.transform_skip)
    b.u(1, 1 if p.cu_qp_delta_depth >= 0 else 0)
    if p.cu_qp_delta_depth >= 0:
        b.ue(p.cu_qp_delta_depth)
    b.se(p.cb_qp_offset)
    b.se(p.cr_qp_offset)
    b.u(1, 1)                     # slice-level chroma qp offsets present
    b.u(1, p.weighted_pred)
    b.u(1, p.weighted_bipred)
    b.u(1, p.transquant_bypass)
    b.u(1, 1 if p.tiles else 0)
    b.u(1, p.wpp)
    if p.tiles:
        b.ue(p.tiles[0] - 1)
        b.ue(p.tiles[1] - 1)
        b.u(1, 1)
        b.u(1, p.loop_filter_across_tiles)
    b.u(1, p.loop_filter_across_slices)
    b.u(1, p.deblock_control)
    if p.deblock_control:
        b.u(1, p.deblock_override)
        b.u(1, 0)
        b.se(p.pps_beta_div2)
        b.se(p.pps_tc_div2)
    if layer:
        b.u(1, 0)                 # pps_infer_scaling_list_flag (hevc_ps.c:2381-2385)
    b.u(1, 0)                     # scaling list data
    b.u(1, 0)                     # lists_modification_present
    b.ue(p.log2_parallel_merge_level - 2)
    b.u(1, 0)                     # slice header extension
    if p.rext and (p.cross_component or p.log2_max_ts > 2 or any(p.sao_offset_scale)):
        b.u(1, 1)                 # pps_extension_present_flag
        b.u(1, 1)                 # pps_range_extensions_flag (hevc_ps.c:2421-2428, :2086-2150)
        b.u(7, 0)
        if p.transform_skip:
            b.ue(p.log2_max_ts - 2)
        b.u(1, p.cross_component)
        b.u(1, 0)                 # chroma_qp_offset_list_enabled_flag ("not yet implemented" in the reference)
        b.ue(p.sao_offset_scale[0])
        b.ue(p.sao_offset_scale[1])
    else:
        b.u(1, 0)                 # pps extension
    b.trailing()
    return nal(NAL_PPS, b.bytes(), layer=layer)


# ------------------------------------------------------------------------------------------------ GOP plan
@dataclass
class Pic:
    poc: int
    nal_type: int
    slice_type: int
    rps_neg: List[Tuple[int, int]]    # (poc, used_by_curr) sorted by decreasing poc
    rps_pos: List[Tuple[int, int]]    # sorted by increasing poc
    nref: Tuple[int, int] = (0, 0)


def plan_gop(p: StreamParams) -> List[Pic]:
    n = p.nframes
    if p.gop == "intra":
        return [Pic(i, NAL_IDR_W_RADL if i == 0 else NAL_TRAIL_R, SLICE_I, [], []) for i in range(n)]
    if p.gop in ("lowdelay_p", "lowdelay_b"):
        st = SLICE_P if p.gop == "lowdelay_p" else SLICE_B
        pics = [Pic(0, NAL_IDR_W_RADL, SLICE_I, [], [])]
        for i in range(1, n):
            neg = [(i - d, 1) for d in (1, 2, 3) if i - d >= 0]
            k = min(len(neg), 2)
            pics.append(Pic(i, NAL_TRAIL_R, st, neg, [], (k, k if st == SLICE_B else 0)))
        return pics
    assert p.gop == "random_access"
    # hierarchical-B: decode order inside each GOP = anchor first, then recursive midpoints
    order, refs = [0], {0: []}
    g = p.gop_size
    base = 0
    while base + 1 < n:
        top = min(base + g, n - 1)
        order.append(top)
        refs[top] = [base]

        def mid(lo, hi):
            if hi - lo < 2:
                return
            m = (lo + hi) // 2
            order.append(m)
            refs[m] = [lo, hi]
            mid(lo, m)
            mid(m, hi)
        mid(base, top)
        base = top
    pics = []
    for i, poc in enumerate(order):
        if i == 0:
            pics.append(Pic(0, NAL_IDR_W_RADL, SLICE_I, [], []))
            continue
        decoded = set(order[:i])
        later = set()
        for q in order[i:]:
            later.update(refs[q])
        keep = sorted(decoded & later)
        cur = set(refs[poc])
        neg = [(q, int(q in cur)) for q in sorted((q for q in keep if q < poc), reverse=True)]
        pos = [(q, int(q in cur)) for q in sorted(q for q in keep if q > poc)]
        two_sided = any(q > poc for q in cur)
        st = SLICE_B if (two_sided or poc % 2 == 0) else SLICE_P
        k = max(1, min(2, len(cur)))
        later_refs = set()
        for q in order[i + 1:]:
            later_refs.update(refs[q])
        nt = NAL_TRAIL_N if p.nonref_leaves and poc not in later_refs else NAL_TRAIL_R
        if p.foll_leaves and p.nonref_leaves and i >= 2 and pics[-1].nal_type == NAL_TRAIL_N:
            q = pics[-1].poc                      # the leaf decoded just before: kept, unused (it lands in ST_FOLL)
            if q < poc:
                neg = sorted(neg + [(q, 0)], reverse=True)
            else:
                pos = sorted(pos + [(q, 0)])
        pics.append(Pic(poc, nt, st, neg, pos, (k, k if st == SLICE_B else 0)))
    return pics


# ------------------------------------------------------------------------------------------------ slice header
def write_slice_header(p: StreamParams, pic: Pic, rng: np.random.Generator, seg_addr: int, dependent: int,
                       entry_points: Optional[Sequence[int]], fixed: dict, layer: int = 0, pps_id: int = 0) -> Bits:
    """`fixed` carries the per-picture choices (qp delta, flags) so that all segments of a picture agree.  layer > 0: an SHVC
    enhancement-layer slice - slice_pic_order_cnt_lsb in IDR pictures too (hevc.c:728-729) and inter_layer_pred_enabled_flag = 1
    behind the reference picture sets (hevc.c:804-831: one direct reference layer, nothing else to send)."""
    b = Bits()
    first = seg_addr == 0
    b.u(1, 1 if first else 0)
    if 16 <= pic.nal_type <= 23:
        b.u(1, 0)                                 # no_output_of_prior_pics
    b.ue(pps_id)                                  # pps id
    if not first:
        if p.dependent_slices:
            b.u(1, dependent)
        b.u((p.ctb_w * p.ctb_h - 1).bit_length(), seg_addr)       # av_ceil_log2(ctb count), hevc.c:613
    if not dependent:
        b.ue(pic.slice_type)
        idr = pic.nal_type in (19, 20)
        if idr and layer:
            b.u(8, pic.poc & 0xFF)
        if not idr:
            b.u(8, pic.poc & 0xFF)
            b.u(1, 0)                             # short_term_ref_pic_set_sps_flag
            b.ue(len(pic.rps_neg))
            b.ue(len(pic.rps_pos))
            prev = pic.poc
            for q, used in pic.rps_neg:
                b.ue(prev - q - 1)
                b.u(1, used)
                prev = q
            prev = pic.poc
            for q, used in pic.rps_pos:
                b.ue(q - prev - 1)
                b.u(1, used)
                prev = q
            if p.tmvp:
                b.u(1, fixed["tmvp"])
        if layer:
            b.u(1, 1)                             # inter_layer_pred_enabled_flag
        if p.sao:
            b.u(1, fixed["sao_luma"])
            b.u(1, fixed["sao_chroma"])
        if pic.slice_type != SLICE_I:
            b.u(1, 1)                             # num_ref_idx_active_override_flag
            b.ue(pic.nref[0] - 1)
            if pic.slice_type == SLICE_B:
                b.ue(pic.nref[1] - 1)
                b.u(1, fixed["mvd_l1_zero"])
            if p.cabac_init_present:
                b.u(1, fixed["cabac_init"])
            if p.tmvp and not idr and fixed["tmvp"]:
                col_l0 = 1
                if pic.slice_type == SLICE_B:
                    col_l0 = fixed["col_l0"]
                    b.u(1, col_l0)
                if pic.nref[0 if col_l0 else 1] > 1:
                    b.ue(fixed["col_ref"] % pic.nref[0 if col_l0 else 1])
            if (p.weighted_pred and pic.slice_type == SLICE_P) or (p.weighted_bipred and pic.slice_type == SLICE_B):
                for v, kind in fixed["wp"]:
                    getattr(b, kind)(*v) if kind == "u" else getattr(b, kind)(v)
            b.ue(5 - p.max_merge_cand)
        b.se(fixed["qp_delta"])
        b.se(fixed["cb_off"])
        b.se(fixed["cr_off"])
        disable_dbf = 0
        if p.deblock_control:
            if p.deblock_override:
                b.u(1, fixed["dbk_override"])
                if fixed["dbk_override"]:
                    disable_dbf = fixed["dbk_disable"]
                    b.u(1, disable_dbf)
                    if not disable_dbf:
                        b.se(fixed["beta_div2"])
                        b.se(fixed["tc_div2"])
        if p.loop_filter_across_slices and (
                (p.sao and (fixed["sao_luma"] or fixed["sao_chroma"])) or not disable_dbf):
            b.u(1, fixed["lf_across"])
    if p.tiles or p.wpp:
        n = len(entry_points) if entry_points else 0
        b.ue(n)
        if n:
            b.ue(31)                              # offset_len_minus1: fixed 32-bit fields, patched after generation
            for v in entry_points:
                b.u(32, v - 1)
    b.trailing()                                  # byte_alignment()
    return b


def _wp_table(p: StreamParams, pic: Pic, rng) -> list:
    """pred_weight_table() as (value, writer) pairs, hevc.c:438-518"""
    out = []
    denom = int(rng.integers(0, 7))
    out.append((denom, "ue"))
    out.append((int(rng.integers(-1, 2)) if 0 < denom < 6 else 0, "se"))
    for lst in range(2 if pic.slice_type == SLICE_B else 1):
        n = pic.nref[lst]
        lf = [int(rng.integers(0, 2)) for _ in range(n)]
        cf = [int(rng.integers(0, 2)) for _ in range(n)]
        out += [((1, f), "u") for f in lf]
        out += [((1, f), "u") for f in cf]
        for i in range(n):
            if lf[i]:
                out.append((int(rng.integers(-8, 9)), "se"))
                out.append((int(rng.integers(-20, 21)), "se"))
            if cf[i]:
                for _ in range(2):
                    out.append((int(rng.integers(-8, 9)), "se"))
                    out.append((int(rng.integers(-40, 41)), "se"))
    return out


def _fixed_choices(p: StreamParams, pic: Pic, rng) -> dict:
    f = dict(
        tmvp=int(rng.integers(0, 4) != 0), sao_luma=int(rng.integers(0, 5) != 0), sao_chroma=int(rng.integers(0, 5) != 0),
        mvd_l1_zero=int(rng.integers(0, 4) == 0), cabac_init=int(rng.integers(0, 2)), col_l0=int(rng.integers(0, 2)),
        col_ref=int(rng.integers(0, 4)), qp_delta=int(rng.integers(-p.slice_qp_delta_range, p.slice_qp_delta_range + 1)),
        cb_off=int(rng.integers(-2, 3)), cr_off=int(rng.integers(-2, 3)), dbk_override=int(rng.integers(0, 2)),
        dbk_disable=int(rng.integers(0, 8) == 0), beta_div2=int(rng.integers(-3, 4)), tc_div2=int(rng.integers(-3, 4)),
        lf_across=int(rng.integers(0, 4) != 0),
    )
    qp = p.init_qp + f["qp_delta"]
    if not (-6 * (p.bit_depth - 8) <= qp <= 51):
        f["qp_delta"] = 0
    if pic.slice_type != SLICE_I:
        f["wp"] = _wp_table(p, pic, rng)
    return f


# ------------------------------------------------------------------------------------------------ the generator
def _slice_layout(p: StreamParams, rng) -> List[Tuple[int, int, int]]:
    """(first CTB in tile-scan order, CTB count, dependent) per slice segment.  With tiles or WPP the segments are
    whole tiles / whole CTB rows so that every legal combination rule of 6.3.1 holds trivially."""
    total = p.ctb_w * p.ctb_h
    n = max(1, min(p.slices_per_picture, total))
    if n == 1:
        return [(0, total, 0)]
    if p.tiles:
        raise NotImplementedError("multiple slices with tiles")
    if p.wpp:
        rows = sorted(set([0] + [int(r) for r in rng.choice(np.arange(1, p.ctb_h), size=min(n - 1, p.ctb_h - 1),
                                                              replace=False)]))
        cuts = [r * p.ctb_w for r in rows]
    else:
        cuts = sorted(set([0] + [int(c) for c in rng.choice(np.arange(1, total), size=n - 1, replace=False)]))
    out = []
    for i, c in enumerate(cuts):
        end = cuts[i + 1] if i + 1 < len(cuts) else total
        dep = int(p.dependent_slices and i > 0 and rng.integers(0, 2) == 1)
        out.append((c, end - c, dep))
    return out


def _entry_point_count(p: StreamParams, first_ctb: int, count: int) -> int:
    if p.tiles:
        return p.tiles[0] * p.tiles[1] - 1          # one slice per picture in tile mode
    if p.wpp:
        r0 = first_ctb // p.ctb_w
        r1 = (first_ctb + count - 1) // p.ctb_w
        return r1 - r0
    return 0


def _generate_picture(L, gen: "Decoder", p: StreamParams, pic: Pic, rng, pts: int, prefix: bytes, layer: int = 0, pps_id: int = 0):
    """One picture of one layer through the generator: returns (its slice NAL units, the picture the generator put out or None).
    `prefix`: parameter sets the generator's decoder has to see in front of the slices (not part of the returned bytes)."""
    fixed = _fixed_choices(p, pic, rng)
    layout = _slice_layout(p, rng)
    # pass 1: headers with placeholder entry points + a dummy payload, through the generator
    hdr_bits, au = [], prefix
    for first_ctb, count, dep in layout:
        nep = _entry_point_count(p, first_ctb, count)
        hb = write_slice_header(p, pic, rng, first_ctb, dep, [1] * nep if nep else None, fixed, layer, pps_id)
        hdr_bits.append((hb, nep))
        au += nal(pic.nal_type, hb.bytes() + b"\xff" * (8 + 4 * nep), layer=layer)
    counts = (C.c_int * len(layout))(*[c for _, c, _ in layout])
    L.ohsyn_begin_au(counts, len(layout))
    f = gen.decode(au, pts)
    if L.ohsyn_num_slices() != len(layout):
        raise RuntimeError(f"generator produced {L.ohsyn_num_slices()} slice payloads, planned {len(layout)}")
    # pass 2: real NAL units
    out = b""
    for k, ((first_ctb, count, dep), (hb, nep)) in enumerate(zip(layout, hdr_bits)):
        ptr = C.POINTER(C.c_uint8)()
        n = L.ohsyn_slice_payload(k, C.byref(ptr))
        if n < 0:
            raise RuntimeError("slice segment incomplete: the reference parser rejected the random syntax")
        payload = bytes(np.ctypeslib.as_array(ptr, shape=(n,))) if n else b""
        if nep:
            sp = C.POINTER(C.c_uint32)()
            ns = L.ohsyn_slice_substreams(k, C.byref(sp))
            starts = [int(sp[j]) for j in range(ns)] + [n]
            if ns != nep + 1:
                raise RuntimeError(f"{ns} substreams for {nep} entry points")
            # entry points count the bytes of the ESCAPED substreams (7.4.7.1)
            sizes = _escaped_sizes(hb.bytes(), payload, starts)
            hb = write_slice_header(p, pic, rng, first_ctb, dep, sizes[:-1], fixed, layer, pps_id)
            # rewriting the sizes can move emulation-prevention bytes of the header only, not of the data
        out += nal(pic.nal_type, hb.bytes() + payload, layer=layer)
    return out, f


def _set_probs(L, p: StreamParams):
    probs = prob_table(p.probs)
    L.ohsyn_set_probs(probs.ctypes.data_as(C.POINTER(C.c_float)), len(probs), p.bypass_prob, p.pcm_prob if p.pcm else 0.0)


def generate(p: StreamParams, check: bool = True):
    """Returns (list of access units as bytes, list of pictures the generator reconstructed in output order)."""
    L = _load("gen")
    rng = np.random.default_rng(p.seed)
    L.ohsyn_reset(p.seed)
    _set_probs(L, p)
    headers = write_vps(p) + write_sps(p) + write_pps(p)
    aus, frames = [], []
    gen = Decoder("gen")
    try:
        for i, pic in enumerate(plan_gop(p)):
            out, f = _generate_picture(L, gen, p, pic, rng, i + 1, headers if i == 0 else b"")
            if f is not None:
                frames.append(f)
            aus.append((headers if i == 0 else b"") + out)
        frames += gen.flush()
    finally:
        gen.close()
    if p.md5_sei:
        # pictures come out in output order = increasing POC (one IDR per stream): access unit i carries picture pocs[i]
        pocs = [pic.poc for pic in plan_gop(p)]
        rank = {poc: k for k, poc in enumerate(sorted(pocs))}
        assert len(frames) == len(pocs)
        aus = [au + md5_sei_nal(frames[rank[poc]]) for au, poc in zip(aus, pocs)]
    return aus, frames


# ------------------------------------------------------------------------------------------------ SHVC: two spatial layers
def enhancement_plan(pics: List[Pic]) -> List[Pic]:
    """The enhancement layer's pictures: the base layer's plan (same POCs, same decoding order, same short-term reference picture
    sets) with one more active reference - the inter-layer reference picture, the resampled base-layer picture of the same access unit
    (list order: ST_CURR_BEF, inter-layer, ST_CURR_AFT for L0; ST_CURR_AFT, ST_CURR_BEF, inter-layer for L1 - ff_hevc_slice_rpl,
    hevc_refs.c:443-449).  Intra base-layer pictures become P pictures predicted from the inter-layer picture alone (an IRAP with P
    slices is what decoder_id > 0 admits, hevc.c:712)."""
    out = []
    for pic in pics:
        used = sum(u for _, u in pic.rps_neg) + sum(u for _, u in pic.rps_pos)
        if pic.slice_type == SLICE_I:
            out.append(Pic(pic.poc, pic.nal_type, SLICE_P, pic.rps_neg, pic.rps_pos, (1 if pic.nal_type in (19, 20, 21) else used + 1, 0)))
        else:
            n = used + 1
            out.append(Pic(pic.poc, pic.nal_type, pic.slice_type, pic.rps_neg, pic.rps_pos, (n, n if pic.slice_type == SLICE_B else 0)))
    return out


def shvc_headers(pb: StreamParams, pe: StreamParams, phase_align: int = 0) -> bytes:
    """VPS with its extension, base-layer SPS 0 / PPS 0, enhancement-layer SPS 1 / PPS 1 (nuh_layer_id 1).  The enhancement-layer decoder
    looks the base layer's SPS up as sps_list[decoder_id - 1] (hevc.c:453): SPS 0 must be the base layer's."""
    return (write_vps(pb, pe, phase_align) + write_sps(pb) + write_pps(pb) + write_sps(pe, layer=1, sps_id=1)
            + write_pps(pe, layer=1, pps_id=1, sps_id=1))


def generate_shvc(pb: StreamParams, pe: StreamParams, phase_align: int = 0):
    """A two-layer (spatially scalable) stream: returns (access units, base-layer pictures, enhancement-layer pictures) - the pictures as
    the generator's own two decoders reconstructed them, in output order.  Every access unit holds the base-layer slices (nuh_layer_id 0)
    and the enhancement-layer slices (nuh_layer_id 1) of one picture; two decoders opened the way openHevcWrapper.c does take them
    (decode_stream_shvc).  Vectors into the inter-layer reference picture are zero (oracle/synth_gen.c: ohsyn_mvd_coding)."""
    assert pe.bit_depth == pb.bit_depth == 8 and pe.chroma_format == pb.chroma_format == 1, "the reference's rep_format path: 8-bit 4:2:0"
    L = _load("gen")
    rng = np.random.default_rng(pb.seed)
    L.ohsyn_reset(pb.seed)
    headers = shvc_headers(pb, pe, phase_align)
    aus, frames_bl, frames_el = [], [], []
    gen_bl = Decoder("gen")
    gen_el = Decoder("gen", decoder_id=1, base=gen_bl)
    gen_bl.set_active_layer(1)
    try:
        plan_bl = plan_gop(pb)
        plan_el = enhancement_plan(plan_bl)
        for i, (pic_b, pic_e) in enumerate(zip(plan_bl, plan_el)):
            pre = headers if i == 0 else b""
            _set_probs(L, pb)
            out_b, f = _generate_picture(L, gen_bl, pb, pic_b, rng, i + 1, pre)
            if f is not None:
                frames_bl.append(f)
            _set_probs(L, pe)
            gen_el.take_base(gen_bl)
            out_e, f = _generate_picture(L, gen_el, pe, pic_e, rng, i + 1, pre, layer=1, pps_id=1)
            if f is not None:
                frames_el.append(f)
            aus.append(pre + out_b + out_e)
        frames_bl += gen_bl.flush()
        frames_el += gen_el.flush()
    finally:
        gen_el.close()
        gen_bl.close()
    if pb.md5_sei:
        # one decoded-picture-hash SEI per layer and access unit (nuh_layer_id 0 / 1: each decoder takes its own, hevc.c:3303); pictures come
        # out in increasing POC order, access unit i carries the pictures of pocs[i]
        pocs = [pic.poc for pic in plan_bl]
        rank = {poc: k for k, poc in enumerate(sorted(pocs))}
        assert len(frames_bl) == len(frames_el) == len(pocs)
        aus = [au + md5_sei_nal(frames_bl[rank[poc]]) + md5_sei_nal(frames_el[rank[poc]], layer=1) for au, poc in zip(aus, pocs)]
    return aus, frames_bl, frames_el


def decode_stream_shvc(kind: str, aus: Sequence[bytes], threads: int = 1, thread_type: int = 1, checksum: bool = False):
    """Both layers of a two-layer stream, the way libOpenHevcDecode drives its two decoders (openHevcWrapper.c:110-156): every access unit
    goes to the base-layer decoder, then - with the base-layer picture handed over - to the enhancement-layer decoder.  Returns
    (base-layer pictures, enhancement-layer pictures) in output order."""
    out_b, out_e = [], []
    bl = Decoder(kind, threads, thread_type, checksum)
    el = Decoder(kind, threads, thread_type, checksum, decoder_id=1, base=bl)
    bl.set_active_layer(1)
    res = None
    try:
        for i, au in enumerate(aus):
            f = bl.decode(au, i + 1)
            if f is not None:
                out_b.append(f)
            el.take_base(bl)
            f = el.decode(au, i + 1)
            if f is not None:
                out_e.append(f)
        out_b += bl.flush()
        out_e += el.flush()
        res = el.md5_results() if checksum else None      # (the counters are the library's: both decoders' checks)
    finally:
        el.close()
        bl.close()
    return (out_b, out_e, res) if checksum else (out_b, out_e)


def md5_sei_nal(planes, layer: int = 0) -> bytes:
    """Suffix SEI, payload type 132 (decoded picture hash, D.2.19), hash_type 0: one MD5 per colour plane over the samples in raster
    order, 16-bit samples little-endian -- parsed by decode_nal_sei_decoded_picture_hash (hevc_sei.c:28-45), checked against
    calc_md5 of the decoded planes in hevc_decode_frame (hevc.c:4146-4162, :4623-4637)."""
    import hashlib
    body = bytes([0])
    for pl in planes:
        body += hashlib.md5(np.ascontiguousarray(pl).astype("<u2" if pl.dtype.itemsize == 2 else np.uint8).tobytes()).digest()
    rbsp = bytes([132, len(body)]) + body + b"\x80"
    return nal(NAL_SEI_SUFFIX, rbsp, layer=layer)


def _escaped_sizes(header: bytes, payload: bytes, starts: List[int]) -> List[int]:
    """size of every substream after emulation prevention, given the RBSP byte offsets where each starts"""
    rbsp = header + payload
    # positions (in RBSP coordinates) before which an emulation prevention byte is inserted
    ins = []
    z = 0
    for i, b in enumerate(rbsp):
        if z >= 2 and b <= 3:
            ins.append(i)
            z = 0
        z = z + 1 if b == 0 else 0
    ins = np.array(ins, dtype=np.int64)
    h = len(header)
    sizes = []
    for a, e in zip(starts[:-1], starts[1:]):
        extra = int(np.count_nonzero((ins >= h + a) & (ins < h + e)))
        sizes.append(e - a + extra)
    return sizes
