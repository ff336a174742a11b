"""ctypes access to the CPU oracles -- TEST INFRASTRUCTURE ONLY.

Two libraries expose the same C API (oracle/oracle_api.h):

* ``ref``    -> oracle/_ref/libhevcref.so : the reference's own hevcdsp.c/hevcpred.c compiled unmodified
* ``oracle`` -> oracle/liboracle.so       : our plain-C restatement (oracle/hevc_oracle.c)

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

TU_IDCT, TU_DC, TU_DST4, TU_SKIP, TU_SKIP_RDPCM_H, TU_SKIP_RDPCM_V, TU_BYPASS, TU_BYPASS_RDPCM_H, TU_BYPASS_RDPCM_V = range(9)
MC_PUT, MC_UNI, MC_UNI_W, MC_BI, MC_BI_W = range(5)


class IntraPic(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("linesize", C.c_int32 * 3), ("width", C.c_int32), ("height", C.c_int32),
                ("chroma_format_idc", C.c_int32), ("log2_ctb_size", C.c_int32), ("log2_min_tb_size", C.c_int32),
                ("log2_min_pu_size", C.c_int32), ("strong_intra_smoothing", C.c_int32),
                ("intra_smoothing_disabled", C.c_int32), ("constrained_intra_pred", C.c_int32),
                ("is_intra", C.c_void_p)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build(target="all"):
    """(Re)build the oracle libraries; the reference part is skipped when /root/reference is absent."""
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


class OracleLib:
    """One implementation of oracle_api.h (prefix ``ohref_`` or ``ohor_``)."""

    def __init__(self, path, prefix):
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path)

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # ---- transforms
    def tu_residual(self, bd, kind, log2, coeffs, col_limit=None):
        """Returns the residual (int16, N x N) of a dense coefficient block."""
        n = 1 << log2
        c = np.ascontiguousarray(coeffs, dtype=np.int16).reshape(n * n).copy()
        buf = _aligned_copy(c)
        self._f("tu_residual")(C.c_int(bd), C.c_int(kind), C.c_int(log2), _p(buf), C.c_int(n if col_limit is None else col_limit))
        return buf.reshape(n, n).copy()

    def tu_cross(self, bd, log2, kind_c, coeffs_c, kind_y, coeffs_y, res_scale_val, plane, x, y):
        """Cross-component prediction of one chroma block (RExt 4:4:4), in place on `plane`: the tail of
        ff_hevc_hls_residual_coding (hevc_cabac.c:1942-1949: coeffs[i] += (res_scale_val * coeffs_y[i]) >> 3 on the int16
        residuals, then transform_add) and, for a block without coded coefficients (kind_c None), hls_transform_unit
        (hevc.c:1315-1330: coeffs[i] = (res_scale_val * coeffs_y[i]) >> 3).  coeffs_y are the luma block's coefficients,
        whose residual the reference finds in lc->tu.coeffs[0] after the luma block's in-place inverse transform."""
        n = 1 << log2
        ry = self.tu_residual(bd, kind_y, log2, coeffs_y).astype(np.int32)
        rc = self.tu_residual(bd, kind_c, log2, coeffs_c).astype(np.int32) if kind_c is not None else np.zeros((n, n), np.int32)
        res = (rc + ((res_scale_val * ry) >> 3)).astype(np.int16)         # the reference stores into int16_t coeffs[]
        buf = _aligned_copy(res.reshape(-1))
        addr = plane.ctypes.data + y * plane.strides[0] + x * plane.itemsize
        self._f("transform_add")(C.c_int(bd), C.c_int(log2), C.c_void_p(addr), C.c_ssize_t(plane.strides[0]), _p(buf))
        return plane

    def tu_batch(self, bd, kind, log2, coeffs, plane, xy, col_limit=None, threads=0):
        """In-place on `plane` (2-D uint8/uint16 array): adds the residual of every block at its (x,y)."""
        n = 1 << log2
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        nblk = xy.shape[0]
        assert coeffs.size == nblk * n * n and plane.flags.c_contiguous
        args = [C.c_int(bd), C.c_int(kind), C.c_int(log2), C.c_int(nblk), _p(coeffs), _p(plane),
                C.c_ssize_t(plane.strides[0]), _p(xy), C.c_int(n if col_limit is None else col_limit)]
        if threads:
            self._f("tu_batch_mt")(*args, C.c_int(threads))
        else:
            self._f("tu_batch")(*args)
        return plane

    # ---- motion compensation.  `ref` is a 2-D pixel array, (sx, sy) the integer sample the block starts at.
    def mc(self, bd, luma, variant, ref, sx, sy, w, h, mx, my, src2=None, denom=0, wx0=0, wx1=0, ox0=0, ox1=0):
        ps = ref.itemsize
        src_addr = ref.ctypes.data + sy * ref.strides[0] + sx * ps
        if variant == MC_PUT:
            out = np.zeros((h, 64), dtype=np.int16)
            dst, dstride = out, 64
        else:
            out = np.zeros((h, w), dtype=ref.dtype)
            dst, dstride = out, out.strides[0]
        s2 = np.ascontiguousarray(src2, dtype=np.int16) if src2 is not None else np.zeros((1, 64), np.int16)
        self._f("mc")(C.c_int(bd), C.c_int(luma), C.c_int(variant), _p(dst), C.c_ssize_t(dstride),
                      C.c_void_p(src_addr), C.c_ssize_t(ref.strides[0]), _p(s2), C.c_ssize_t(s2.shape[1]),
                      C.c_int(h), C.c_int(mx), C.c_int(my), C.c_int(w),
                      C.c_int(denom), C.c_int(wx0), C.c_int(wx1), C.c_int(ox0), C.c_int(ox1))
        return out[:, :w].copy() if variant == MC_PUT else out

    # ---- deblocking: `plane` modified in place; (x, y) is the q0 sample of the first line of the 8-line edge
    def deblock_luma(self, bd, vertical_edge, plane, x, y, beta, tc, no_p, no_q):
        tc_a = (C.c_int * 2)(*tc); p_a = (C.c_uint8 * 2)(*no_p); q_a = (C.c_uint8 * 2)(*no_q)
        addr = plane.ctypes.data + y * plane.strides[0] + x * plane.itemsize
        self._f("deblock_luma")(C.c_int(bd), C.c_int(vertical_edge), C.c_void_p(addr), C.c_ssize_t(plane.strides[0]),
                                C.c_int(beta), tc_a, p_a, q_a)

    def deblock_chroma(self, bd, vertical_edge, plane, x, y, tc, no_p, no_q):
        tc_a = (C.c_int * 2)(*tc); p_a = (C.c_uint8 * 2)(*no_p); q_a = (C.c_uint8 * 2)(*no_q)
        addr = plane.ctypes.data + y * plane.strides[0] + x * plane.itemsize
        self._f("deblock_chroma")(C.c_int(bd), C.c_int(vertical_edge), C.c_void_p(addr), C.c_ssize_t(plane.strides[0]),
                                  tc_a, p_a, q_a)

    # ---- SAO on a w x h block at (x,y): reads `src` (needs a 1-px ring for edge), writes `dst`
    def sao_band(self, bd, dst, src, x, y, w, h, offset_val, band_position):
        ov = np.ascontiguousarray(offset_val, dtype=np.int16)
        self._f("sao_band")(C.c_int(bd), C.c_void_p(dst.ctypes.data + y * dst.strides[0] + x * dst.itemsize),
                            C.c_void_p(src.ctypes.data + y * src.strides[0] + x * src.itemsize),
                            C.c_ssize_t(dst.strides[0]), C.c_ssize_t(src.strides[0]), _p(ov), C.c_int(band_position),
                            C.c_int(w), C.c_int(h))

    def sao_edge(self, bd, restore, dst, src, x, y, w, h, offset_val, eo_class, borders,
                 vert_edge=(0, 0), horiz_edge=(0, 0), diag_edge=(0, 0, 0, 0)):
        ov = np.ascontiguousarray(offset_val, dtype=np.int16)
        b = (C.c_int * 4)(*borders); ve = (C.c_uint8 * 2)(*vert_edge); he = (C.c_uint8 * 2)(*horiz_edge)
        de = (C.c_uint8 * 4)(*diag_edge)
        self._f("sao_edge")(C.c_int(bd), C.c_int(restore),
                            C.c_void_p(dst.ctypes.data + y * dst.strides[0] + x * dst.itemsize),
                            C.c_void_p(src.ctypes.data + y * src.strides[0] + x * src.itemsize),
                            C.c_ssize_t(dst.strides[0]), C.c_ssize_t(src.strides[0]), _p(ov), C.c_int(eo_class), b,
                            C.c_int(w), C.c_int(h), ve, he, de)

    # ---- pure predictors: top/left are 1-D pixel arrays holding [-1 .. 2N-1] (so element 0 is at index 1)
    def pred(self, bd, log2, mode, top, left, c_idx=0):
        n = 1 << log2
        dt = np.uint16 if bd > 8 else np.uint8
        out = np.zeros((n, n), dtype=dt)
        t = np.ascontiguousarray(top, dtype=dt); l = np.ascontiguousarray(left, dtype=dt)
        tp = C.c_void_p(t.ctypes.data + t.itemsize); lp = C.c_void_p(l.ctypes.data + l.itemsize)
        if mode == 0:
            self._f("pred_planar")(C.c_int(bd), C.c_int(log2), _p(out), tp, lp, C.c_ssize_t(out.strides[0]))
        elif mode == 1:
            self._f("pred_dc")(C.c_int(bd), C.c_int(log2), _p(out), tp, lp, C.c_ssize_t(out.strides[0]), C.c_int(c_idx))
        else:
            self._f("pred_angular")(C.c_int(bd), C.c_int(log2), _p(out), tp, lp, C.c_ssize_t(out.strides[0]),
                                    C.c_int(c_idx), C.c_int(mode))
        return out

    # ---- full intra_pred() on a picture (list of 3 plane arrays, modified in place)
    def intra_pred(self, bd, planes, width, height, x0, y0, log2, c_idx, mode, cands, chroma_format_idc=1,
                   log2_ctb_size=6, log2_min_tb_size=2, log2_min_pu_size=2, strong=1, smoothing_disabled=0,
                   constrained=0, is_intra=None):
        pic = IntraPic()
        for i in range(3):
            pic.data[i] = planes[i].ctypes.data
            pic.linesize[i] = planes[i].strides[0]
        pic.width, pic.height, pic.chroma_format_idc = width, height, chroma_format_idc
        pic.log2_ctb_size, pic.log2_min_tb_size, pic.log2_min_pu_size = log2_ctb_size, log2_min_tb_size, log2_min_pu_size
        pic.strong_intra_smoothing, pic.intra_smoothing_disabled, pic.constrained_intra_pred = strong, smoothing_disabled, constrained
        keep = None
        if is_intra is not None:
            keep = np.ascontiguousarray(is_intra, dtype=np.uint8)
            pic.is_intra = keep.ctypes.data
        bl, lf, ul, up, ur = cands
        self._f("intra_pred")(C.c_int(bd), C.byref(pic), C.c_int(x0), C.c_int(y0), C.c_int(log2), C.c_int(c_idx),
                              C.c_int(mode), C.c_int(bl), C.c_int(lf), C.c_int(ul), C.c_int(up), C.c_int(ur))


def _aligned_copy(a, align=64):
    raw = np.empty(a.nbytes + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + a.nbytes].view(a.dtype)
    out[:] = a
    return out


_cache = {}


def load(which):
    """which: 'oracle' (restatement, always buildable), 'ref' (reference build) or 'sse'. Returns None if absent."""
    if which in _cache:
        return _cache[which]
    if which == "oracle":
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(path)
        lib = OracleLib(path, "ohor_")
    elif which == "ref":
        path = os.path.join(HERE, "_ref", "libhevcref.so")
        lib = OracleLib(path, "ohref_") if os.path.exists(path) else None
    elif which == "sse":
        path = os.path.join(HERE, "_ref", "libhevcref_sse.so")
        lib = C.CDLL(path) if os.path.exists(path) else None
    else:
        raise ValueError(which)
    _cache[which] = lib
    return lib


def restore_tqb_pixels(frame_plane, deblocked_plane, x0, y0, width, height, is_pcm, log2_min_pu_size, hshift, vshift,
                       exact_reference=True):
    """numpy restatement of restore_tqb_pixels (hevc_filter.c:163-193), called by sao_filter_CTB right after each SAO table
    call (:275,:316): min-PU blocks flagged in `is_pcm` (2-D array, [y_pu, x_pu]) get their deblocked samples back.
    x0/y0 are the CTB origin in LUMA samples, width/height the size of the filtered block in samples of THIS plane --
    exactly the reference's argument mix, which bounds the PU walk to the CTB's first half for subsampled chroma; and each
    row copies `len = min_pu_size >> hshift` BYTES (memcpy(src, dst, len), :176,:184), half a PU row of 16-bit samples.
    exact_reference=False is the standard's behaviour (every sample of a flagged PU, H.265 8.7.1).
    Pinned only at bitstream level (tests/test_stream_gpu.py against the untouched decoder): the function is static in
    the reference, so there is no kernel-level call-through."""
    l = log2_min_pu_size
    n, ln = (1 << l) >> vshift, (1 << l) >> hshift
    if exact_reference:
        yr, xr = range(y0 >> l, (y0 + height) >> l), range(x0 >> l, (x0 + width) >> l)
        ln = ln // frame_plane.itemsize
    else:
        yr = range(y0 >> l, -(-(y0 + (height << vshift)) >> l))
        xr = range(x0 >> l, -(-(x0 + (width << hshift)) >> l))
    for y in yr:
        for x in xr:
            if y < is_pcm.shape[0] and x < is_pcm.shape[1] and is_pcm[y, x]:
                ys, xs = (y << l) >> vshift, (x << l) >> hshift
                frame_plane[ys:ys + n, xs:xs + ln] = deblocked_plane[ys:ys + n, xs:xs + ln]


# ------------------------------------------------------------------------------------------------ SHVC inter-layer up-sampling
SHVC_DEFAULT, SHVC_X2, SHVC_X1_5, SHVC_SNR = 0, 1, 2, 3      # UpsamplInf.idx, hevc.h:340-345


def shvc_params(bl_w, bl_h, el_w, el_h, win=(0, 0, 0, 0), phase_align=0):
    """UpsamplInf as set_sps derives it (hevc.c:445-499) from the base-layer size, the enhancement-layer size and its
    scaled_ref_layer_window (left, right, top, bottom): [addXLum, addYLum, scaleXLum, scaleYLum, addXCr, addYCr, scaleXCr, scaleYCr, idx]."""
    height_el = el_h - win[3] - win[2]
    width_el = el_w - win[0] - win[1]
    sx = ((bl_w << 16) + (width_el >> 1)) // width_el
    sy = ((bl_h << 16) + (height_el >> 1)) // height_el
    phase_x = phase_y = phase_align << 1
    add_x = ((phase_x * sx + 2) >> 2) + (1 << 11)
    add_y = ((phase_y * sy + 2) >> 2) + (1 << 11)
    add_xc = (((0 + phase_align) * sx + 2) >> 2) + (1 << 11)
    add_yc = (((1 + phase_align) * sy + 2) >> 2) + (1 << 11)
    idx = SHVC_SNR if (sx, sy) == (65536, 65536) else SHVC_X2 if (sx, sy) == (32768, 32768) else SHVC_X1_5 if (sx, sy) == (43691, 43691) else SHVC_DEFAULT
    return np.array([add_x, add_y, sx, sy, add_xc, add_yc, sx, sy, idx], dtype=np.int32)


class _DrvPic(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("linesize", C.c_int32 * 3)]


def _drv_pic(planes):
    p = _DrvPic()
    for c in range(3):
        p.data[c] = planes[c].ctypes.data
        p.linesize[c] = planes[c].strides[0]
    return p


def padded_planes(planes, pad=64):
    """copies of the planes inside `pad` samples of replicated border (the decoder's frame buffers have such edges and the
    reference's emulated_edge_up_h writes a few samples into them); returns (padded arrays, views of the picture area)"""
    big = [np.ascontiguousarray(np.pad(p, pad, mode="edge")) for p in planes]
    return big, [b[pad:-pad, pad:-pad] for b in big]


def shvc_reference(path, mode, bd, el_planes, el_w, el_h, bl_planes, bl_w, bl_h, win, up, log2_ctb=6, conf=(0, 0), hooks=(None, None),
                   frame_helper=None):
    """Runs the reference's slots (oracle/shvc_driver.c in _ref/libhevcref.so): mode "frame" = upsample_base_layer_frame,
    "blocks" = the per-CTB sequences of upsample_block_luma / upsample_block_mc.  Planes are modified in place."""
    L = C.CDLL(path)
    el, bl = _drv_pic(el_planes), _drv_pic(bl_planes)
    win = np.asarray(win, np.int32); up = np.asarray(up, np.int32); conf = np.asarray(conf, np.int32)
    h0 = C.cast(hooks[0], C.c_void_p) if hooks[0] else None
    h1 = C.cast(hooks[1], C.c_void_p) if hooks[1] else None
    if mode == "frame":
        fh = C.cast(frame_helper, C.c_void_p) if frame_helper else None
        return L.ohref_shvc_frame(C.c_int(bd), C.byref(el), el_w, el_h, C.byref(bl), bl_w, bl_h, _p(win), _p(up), h0, h1, fh)
    return L.ohref_shvc_blocks(C.c_int(bd), log2_ctb, C.byref(el), el_w, el_h, C.byref(bl), bl_w, bl_h, _p(win), _p(conf), _p(up), h0, h1)


def shvc_reference_not_a_function_of_its_inputs(path, el_w, el_h, bl_w, bl_h, phase_align=0, log2_ctb=6, seed=1):
    """Whether the reference's CTB-by-CTB sequence (upsample_block_luma / _mc, hevc_filter.c:1175-1310) computes, for this pair of sizes, samples
    from memory that is not the base-layer picture: it sizes a CTB's source window as ((ePb + 1 or 2) * scale + add) >> 16 columns / rows plus at
    most MAX_EDGE / MAX_EDGE_CR more (:1194-1210, :1262-1283; the chroma window is positioned with the LUMA offsets addXLum / addYLum), and for
    some ratios that is one column or row short of what the filter taps of the CTB's last columns / rows read.  A missing column lies in the
    base-layer frame buffer's edge (zeros, or what emulated_edge_up_h wrote there for another CTB, possibly of an earlier picture of that
    buffer); a missing row in the thread's scratch buffer (what the previous slot calls - of another CTB, in the decoder's on-demand order -
    left there).  Found by running the sequence three times on one random picture: inside a replicated border and inside a border of zeros,
    with the scratch buffer refilled with three different values in front of every plane of every CTB (oracle/shvc_driver.c).  Returns the
    number of samples per plane that differ between the runs: the decoder's output there depends on the order CTBs were resampled in and on
    the buffers' history, not on the stream alone."""
    rng = np.random.default_rng(seed)
    bl = [rng.integers(1, 256, size=(bl_h >> (c > 0), bl_w >> (c > 0))).astype(np.uint8) for c in range(3)]
    up = shvc_params(bl_w, bl_h, el_w, el_h, (0, 0, 0, 0), phase_align=phase_align)
    res = []
    keep = {k: os.environ.get(k) for k in ("OHREF_SHVC_POISON", "OHREF_SHVC_POISON_EACH")}
    try:
        for mode, poison in (("edge", "0"), ("constant", "1357"), ("edge", "-2468")):
            os.environ["OHREF_SHVC_POISON"] = poison
            os.environ["OHREF_SHVC_POISON_EACH"] = "2"
            big = [np.ascontiguousarray(np.pad(p, 64, mode=mode)) for p in bl]
            blv = [b[64:-64, 64:-64] for b in big]
            el = [np.zeros((el_h >> (c > 0), el_w >> (c > 0)), np.uint8) for c in range(3)]
            _, view = padded_planes(el)
            shvc_reference(path, "blocks", 8, view, el_w, el_h, blv, bl_w, bl_h, (0, 0, 0, 0), up, log2_ctb=log2_ctb)
            res.append([v.copy() for v in view])
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return [int(np.count_nonzero((a != b) | (b != c))) for a, b, c in zip(*res)]


def shvc_upsample_frame(oracle_lib_path, bd, el_planes, el_w, el_h, bl_planes, bl_w, bl_h, win, up, block_slots=0):
    """our restatement (ohor_shvc_upsample_frame in liboracle.so); el_planes are modified in place"""
    L = C.CDLL(oracle_lib_path)
    elp = (C.c_void_p * 3)(*[p.ctypes.data for p in el_planes]); els = (C.c_int32 * 3)(*[p.strides[0] for p in el_planes])
    blp = (C.c_void_p * 3)(*[p.ctypes.data for p in bl_planes]); bls = (C.c_int32 * 3)(*[p.strides[0] for p in bl_planes])
    win = np.asarray(win, np.int32); up = np.asarray(up, np.int32)
    L.ohor_shvc_upsample_frame(C.c_int(bd), C.c_int(block_slots), elp, els, el_w, el_h, blp, bls, bl_w, bl_h, _p(win), _p(up))


# ---- boundary strengths (ohor_boundary_strengths; reference side: the tap in oracle/null_hooks.c) ----
BS_FIELD = np.dtype([("mv", np.int16, (2, 2)), ("poc", np.int32, (2,)), ("pred_flag", np.uint32)])      # oh_bs_field, 20 bytes
BS_CALL = np.dtype([("x0", np.uint16), ("y0", np.uint16), ("log2_size", np.uint8), ("flags", np.uint8), ("reserved", np.uint16)])


class BsGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("min_pu_width", "log2_min_pu_size", "min_tb_width", "log2_min_tb_size", "log2_ctb_size", "bs_width",
                                         "loop_filter_across_tiles")]


def boundary_strengths(oracle_lib_path, geom, mvf, cbf_luma, calls, n_bs):
    """our restatement over one picture's calls -> (vertical_bs, horizontal_bs), n_bs entries each, zero where no call writes"""
    L = C.CDLL(oracle_lib_path)
    assert mvf.dtype == BS_FIELD and calls.dtype == BS_CALL and cbf_luma.dtype == np.uint8
    g = BsGeom(**geom)
    v, h = np.zeros(n_bs, np.uint8), np.zeros(n_bs, np.uint8)
    mvf, cbf_luma, calls = np.ascontiguousarray(mvf), np.ascontiguousarray(cbf_luma), np.ascontiguousarray(calls)
    L.ohor_boundary_strengths(C.byref(g), _p(mvf), _p(cbf_luma), _p(calls), C.c_int(len(calls)), _p(v), _p(h))
    return v, h


class _BsFrame(C.Structure):
    _fields_ = [("calls", C.c_void_p), ("ncalls", C.c_int32), ("mvf", C.c_void_p), ("cbf_luma", C.c_void_p), ("vertical_bs", C.c_void_p),
                ("horizontal_bs", C.c_void_p), ("n_vertical", C.c_int32), ("n_horizontal", C.c_int32)] + \
               [(n, C.c_int32) for n in ("min_pu_width", "min_pu_height", "log2_min_pu_size", "min_tb_width", "min_tb_height", "log2_min_tb_size",
                                         "log2_ctb_size", "bs_width", "width", "height", "loop_filter_across_tiles")]


def bs_tap(null_lib, on):
    """oracle/_ref/libopenhevc_null.so: start (a fresh log) / stop logging the reference's ff_hevc_deblocking_boundary_strengths calls"""
    null_lib.ohnull_bs_tap(C.c_int(1 if on else 0))


def bs_tap_fetch(null_lib):
    """-> None (no call since bs_tap) or dict(calls, mvf, cbf_luma, vertical_bs, horizontal_bs (the reference's), geom, n_bs): copies"""
    f = _BsFrame()
    if null_lib.ohnull_bs_fetch(C.byref(f)) != 0:
        return None

    def arr(ptr, count, dt):
        return np.frombuffer((C.c_uint8 * (count * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy()
    geom = dict(min_pu_width=f.min_pu_width, log2_min_pu_size=f.log2_min_pu_size, min_tb_width=f.min_tb_width, log2_min_tb_size=f.log2_min_tb_size,
                log2_ctb_size=f.log2_ctb_size, bs_width=f.bs_width, loop_filter_across_tiles=f.loop_filter_across_tiles)
    return dict(calls=arr(f.calls, f.ncalls, BS_CALL), mvf=arr(f.mvf, f.min_pu_width * f.min_pu_height, BS_FIELD),
                cbf_luma=arr(f.cbf_luma, f.min_tb_width * f.min_tb_height, np.uint8), vertical_bs=arr(f.vertical_bs, f.n_vertical, np.uint8),
                horizontal_bs=arr(f.horizontal_bs, f.n_horizontal, np.uint8), geom=geom, n_bs=f.n_vertical, width=f.width, height=f.height,
                min_pu_height=f.min_pu_height, min_tb_height=f.min_tb_height)
