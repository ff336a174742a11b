"""ctypes binding of libohevc_hip.so (include/ohevc_hip.h).  No compute happens in Python."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libohevc_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_NODEV, ERR_STATE = 0, -1, -2, -3, -4
TU_IDCT, TU_DC, TU_DST4, TU_SKIP, TU_SKIP_RDPCM_H, TU_SKIP_RDPCM_V, TU_BYPASS, TU_BYPASS_RDPCM_H, TU_BYPASS_RDPCM_V, TU_PCM, TU_CROSS = range(11)


class OhevcError(RuntimeError):
    pass


class Plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


# numpy view of ohevc_tu_job (16 bytes)
TU_JOB = np.dtype([("x", "<u2"), ("y", "<u2"), ("plane", "u1"), ("reserved0", "u1"), ("dc", "<i2"),
                   ("coeff_off", "<u4"), ("reserved1", "<u4")])
assert TU_JOB.itemsize == 16


def build(verbose=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(HERE, "csrc")] + ([] if verbose else ["-s"]), check=True)


def bind_prototypes(lib):
    """Declare the result / argument types of the entry points that need them on a loaded libohevc_hip.so."""
    lib.ohevc_last_error.restype = C.c_char_p
    lib.ohevc_version.restype = C.c_char_p
    lib.ohevc_tu_kernel_name.restype = C.c_char_p
    lib.ohevc_tu_kernel_name.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.ohevc_device_count.restype = C.c_int
    lib.ohevc_set_device.argtypes = [C.c_int]
    lib.ohevc_dev_tu_batch.restype = C.c_int
    lib.ohevc_dev_tu_batch.argtypes = [C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return lib


_lib = None


def load_library():
    """Load libohevc_hip.so; raise if it is missing (never fall back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime (libamdhip64.so.7, same SONAME as /opt/rocm's).  Whichever copy is loaded
    # first serves the whole process, and torch only works on its own copy -- so in a Python process torch must be
    # imported BEFORE this library is dlopen'ed.  (A plain C host links /opt/rocm's runtime and never sees torch.)
    import torch  # noqa: F401
    # OHEVC_LAB_LIBRARY=1 (A/B sessions of tools/ only): the lab build, which carries the measurement kernels next to the shipped ones
    path = os.path.join(HERE, "libohevc_hip_lab.so") if os.environ.get("OHEVC_LAB_LIBRARY") == "1" else LIB_PATH
    if not os.path.exists(path):
        raise OhevcError(f"{path} is missing: run `make -C openhevc_amd/csrc` (or __graft_entry__.build()) first")
    lib = bind_prototypes(C.CDLL(path))
    _lib = lib
    return lib


# every symbol include/ohevc_hip.h declares (checked by the CPU-side ABI test)
EXPORTED_SYMBOLS = ["ohevc_dev_tu_batch", "ohevc_dev_tu_multi", "ohevc_last_error", "ohevc_device_count", "ohevc_set_device",
                    "ohevc_tu_kernel_name", "ohevc_version"]


def check(rc):
    if rc != OK:
        raise OhevcError(f"ohevc error {rc}: {load_library().ohevc_last_error().decode()}")


def planes_of(tensors):
    """Build ohevc_plane[3] from up to three 2-D torch CUDA tensors (uint8 / uint16|int16 storage)."""
    arr = (Plane * 3)()
    for i, t in enumerate(tensors):
        if t is None:
            continue
        assert t.dim() == 2 and t.stride(1) == 1
        arr[i].data = t.data_ptr()
        arr[i].stride = t.stride(0) * t.element_size()
        arr[i].width, arr[i].height = t.shape[1], t.shape[0]
    return arr


def dev_tu_batch(planes, bit_depth, log2_size, kind, jobs_ptr, njobs, coeffs_ptr, stream=0):
    check(load_library().ohevc_dev_tu_batch(planes, bit_depth, log2_size, kind, C.c_void_p(jobs_ptr), njobs,
                                            C.c_void_p(coeffs_ptr), C.c_void_p(stream)))


# ---------------------------------------------------------------- MC / deblock / SAO / intra job records
MC_BI, MC_WEIGHTED = 1, 2
MC_JOB = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("plane", "u1"), ("flags", "u1"),
                   ("sx0", "<i2"), ("sy0", "<i2"), ("sx1", "<i2"), ("sy1", "<i2"),
                   ("mx0", "u1"), ("my0", "u1"), ("mx1", "u1"), ("my1", "u1"),
                   ("ref0", "i1"), ("ref1", "i1"), ("denom", "u1"), ("reserved", "u1"),
                   ("wx0", "<i2"), ("wx1", "<i2"), ("ox0", "<i2"), ("ox1", "<i2")])
assert MC_JOB.itemsize == 32

DBK_VERTICAL_EDGE, DBK_NO_P0, DBK_NO_P1, DBK_NO_Q0, DBK_NO_Q1 = 1, 2, 4, 8, 16
DBK_JOB = np.dtype([("x", "<u2"), ("y", "<u2"), ("plane", "u1"), ("flags", "u1"), ("beta", "u1"), ("reserved0", "u1"),
                    ("tc", "<i2", (2,)), ("reserved1", "<u4")])
assert DBK_JOB.itemsize == 16

SAO_BAND, SAO_EDGE = 1, 2
SAO_JOB = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "<u2"), ("h", "<u2"), ("plane", "u1"), ("type", "u1"),
                    ("klass", "u1"), ("borders", "u1"), ("restore", "u1"), ("edges", "u1"),
                    ("offset_val", "<i2", (5,)), ("quirks", "u1"), ("reserved", "u1", (7,))])
assert SAO_JOB.itemsize == 32

INTRA_BOTTOM_LEFT, INTRA_LEFT, INTRA_UP_LEFT, INTRA_UP, INTRA_UP_RIGHT = 1, 2, 4, 8, 16
INTRA_NO_SMOOTHING, INTRA_STRONG, INTRA_LUMA_EDGE = 32, 64, 128
INTRA_JOB = np.dtype([("x", "<u2"), ("y", "<u2"), ("plane", "u1"), ("log2_size", "u1"), ("mode", "u1"), ("flags", "u1"),
                      ("bottom_left_size", "u1"), ("top_right_size", "u1"), ("flags2", "u1"), ("log2_ctb_size", "u1"), ("cip_index", "<u4")])
assert INTRA_JOB.itemsize == 16
INTRA2_CIP = 1
INTRA_CIP = np.dtype([("top_bits", "u1", (9,)), ("left_bits", "u1", (9,)), ("size_max_x", "u1"), ("size_max_y", "u1"),
                      ("x0_nonzero", "u1"), ("y0_nonzero", "u1"), ("reserved", "u1", (10,))])
assert INTRA_CIP.itemsize == 32

EXPORTED_SYMBOLS += ["ohevc_dev_mc_batch", "ohevc_dev_mc_batch_small", "ohevc_dev_deblock_batch", "ohevc_dev_sao_batch", "ohevc_dev_sao_batch_lagged", "ohevc_dev_intra_batch", "ohevc_dev_intra_batch_cip",
                     "ohevc_intra_make_job_cip", "ohevc_rec_intra_cip", "ohevc_tables_intra_pred_cip"]


def planes_table(list_of_plane_triples):
    """numpy bytes of a slot-major ohevc_plane[n*3] table (for the MC reference-picture table)."""
    n = len(list_of_plane_triples)
    arr = (Plane * (3 * n))()
    for s, triple in enumerate(list_of_plane_triples):
        for i, t in enumerate(triple):
            if t is None:
                continue
            arr[3 * s + i].data = t.data_ptr()
            arr[3 * s + i].stride = t.stride(0) * t.element_size()
            arr[3 * s + i].width, arr[3 * s + i].height = t.shape[1], t.shape[0]
    return np.frombuffer(bytes(arr), dtype=np.uint8).copy()


def dev_mc_batch(dst_planes, refs_ptr, n_slots, bit_depth, jobs_ptr, njobs, stream=0):
    lib = load_library()
    check(lib.ohevc_dev_mc_batch(dst_planes, C.c_void_p(refs_ptr), C.c_int(n_slots), C.c_int(bit_depth),
                                 C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(stream)))


def dev_mc_batch_bounded(dst_planes, refs_ptr, n_slots, bit_depth, jobs_ptr, njobs, max_w, max_h, stream=0):
    check(load_library().ohevc_dev_mc_batch_bounded(dst_planes, C.c_void_p(refs_ptr), C.c_int(n_slots), C.c_int(bit_depth),
                                                    C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_int(max_w), C.c_int(max_h), C.c_void_p(stream)))


def dev_mc_batch_small(dst_planes, refs_ptr, n_slots, bit_depth, jobs_ptr, njobs, stream=0):
    lib = load_library()
    check(lib.ohevc_dev_mc_batch_small(dst_planes, C.c_void_p(refs_ptr), C.c_int(n_slots), C.c_int(bit_depth),
                                       C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(stream)))


def dev_deblock_batch(planes, bit_depth, jobs_ptr, njobs, stream=0):
    check(load_library().ohevc_dev_deblock_batch(planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(stream)))


class DbkMaps(C.Structure):
    """ohevc_dbk_maps (include/ohevc_hip.h)"""
    _fields_ = [("vertical_bs", C.c_void_p), ("horizontal_bs", C.c_void_p), ("qp_y_tab", C.c_void_p), ("deblock", C.c_void_p), ("is_pcm", C.c_void_p),
                ("bs_width", C.c_int32), ("min_cb_width", C.c_int32), ("deblock_stride", C.c_int32), ("min_pu_width", C.c_int32), ("min_pu_height", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("log2_ctb_size", C.c_int32), ("log2_min_cb_size", C.c_int32), ("log2_min_pu_size", C.c_int32),
                ("chroma_format_idc", C.c_int32), ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32)]


def dev_deblock_maps(planes, bit_depth, maps, vertical, stream=0):
    check(load_library().ohevc_dev_deblock_maps(planes, C.c_int(bit_depth), C.byref(maps), C.c_int(vertical), C.c_void_p(stream)))


EXPORTED_SYMBOLS += ["ohevc_dev_mc_batch_bounded", "ohevc_pic_export", "ohevc_pic_import", "ohevc_dev_deblock_maps", "ohevc_rec_deblock_maps", "ohevc_ctx_has_device", "ohevc_debug_set_filters_on_device"]


def dev_sao_batch(dst_planes, src_planes, bit_depth, jobs_ptr, njobs, stream=0):
    check(load_library().ohevc_dev_sao_batch(dst_planes, src_planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(stream)))


def dev_sao_batch_sorted(dst_planes, src_planes, bit_depth, jobs_ptr, n_wide, n_other, stream=0):
    check(load_library().ohevc_dev_sao_batch_sorted(dst_planes, src_planes, src_planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr), C.c_int(n_wide), C.c_int(n_other),
                                                    None, C.c_void_p(stream)))


EXPORTED_SYMBOLS += ["ohevc_dev_sao_batch_sorted", "ohevc_sao_job_is_wide", "ohevc_dev_intra_recon_batch", "ohevc_debug_set_fuse_intra"]


class SaoBypass(C.Structure):
    """ohevc_sao_bypass (include/ohevc_hip.h)"""
    _fields_ = [("map", C.c_void_p), ("stride", C.c_int32), ("log2_min_pu_size", C.c_int32),
                ("chroma_hshift", C.c_int32), ("chroma_vshift", C.c_int32), ("exact_reference", C.c_int32)]


EXPORTED_SYMBOLS += ["ohevc_dev_sao_batch_bypass", "ohevc_frame_set_bypass_map", "ohevc_tables_set_bypass_map", "ohevc_tables_set_concurrent", "ohevc_tables_cross_component", "ohevc_rec_tu_cross"]


def dev_sao_batch_bypass(dst_planes, src_planes, bit_depth, jobs_ptr, njobs, map_ptr, map_stride, log2_min_pu_size,
                         chroma_hshift, chroma_vshift, exact_reference=1, stream=0):
    bp = SaoBypass(map_ptr, map_stride, log2_min_pu_size, chroma_hshift, chroma_vshift, exact_reference)
    check(load_library().ohevc_dev_sao_batch_bypass(dst_planes, src_planes, src_planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr),
                                                    C.c_int(njobs), C.byref(bp), C.c_void_p(stream)))


def dev_intra_batch_cip(planes, bit_depth, jobs_ptr, njobs, cip_ptr, stream=0):
    check(load_library().ohevc_dev_intra_batch_cip(planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(cip_ptr), C.c_void_p(stream)))


def dev_intra_batch(planes, bit_depth, jobs_ptr, njobs, stream=0):
    check(load_library().ohevc_dev_intra_batch(planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(stream)))


def dev_intra_recon_sorted(planes, bit_depth, jobs_ptr, residuals_ptr, count_by_size, coeffs_ptr, stream=0):
    """jobs sorted by size (count_by_size[k] blocks of (4 << k) samples); residuals_ptr 0 = prediction only."""
    counts = (C.c_int32 * 4)(*count_by_size)
    check(load_library().ohevc_dev_intra_recon_sorted(planes, C.c_int(bit_depth), C.c_void_p(jobs_ptr), C.c_void_p(residuals_ptr or None), counts,
                                                      C.c_void_p(coeffs_ptr or None), C.c_void_p(stream)))


EXPORTED_SYMBOLS += ["ohevc_dev_intra_recon_sorted", "ohevc_dev_intra_recon_batch", "ohevc_dev_intra_chain", "ohevc_intra_chain_max_waves", "ohevc_host_pin",
                     "ohevc_host_unpin_all", "ohevc_pic_download_planes", "ohevc_frames_transport_create", "ohevc_frames_transport_mode", "ohevc_frames_transport_finish",
                     "ohevc_frames_transport_destroy", "ohevc_frames_transport_stats", "ohevc_frames_transport_selftest", "ohevc_host_unpin", "ohevc_host_alloc", "ohevc_host_free", "ohevc_host_block_pinned", "ohevc_host_alloc_pins", "ohevc_pic_download_queue", "ohevc_debug_set_chain_handover", "ohevc_intra_chain_workgroup_waves", "ohevc_intra_chain_max_levels"]


class BsMaps(C.Structure):
    """ohevc_bs_maps (include/ohevc_hip.h): the motion field and cbf_luma map ohevc_dev_boundary_strengths reads (DEVICE pointers)"""
    _fields_ = [("mvf", C.c_void_p), ("mvf_stride", C.c_int32), ("off_mv", C.c_int32), ("off_poc", C.c_int32), ("off_pred_flag", C.c_int32),
                ("pred_flag_bytes", C.c_int32), ("cbf_luma", C.c_void_p), ("min_pu_width", C.c_int32), ("min_pu_height", C.c_int32),
                ("log2_min_pu_size", C.c_int32), ("min_tb_width", C.c_int32), ("min_tb_height", C.c_int32), ("log2_min_tb_size", C.c_int32),
                ("log2_ctb_size", C.c_int32), ("bs_width", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("loop_filter_across_tiles", C.c_int32)]


BS_CALL = np.dtype([("x0", "<u2"), ("y0", "<u2"), ("log2_size", "u1"), ("flags", "u1"), ("reserved", "<u2")])       # ohevc_bs_call
MOTION_GRID_ENTRY = 20                                                                                             # OHEVC_MOTION_GRID_ENTRY
assert BS_CALL.itemsize == 8


def dev_boundary_strengths(maps, calls_ptr, ncalls, vertical_bs_ptr, horizontal_bs_ptr, stream=0):
    check(load_library().ohevc_dev_boundary_strengths(C.byref(maps), C.c_void_p(calls_ptr), C.c_int(ncalls), C.c_void_p(vertical_bs_ptr),
                                                      C.c_void_p(horizontal_bs_ptr), C.c_void_p(stream)))


def dev_motion_grid(jobs_ptr, njobs, grid_ptr, grid_width, grid_height, log2_unit, stream=0):
    check(load_library().ohevc_dev_motion_grid(C.c_void_p(jobs_ptr), C.c_int(njobs), C.c_void_p(grid_ptr), C.c_int(grid_width), C.c_int(grid_height),
                                               C.c_int(log2_unit), C.c_void_p(stream)))


def dev_motion_grid2(jobs_ptr, njobs, more_ptr, nmore, grid_ptr, grid_width, grid_height, log2_unit, stream=0):
    check(load_library().ohevc_dev_motion_grid2(C.c_void_p(jobs_ptr or None), C.c_int(njobs), C.c_void_p(more_ptr or None), C.c_int(nmore), C.c_void_p(grid_ptr),
                                                C.c_int(grid_width), C.c_int(grid_height), C.c_int(log2_unit), C.c_void_p(stream)))


EXPORTED_SYMBOLS += ["ohevc_dev_boundary_strengths", "ohevc_dev_motion_grid", "ohevc_dev_motion_grid2", "ohevc_dev_copy", "ohevc_dev_zero", "ohevc_dev_upsample_picture", "ohevc_frame_keep_motion", "ohevc_tables_keep_motion", "ohevc_rec_bs_call",
                     "ohevc_rec_deblock_maps_bs", "ohevc_tables_bs_wanted", "ohevc_tables_bs_call", "ohevc_tables_bs_calls", "ohevc_rec_bs_calls"]


class IntraGeom(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("chroma_format_idc", C.c_int32), ("log2_ctb_size", C.c_int32),
                ("log2_min_tb_size", C.c_int32), ("strong_intra_smoothing", C.c_int32), ("intra_smoothing_disabled", C.c_int32),
                ("constrained_intra_pred", C.c_int32)]


EXPORTED_SYMBOLS += ["ohevc_intra_make_job", "ohevc_hevcdsp_init_hip", "ohevc_videodsp_init_hip", "ohevc_tables_bind",
                     "ohevc_tables_register_picture", "ohevc_tables_unregister_picture", "ohevc_tables_begin_frame",
                     "ohevc_tables_end_frame", "ohevc_tables_status", "ohevc_tables_forget", "ohevc_tables_emulate_filter_lag", "ohevc_tables_intra_pred", "ohevc_pic_info"]


def intra_make_job_cip(geom, log2_min_pu_size, is_intra_map, x0, y0, log2_size, c_idx, mode, cands):
    """is_intra_map: uint8 [pu_h, pu_w], 1 = intra.  Returns (job[1], cip[1])."""
    out = np.zeros(1, INTRA_JOB); cip = np.zeros(1, INTRA_CIP)
    m = np.ascontiguousarray(is_intra_map, dtype=np.uint8)
    bl, lf, ul, up, ur = cands
    check(load_library().ohevc_intra_make_job_cip(C.byref(geom), C.c_int(log2_min_pu_size), m.ctypes.data_as(C.c_void_p), C.c_ssize_t(1),
                                                  C.c_int(1), C.c_int(x0), C.c_int(y0), C.c_int(log2_size), C.c_int(c_idx), C.c_int(mode),
                                                  C.c_int(bl), C.c_int(lf), C.c_int(ul), C.c_int(up), C.c_int(ur),
                                                  out.ctypes.data_as(C.c_void_p), cip.ctypes.data_as(C.c_void_p)))
    return out, cip


def intra_make_job(geom, x0, y0, log2_size, c_idx, mode, cands):
    """cands = (bottom_left, left, up_left, up, up_right) as in HEVClc->na.  Returns a 1-element INTRA_JOB array."""
    out = np.zeros(1, INTRA_JOB)
    bl, lf, ul, up, ur = cands
    check(load_library().ohevc_intra_make_job(C.byref(geom), C.c_int(x0), C.c_int(y0), C.c_int(log2_size), C.c_int(c_idx),
                                              C.c_int(mode), C.c_int(bl), C.c_int(lf), C.c_int(ul), C.c_int(up), C.c_int(ur),
                                              out.ctypes.data_as(C.c_void_p)))
    return out


# ---------------------------------------------------------------- ctx layer (include/ohevc_ctx.h)
class FrameStats(C.Structure):
    _fields_ = [("launches", C.c_int32), ("intra_levels", C.c_int32), ("upload_bytes", C.c_int64),
                ("n_tu", C.c_int32), ("n_mc", C.c_int32), ("n_intra", C.c_int32), ("n_dbk", C.c_int32), ("n_sao", C.c_int32),
                ("chose_ctbs", C.c_int32), ("reserved", C.c_int32), ("alg_bytes", C.c_int64)]


EXPORTED_SYMBOLS += ["ohevc_dev_levels", "ohevc_dev_ctbs", "ohevc_frame_abort", "ohevc_level_phase_workgroups", "ohevc_ctx_create_shared", "ohevc_ctx_store_id"]
EXPORTED_SYMBOLS += ["ohevc_ctx_create", "ohevc_ctx_destroy", "ohevc_ctx_stream", "ohevc_ctx_sync", "ohevc_pic_alloc",
                     "ohevc_pic_release", "ohevc_pic_adopt", "ohevc_pic_upload", "ohevc_pic_download", "ohevc_pic_planes", "ohevc_frame_begin",
                     "ohevc_rec_tu", "ohevc_rec_mc", "ohevc_rec_intra", "ohevc_rec_deblock", "ohevc_rec_sao",
                     "ohevc_frame_reconstruct", "ohevc_frame_end", "ohevc_frame_get_stats", "ohevc_rec_mc_bulk",
                     "ohevc_rec_intra_bulk", "ohevc_rec_tu_bulk", "ohevc_rec_deblock_bulk", "ohevc_rec_sao_bulk"]


class UpsampleParams(C.Structure):
    """ohevc_upsample_params (include/ohevc_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in ("el_width", "el_height", "bl_width", "bl_height", "win_left", "win_right", "win_top", "win_bottom",
                                         "add_x_luma", "add_y_luma", "scale_x_luma", "scale_y_luma",
                                         "add_x_chroma", "add_y_chroma", "scale_x_chroma", "scale_y_chroma", "idx", "block_slots")]


def upsample_params(el_w, el_h, bl_w, bl_h, win, up, block_slots):
    """win = (left, right, top, bottom); up = the reference's UpsamplInf as 9 ints (addXLum, addYLum, scaleXLum, scaleYLum, addXCr,
    addYCr, scaleXCr, scaleYCr, idx)"""
    u = [int(v) for v in up]
    return UpsampleParams(el_w, el_h, bl_w, bl_h, *[int(v) for v in win], u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], u[8], int(block_slots))


UPSAMPLE_TAP = np.dtype([("pos", "<i2"), ("phase", "u1"), ("reserved", "u1")])


def upsample_maps(params, plane):
    """ohevc_upsample_make_maps: (cols, col_of, rows, src_cols, src_rows) of one plane as numpy arrays"""
    w = params.el_width >> (1 if plane else 0)
    h = params.el_height >> (1 if plane else 0)
    cols, col_of, rows = np.zeros(w, UPSAMPLE_TAP), np.zeros(w, np.int16), np.zeros(h, UPSAMPLE_TAP)
    sc, sr = C.c_int(), C.c_int()
    check(load_library().ohevc_upsample_make_maps(C.byref(params), plane, cols.ctypes.data_as(C.c_void_p), col_of.ctypes.data_as(C.c_void_p),
                                                  rows.ctypes.data_as(C.c_void_p), C.byref(sc), C.byref(sr)))
    return cols, col_of, rows, sc.value, sr.value


def dev_upsample_plane(dst_tensor, src_tensor, bit_depth, chroma, cols_ptr, col_of_ptr, rows_ptr, src_cols, src_rows, stream=0):
    d, s = planes_of([dst_tensor]), planes_of([src_tensor])
    check(load_library().ohevc_dev_upsample_plane(d, s, C.c_int(bit_depth), C.c_int(chroma), C.c_void_p(cols_ptr), C.c_void_p(col_of_ptr),
                                                  C.c_void_p(rows_ptr), C.c_int(src_cols), C.c_int(src_rows), C.c_void_p(stream)))


EXPORTED_SYMBOLS += ["ohevc_frame_end_deferred", "ohevc_debug_set_park_frames", "ohevc_debug_parked_total", "ohevc_pic_export_band", "ohevc_pic_import_band", "ohevc_upsample_make_maps", "ohevc_dev_upsample_plane", "ohevc_pic_upsample", "ohevc_tables_upsample_frame", "ohevc_ctx_set_concurrent", "ohevc_tables_host_planes", "ohevc_tables_derive_filters"]


class Ctx:
    """Thin wrapper over ohevc_ctx_* (one decoding thread's device context)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.lib.ohevc_ctx_stream.restype = C.c_void_p
        self.h = C.c_void_p()
        check(self.lib.ohevc_ctx_create(C.byref(self.h), C.c_int(device)))

    def close(self):
        if self.h:
            self.lib.ohevc_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.ohevc_ctx_sync(self.h))

    def pic_alloc(self, width, height, chroma_format_idc, bit_depth):
        slot = self.lib.ohevc_pic_alloc(self.h, width, height, chroma_format_idc, bit_depth)
        if slot < 0:
            check(slot)
        return slot

    def pic_adopt(self, tensors, width, height, chroma_format_idc, bit_depth):
        slot = self.lib.ohevc_pic_adopt(self.h, planes_of(tensors), width, height, chroma_format_idc, bit_depth)
        if slot < 0:
            check(slot)
        return slot

    def pic_release(self, slot):
        check(self.lib.ohevc_pic_release(self.h, slot))

    def pic_upload(self, slot, planes):
        for i, p in enumerate(planes):
            p = np.ascontiguousarray(p)
            check(self.lib.ohevc_pic_upload(self.h, slot, i, p.ctypes.data_as(C.c_void_p), C.c_ssize_t(p.strides[0])))

    def pic_download(self, slot, shapes, dtype):
        out = []
        for i, shp in enumerate(shapes):
            a = np.zeros(shp, dtype)
            check(self.lib.ohevc_pic_download(self.h, slot, i, a.ctypes.data_as(C.c_void_p), C.c_ssize_t(a.strides[0])))
            out.append(a)
        return out

    def pic_planes(self, slot):
        arr = (Plane * 3)()
        check(self.lib.ohevc_pic_planes(self.h, slot, arr))
        return arr

    def pic_export(self, slot, plane, device_ptr, nbytes):
        check(self.lib.ohevc_pic_export(self.h, slot, plane, C.c_void_p(device_ptr), C.c_size_t(nbytes)))

    def pic_import(self, slot, plane, device_ptr, nbytes):
        check(self.lib.ohevc_pic_import(self.h, slot, plane, C.c_void_p(device_ptr), C.c_size_t(nbytes)))

    def pic_upsample(self, dst_slot, src_slot, params):
        check(self.lib.ohevc_pic_upsample(self.h, dst_slot, src_slot, C.byref(params)))

    def frame_begin(self, slot):
        check(self.lib.ohevc_frame_begin(self.h, slot))

    def rec_tu(self, plane, x, y, log2, kind, coeffs, intra):
        c = np.ascontiguousarray(coeffs, dtype=np.int16)
        check(self.lib.ohevc_rec_tu(self.h, plane, x, y, log2, kind, c.ctypes.data_as(C.c_void_p), int(intra)))

    def _rec(self, fn, job):
        check(fn(self.h, job.ctypes.data_as(C.c_void_p)))

    def rec_mc(self, job):
        self._rec(self.lib.ohevc_rec_mc, job)

    def rec_intra(self, job):
        self._rec(self.lib.ohevc_rec_intra, job)

    def rec_deblock(self, job):
        self._rec(self.lib.ohevc_rec_deblock, job)

    def rec_sao(self, job):
        self._rec(self.lib.ohevc_rec_sao, job)

    def rec_bulk(self, mc=None, intra=None, tu_desc=None, tu_coeffs=None, dbk=None, sao=None):
        """Bulk recording from numpy arrays (job dtypes above; tu_desc int32 [n,6], tu_coeffs int16 concatenated)."""
        def ptr(a):
            return a.ctypes.data_as(C.c_void_p)
        if mc is not None and len(mc):
            check(self.lib.ohevc_rec_mc_bulk(self.h, ptr(mc), len(mc)))
        if intra is not None and len(intra):
            check(self.lib.ohevc_rec_intra_bulk(self.h, ptr(intra), len(intra)))
        if tu_desc is not None and len(tu_desc):
            check(self.lib.ohevc_rec_tu_bulk(self.h, len(tu_desc), ptr(tu_desc), ptr(tu_coeffs)))
        if dbk is not None and len(dbk):
            check(self.lib.ohevc_rec_deblock_bulk(self.h, ptr(dbk), len(dbk)))
        if sao is not None and len(sao):
            check(self.lib.ohevc_rec_sao_bulk(self.h, ptr(sao), len(sao)))

    def frame_reconstruct(self):
        check(self.lib.ohevc_frame_reconstruct(self.h))

    def frame_end(self):
        check(self.lib.ohevc_frame_end(self.h))

    def stats(self):
        st = FrameStats()
        check(self.lib.ohevc_frame_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in FrameStats._fields_}
