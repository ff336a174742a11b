"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports every symbol
include/*.h declares; job records have the documented layout.  No compute calls here."""
import ctypes as C
import glob
import os
import re

import numpy as np

from openhevc_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(ohevc_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load_library()
    decl = declared_functions()
    assert decl, "no declarations found"
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by libohevc_hip.so"
    assert set(L.EXPORTED_SYMBOLS) <= decl


def test_table_mirror_layout_matches_reference_headers():
    """sizeof/offsets of the ABI mirrors (include/ohevc_tables.h) against the reference's real structs, compiled here
    when /root/reference is available."""
    import shutil, subprocess, tempfile
    if not os.path.exists("/root/reference/libavcodec/hevcdsp.h") or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "c", "config.h")):
        import pytest
        pytest.skip("reference headers not available")
    src = r"""
#include <stdio.h>
#include <stddef.h>
#include "libavcodec/get_bits.h"
#include "libavcodec/hevc.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/videodsp.h"
#include "ohevc_tables.h"
#define CHK(a, b) if ((a) != (b)) { printf("MISMATCH %s %zu %zu\n", #a, (size_t)(a), (size_t)(b)); bad = 1; }
int main(void) { int bad = 0;
  CHK(sizeof(HEVCDSPContext), sizeof(ohevc_HEVCDSPContext))
  CHK(offsetof(HEVCDSPContext, transform_add), offsetof(ohevc_HEVCDSPContext, transform_add))
  CHK(offsetof(HEVCDSPContext, idct), offsetof(ohevc_HEVCDSPContext, idct))
  CHK(offsetof(HEVCDSPContext, sao_edge_filter), offsetof(ohevc_HEVCDSPContext, sao_edge_filter))
  CHK(offsetof(HEVCDSPContext, put_hevc_qpel_bi_w), offsetof(ohevc_HEVCDSPContext, put_hevc_qpel_bi_w))
  CHK(offsetof(HEVCDSPContext, put_hevc_epel_bi_w), offsetof(ohevc_HEVCDSPContext, put_hevc_epel_bi_w))
  CHK(offsetof(HEVCDSPContext, hevc_h_loop_filter_luma), offsetof(ohevc_HEVCDSPContext, hevc_h_loop_filter_luma))
  CHK(offsetof(HEVCDSPContext, hevc_v_loop_filter_chroma_c), offsetof(ohevc_HEVCDSPContext, hevc_v_loop_filter_chroma_c))
  CHK(offsetof(HEVCDSPContext, upsample_base_layer_frame), offsetof(ohevc_HEVCDSPContext, upsample_base_layer_frame))
  CHK(offsetof(HEVCDSPContext, upsample_filter_block_luma_v), offsetof(ohevc_HEVCDSPContext, upsample_filter_block_luma_v))
  CHK(offsetof(HEVCDSPContext, upsample_filter_block_cr_v), offsetof(ohevc_HEVCDSPContext, upsample_filter_block_cr_v))
  CHK(offsetof(VideoDSPContext, emulated_edge_up_v), offsetof(ohevc_VideoDSPContext, emulated_edge_up_v))
  CHK(sizeof(HEVCWindow), sizeof(ohevc_HEVCWindow))
  CHK(sizeof(UpsamplInf), sizeof(ohevc_UpsamplInf))
  CHK(offsetof(UpsamplInf, idx), offsetof(ohevc_UpsamplInf, idx))
  CHK(sizeof(VideoDSPContext), sizeof(ohevc_VideoDSPContext))
  CHK(offsetof(VideoDSPContext, prefetch), offsetof(ohevc_VideoDSPContext, prefetch))
  CHK(sizeof(SAOParams), sizeof(ohevc_SAOParams))
  CHK(offsetof(SAOParams, offset_val), offsetof(ohevc_SAOParams, offset_val))
  CHK(offsetof(SAOParams, eo_class), offsetof(ohevc_SAOParams, eo_class))
  return bad; }
"""
    d = tempfile.mkdtemp()
    try:
        open(os.path.join(d, "chk.c"), "w").write(src)
        subprocess.run(["gcc", "-std=gnu99", "-w", "-I" + os.path.join(ROOT, "oracle", "_ref", "c"), "-I/root/reference",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(d, "chk.c"), "-o", os.path.join(d, "chk")], check=True)
        r = subprocess.run([os.path.join(d, "chk")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
    finally:
        shutil.rmtree(d)


def test_job_record_layouts():
    assert L.TU_JOB.itemsize == 16
    assert [L.TU_JOB.fields[k][1] for k in ("x", "y", "plane", "dc", "coeff_off")] == [0, 2, 4, 6, 8]
    assert C.sizeof(L.Plane) == 24


def test_argument_validation_needs_no_gpu():
    lib = L.load_library()
    planes = (L.Plane * 3)()
    # bad bit depth / size / kind are rejected before any HIP call
    assert lib.ohevc_dev_tu_batch(planes, 7, 5, 0, None, 1, None, None) == L.ERR_ARG
    assert lib.ohevc_dev_tu_batch(planes, 8, 6, 0, None, 1, None, None) == L.ERR_ARG
    assert lib.ohevc_dev_tu_batch(planes, 8, 5, 99, None, 1, None, None) == L.ERR_ARG
    assert lib.ohevc_dev_tu_batch(planes, 8, 3, L.TU_DST4, None, 1, None, None) == L.ERR_ARG
    assert b"bad argument" in lib.ohevc_last_error()
    # an empty batch is a no-op
    assert lib.ohevc_dev_tu_batch(planes, 8, 5, 0, None, 0, None, None) == L.OK
    assert lib.ohevc_version().startswith(b"ohevc_hip")


def test_product_does_not_touch_the_oracle():
    """The product path must never route through oracle/ (or any CPU fallback)."""
    for path in glob.glob(os.path.join(ROOT, "openhevc_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
            text = open(path, errors="ignore").read()
            assert "pyoracle" not in text and "liboracle" not in text and "hevcref" not in text, path


def test_gpu_backed_decoder_links_no_oracle():
    """oracle/_ref/libopenhevc_hip.so = the reference's objects + integration/hip_hooks.c + libohevc_hip.so and nothing else that
    could produce a pixel: no symbol of the oracle (ohor_*), of the software executor (ohsw_*) or of the compiled reference kernels'
    shim (ohref_*) is defined or wanted; the only library it needs besides libc / libm / libpthread is libohevc_hip.so."""
    import subprocess
    import pytest
    so = os.path.join(ROOT, "oracle", "_ref", "libopenhevc_hip.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libopenhevc_hip.so not built (needs /root/reference once)")
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    bad = [ln for ln in syms.splitlines() if re.search(r"\b(ohor_|ohsw_|ohref_|ohsse_)", ln)]
    assert not bad, bad[:5]
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[(.*?)\]", subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout)
    assert "libohevc_hip.so" in needed
    assert all(n.startswith(("libohevc_hip", "libc.", "libm.", "libpthread", "ld-linux")) for n in needed), needed
    hooks = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "integration", "hip_hooks.c")).read(), flags=re.S)
    assert "oracle_api.h" not in hooks and "ohsw_" not in hooks and "ohor_" not in hooks


def test_picture_store_limits_fail_loudly():
    """The fixed sizes of the picture store (include/ohevc_ctx.h: OHEVC_MAX_PICTURES pictures, sides of at most 65535 samples) are met with an error
    and a message, never with a write past a table: a record-only context (no device) takes exactly OHEVC_MAX_PICTURES pictures."""
    lib = L.load_library()
    lib.ohevc_debug_set_record_only.argtypes = [C.c_int]
    prev = lib.ohevc_debug_set_record_only(1)
    try:
        ctx = L.Ctx(0)
        MAXP = 127
        slots = [ctx.pic_alloc(64, 64, 1, 8) for _ in range(MAXP)]
        assert sorted(slots) == list(range(MAXP))
        assert lib.ohevc_pic_alloc(ctx.h, 64, 64, 1, 8) == L.ERR_ARG and b"too many pictures" in lib.ohevc_last_error()
        ctx.pic_release(slots[5])                            # a freed slot is handed out again
        assert ctx.pic_alloc(64, 64, 1, 8) == slots[5]
        assert lib.ohevc_pic_alloc(ctx.h, 64, 64, 1, 8) == L.ERR_ARG
        lib.ohevc_frame_ref_reach.argtypes = [C.c_void_p, C.c_int]
        assert lib.ohevc_frame_ref_reach(ctx.h, MAXP + 1) == -1 and lib.ohevc_frame_ref_reach(ctx.h, -1) == -1
        ctx.pic_release(slots[6])
        assert lib.ohevc_pic_alloc(ctx.h, 65536, 64, 1, 8) == L.ERR_ARG and b"picture size" in lib.ohevc_last_error()
        assert lib.ohevc_pic_alloc(ctx.h, 64, 70000, 1, 8) == L.ERR_ARG
        ctx.close()
    finally:
        lib.ohevc_debug_set_record_only(prev)
