"""ctypes binding of libohevc_hip.so (include/ohevc_hip.h).  No compute happens in Python."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libohevc_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_NODEV, ERR_STATE = 0, -1, -2, -3, -4
TU_IDCT, TU_DC, TU_DST4, TU_SKIP, TU_SKIP_RDPCM_H, TU_SKIP_RDPCM_V, TU_BYPASS, TU_BYPASS_RDPCM_H, TU_BYPASS_RDPCM_V = range(9)


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


_lib = None


def load_library():
    """Load libohevc_hip.so; raise if it is missing (never fall back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OhevcError(f"{LIB_PATH} is missing: run `make -C openhevc_amd/csrc` (or __graft_entry__.build()) first")
    lib = C.CDLL(LIB_PATH)
    lib.ohevc_last_error.restype = C.c_char_p
    lib.ohevc_version.restype = C.c_char_p
    lib.ohevc_tu_kernel_name.restype = C.c_char_p
    lib.ohevc_tu_kernel_name.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.ohevc_device_count.restype = C.c_int
    lib.ohevc_set_device.argtypes = [C.c_int]
    lib.ohevc_dev_tu_batch.restype = C.c_int
    lib.ohevc_dev_tu_batch.argtypes = [C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


# every symbol include/ohevc_hip.h declares (checked by the CPU-side ABI test)
EXPORTED_SYMBOLS = ["ohevc_dev_tu_batch", "ohevc_last_error", "ohevc_device_count", "ohevc_set_device",
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
