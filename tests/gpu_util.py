"""Helpers shared by the parity tests: device buffers via torch, calls via the C ABI.

Two back-ends.  The default one is the real thing (-m gpu): CUDA tensors, libohevc_hip.so.  tests/test_hipemu_cpu.py switches to
the kernel emulator (use_emulator()): "device" memory is host numpy memory and the library is tests/hipemu/libohevc_hip_emu.so, the
same kernel sources compiled for the host - a check of the device code's arithmetic that needs no GPU (tests/hipemu/README.md)."""
import ctypes as C
import os

import numpy as np

from openhevc_amd import lib as L

_EMU = None


def _guarded_copy(a):
    """HIPEMU_GUARD=1: a copy of `a` whose last 16-byte piece ends where an inaccessible page begins.  A kernel of the emulated device code
    that reads or writes past the end of a plane / job array / coefficient arena then dies with SIGSEGV instead of reading its
    neighbour in silence (tests/test_guard_pages_cpu.py; the AddressSanitizer build, tools/hipemu_asan.sh, is the thorough form)."""
    import mmap
    a = np.ascontiguousarray(a)
    page = mmap.PAGESIZE
    span = max((a.nbytes + 15) // 16 * 16, 16)
    total = (span + page - 1) // page * page + page
    m = mmap.mmap(-1, total)
    addr = C.addressof(C.c_char.from_buffer(m))
    libc = C.CDLL(None, use_errno=True)
    if libc.mprotect(C.c_void_p(addr + total - page), C.c_size_t(page), 0) != 0:
        raise OSError(C.get_errno(), "mprotect")
    buf = np.frombuffer(m, dtype=np.uint8, count=a.nbytes, offset=total - page - span).view(a.dtype).reshape(a.shape)
    buf[...] = a
    return buf                                               # keeps `m` alive through its base


class HostTensor:
    """The few tensor methods the helpers use, over a numpy array (the emulator's device memory)."""

    def __init__(self, a):
        self.a = _guarded_copy(a) if os.environ.get("HIPEMU_GUARD") == "1" else np.ascontiguousarray(a).copy()
        self.shape = self.a.shape

    def data_ptr(self):
        return self.a.ctypes.data

    def dim(self):
        return self.a.ndim

    def stride(self, i):
        return self.a.strides[i] // self.a.itemsize

    def element_size(self):
        return self.a.itemsize

    def cpu(self):
        return self

    def numpy(self):
        return self.a

    def clone(self):
        return HostTensor(self.a)


def emulator_path():
    """tests/hipemu/libohevc_hip_emu.so, or the AddressSanitizer build with HIPEMU_ASAN=1 (tests/hipemu/README.md)."""
    name = "libohevc_hip_emu_asan.so" if os.environ.get("HIPEMU_ASAN") == "1" else "libohevc_hip_emu.so"
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu", name)


def use_emulator(on=True):
    """Point openhevc_amd.lib at the emulator library (tests only; the product never does this).  Returns the previous library."""
    global _EMU
    prev = L._lib
    if on:
        _EMU = _EMU or L.bind_prototypes(C.CDLL(emulator_path()))
        L._lib = _EMU
    return prev


def emulating():
    return _EMU is not None and L._lib is _EMU


def sync():
    if not emulating():
        import torch
        torch.cuda.synchronize()


def zeros_dev(n, dtype):
    return HostTensor(np.zeros(n, dtype)) if emulating() else to_dev(np.zeros(n, dtype))


def pixdt(bd):
    return np.uint16 if bd > 8 else np.uint8


def to_dev(a):
    """numpy -> CUDA tensor (uint16 goes through an int16 view: torch has no full uint16 support)."""
    a = np.ascontiguousarray(a)
    if emulating():
        return HostTensor(a.view(np.uint8) if a.dtype.fields is not None else a)
    import torch
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).cuda()
    if a.dtype.fields is not None:
        return torch.from_numpy(a.view(np.uint8)).cuda()
    return torch.from_numpy(a).cuda()


def to_host(t, dtype):
    a = t.cpu().numpy()
    return a.view(dtype) if a.dtype != dtype else a


def run_tu(bd, log2, kind, planes_np, jobs, coeffs):
    """planes_np: list of up to 3 numpy planes (modified copies are returned)."""
    d_planes = [to_dev(p) if p is not None else None for p in planes_np]
    while len(d_planes) < 3:
        d_planes.append(None)
    d_jobs = to_dev(jobs) if len(jobs) else zeros_dev(16, np.uint8)
    d_coeffs = to_dev(np.ascontiguousarray(coeffs, dtype=np.int16).reshape(-1)) if coeffs is not None and coeffs.size else zeros_dev(8, np.int16)
    L.dev_tu_batch(L.planes_of(d_planes), bd, log2, kind, d_jobs.data_ptr(), len(jobs), d_coeffs.data_ptr(),
                   stream())
    sync()
    return [to_host(t, p.dtype) if t is not None else None for t, p in zip(d_planes, planes_np + [None] * 3)]


def make_tu_jobs(xy, n, planes=None, dcs=None):
    jobs = np.zeros(len(xy), L.TU_JOB)
    if len(xy):
        xy = np.asarray(xy)
        jobs["x"], jobs["y"] = xy[:, 0], xy[:, 1]
        jobs["coeff_off"] = np.arange(len(xy), dtype=np.uint32) * n * n
        if planes is not None:
            jobs["plane"] = planes
        if dcs is not None:
            jobs["dc"] = dcs
    return jobs


def planes3(d_planes):
    d = list(d_planes) + [None] * (3 - len(d_planes))
    return L.planes_of(d)


def stream():
    if emulating():
        return 0
    import torch
    return torch.cuda.current_stream().cuda_stream
