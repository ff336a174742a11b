"""Helpers shared by the -m gpu parity tests: device buffers via torch, calls via the C ABI."""
import numpy as np
import torch

from openhevc_amd import lib as L


def pixdt(bd):
    return np.uint16 if bd > 8 else np.uint8


def to_dev(a):
    """numpy -> CUDA tensor (uint16 goes through an int16 view: torch has no full uint16 support)."""
    a = np.ascontiguousarray(a)
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
    d_jobs = to_dev(jobs) if len(jobs) else torch.zeros(16, dtype=torch.uint8, device="cuda")
    d_coeffs = to_dev(np.ascontiguousarray(coeffs, dtype=np.int16).reshape(-1)) if coeffs is not None and coeffs.size else torch.zeros(8, dtype=torch.int16, device="cuda")
    L.dev_tu_batch(L.planes_of(d_planes), bd, log2, kind, d_jobs.data_ptr(), len(jobs), d_coeffs.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
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
    return torch.cuda.current_stream().cuda_stream
