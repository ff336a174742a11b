import torch, time
n = 100 << 20
d = torch.empty(n, dtype=torch.uint8, device='cuda')
h = torch.empty(n, dtype=torch.uint8).pin_memory()
hp = torch.empty(n, dtype=torch.uint8)
for name, dst, src in (("D2H pinned", h, d), ("H2D pinned", d, h), ("D2H pageable", hp, d)):
    for _ in range(2): dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    print(name, round(5 * n / (time.perf_counter() - t) / 1e9, 1), "GB/s")
# 2-D: 4320 rows of 15360 bytes out of a pitch of 15360 / 15616
for pitch in (15360, 15616):
    dd = torch.empty(4320 * pitch, dtype=torch.uint8, device='cuda').view(4320, pitch)[:, :15360]
    hh = torch.empty(4320 * 15360, dtype=torch.uint8).pin_memory().view(4320, 15360)
    for _ in range(2): hh.copy_(dd, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): hh.copy_(dd, non_blocking=True)
    torch.cuda.synchronize()
    print("D2H pinned 2-D device pitch", pitch, round(5 * 4320 * 15360 / (time.perf_counter() - t) / 1e9, 1), "GB/s")
# the decoder's frame buffers are ordinary allocations page-locked afterwards (hipHostRegister): same rate as memory born page-locked?
import ctypes
rt = torch.cuda.cudart()
buf = torch.empty(n + 4096, dtype=torch.uint8)
ptr = (buf.data_ptr() + 4095) & ~4095
assert int(rt.cudaHostRegister(ptr, n, 0)) == 0
reg = torch.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=torch.uint8)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
for name, dstp in (("D2H into hipHostRegister-ed memory", ptr), ("D2H into hipHostMalloc-ed memory", h.data_ptr())):
    for _ in range(2): hip.hipMemcpyAsync(dstp, d.data_ptr(), n, 2, st); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): hip.hipMemcpyAsync(dstp, d.data_ptr(), n, 2, st)
    torch.cuda.synchronize()
    print(name, round(5 * n / (time.perf_counter() - t) / 1e9, 1), "GB/s")
for sz in (3 << 20, 1 << 20, 256 << 10):
    t = time.perf_counter()
    for _ in range(20): hip.hipMemcpyAsync(ptr, d.data_ptr(), sz, 2, st); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print("one D2H of", sz >> 10, "KiB into registered memory + sync:", round(dt * 1e6, 1), "us =", round(sz / dt / 1e9, 1), "GB/s")
