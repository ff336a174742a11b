#!/usr/bin/env python3
"""What this GPU sustains on plain streaming patterns, for comparison with the kernels' algorithmic GB/s: device-to-device copy
(1 read : 1 write), a 3 : 1 read/write mix like the 8-bit residual kernel's (two inputs added into a third buffer of half... see
below), and a read-only reduction.  One JSON line.     python tools/bench_hbm_ceiling.py"""
import json

import torch


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    n = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.empty_like(x)
    out = {}
    t = timed(lambda: y.copy_(x))
    out["copy_1r_1w_GBps"] = round(2 * n / t / 1e9, 1)
    xi = x.view(torch.int32)
    t = timed(lambda: xi.sum())
    out["read_only_sum_i32_GBps"] = round(n / t / 1e9, 1)
    # 3 reads : 1 write, in place like the residual kernel: int16 coefficients (2 B) + 1 B prediction read, 1 B written over it
    c = torch.randint(-1024, 1024, (n // 4,), dtype=torch.int16, device="cuda", generator=g)
    p = torch.randint(0, 256, (n // 4,), dtype=torch.uint8, device="cuda", generator=g)
    t = timed(lambda: p.add_(c.to(torch.uint8)))      # torch materialises the cast: extra traffic, lower bound only
    out["torch_inplace_add_lower_bound_GBps"] = round((n // 4) * 4 / t / 1e9, 1)
    y.zero_()
    t = timed(lambda: y.zero_())
    out["write_only_memset_GBps"] = round(n / t / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
