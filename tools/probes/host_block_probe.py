"""What a frame buffer costs to make: ohevc_host_alloc (hipHostMalloc) against calloc + ohevc_host_pin (hipHostRegister), per block size.
    python tools/probes/host_block_probe.py [blocks]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openhevc_amd import lib as L
lib = L.load_library()
libc = C.CDLL("libc.so.6")
libc.calloc.restype = C.c_void_p; libc.calloc.argtypes = [C.c_size_t, C.c_size_t]; libc.free.argtypes = [C.c_void_p]
lib.ohevc_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
lib.ohevc_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.ohevc_host_free.argtypes = [C.c_void_p]
lib.ohevc_host_pin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.ohevc_host_unpin_all.argtypes = [C.c_void_p]
ctx = C.c_void_p(); assert lib.ohevc_ctx_create(C.byref(ctx), 0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for size, name in ((4_700_000, "1080p 8 bit frame"), (18_700_000, "4K 8 bit"), (101_800_000, "8K 10 bit")):
    k = n if size < 50_000_000 else max(4, n // 4)
    t = time.perf_counter(); ps = []
    for _ in range(k):
        p = C.c_void_p(); assert lib.ohevc_host_alloc(ctx, size, C.byref(p)) == 0; ps.append(p)
    t_alloc = (time.perf_counter() - t) / k
    t = time.perf_counter()
    for p in ps: lib.ohevc_host_free(p)
    t_free = (time.perf_counter() - t) / k
    t = time.perf_counter(); qs = []
    for _ in range(k):
        q = libc.calloc(1, size); C.memset(q, 0, size); qs.append(q)
    t_calloc = (time.perf_counter() - t) / k
    t = time.perf_counter()
    for q in qs: assert lib.ohevc_host_pin(ctx, q, size) == 0
    t_pin = (time.perf_counter() - t) / k
    t = time.perf_counter(); lib.ohevc_host_unpin_all(ctx); t_unpin = (time.perf_counter() - t) / k
    for q in qs: libc.free(q)
    print(f"{name:18s} {size / 1e6:6.1f} MB x {k}: host_alloc {1e3 * t_alloc:7.3f} ms  host_free {1e3 * t_free:7.3f} | calloc+touch {1e3 * t_calloc:7.3f}  host_pin {1e3 * t_pin:7.3f}  unpin {1e3 * t_unpin:7.3f}")
# the first allocations after everything was freed (what a fresh decoder instance sees), one by one
for rnd in range(3):
    ts = []; ps = []
    for _ in range(6):
        t = time.perf_counter(); p = C.c_void_p(); assert lib.ohevc_host_alloc(ctx, 4_700_000, C.byref(p)) == 0; ts.append(1e3 * (time.perf_counter() - t)); ps.append(p)
    t = time.perf_counter()
    for p in ps: lib.ohevc_host_free(p)
    print(f"round {rnd}: six 4.7 MB blocks one by one (ms): {' '.join(f'{x:.3f}' for x in ts)}; freeing all six {1e3 * (time.perf_counter() - t):.3f} ms")
