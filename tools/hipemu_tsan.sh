#!/bin/bash
# Data-race pass over the HOST layer without a device: ThreadSanitizer build of the library's host side (ctx.hip: picture store, executor, issuer,
# page locks; tables.hip: recorder, address registry; ...) and of the reference-side hooks (integration/hip_hooks.c), linked with the reference's
# decoder against the kernel emulator (tests/hipemu, SAN=thread; the kernels and the emulator itself are not instrumented - emulated lanes are
# fibers).  Runs the thread modes that share state: frame threads, slice threads, both, two decoders of one process at once, the two-layer pair
# with slice threads, asynchronous frame ends.
#   tools/hipemu_tsan.sh > profiles/<name>.txt
set -e
cd "$(dirname "$0")/.."
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.tsan-x86_64.so
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 second_deadlock_stack=1 history_size=3 exitcode=0 ${TSAN_EXTRA}"
export OHEVC_REF_WAIT_SECONDS=900
make -s -j8 -C tests/hipemu SAN=thread
make -s -C oracle hipemu_tsan
(cd tests && LD_PRELOAD=$RT python - <<'PY' 2>&1 | grep -v "^\[hevc\|makecontext\|IRAP\|^POC\|^\[MD5\|^[0-9a-f]\{32\}$\|^\]$"
import sys, time, threading
sys.path.insert(0, "..")
from oracle import pystream as ps
import test_stream_cpu as S
import shvc_exec as X
K = "hipemu_tsan"
t0 = time.time()
bad = 0
for name, threads, tt in (("ra_8b_ctb64", 4, 1), ("ldb_8b", 8, 1), ("wpp", 4, 2), ("tiles", 4, 2), ("slices_dep_wpp", 4, 3), ("intra_8b", 4, 1), ("ra_10b_odd", 3, 1)):
    aus, md5 = S.load_golden(name)
    ok = S.frames_md5(ps.decode_stream(K, aus, threads, tt)) == md5
    bad += not ok
    print("stream", name, "threads", threads, "type", tt, "equal to the reference:", ok, flush=True)
# two decoders of one process, each with frame threads, at the same time
import instance_cases
instance_cases.two_streams_concurrently(K, ("ra_10b_odd", "ldb_10b"), threads=3)
print("two decoders at once: equal to the reference", flush=True)
X.check_both_layers(K, "x2_wpp", threads=4, thread_type=2)
X.check_both_layers(K, "snr_wpp", threads=4, thread_type=2)
X.check_both_layers(K, "x2_ra")
print("two-layer pairs (slice threads; one thread per layer): equal to the reference", flush=True)
print("mismatches", bad, "; seconds", round(time.time() - t0, 1))
PY
)
# the switches that move work to other threads: asynchronous frame ends (the library's issuer threads), deferred copy-back, both
for env in "OHHIP_ASYNC_ISSUE=1" "OHHIP_DEFER_DOWNLOAD=0" "OHHIP_ASYNC_ISSUE=1 OHHIP_DEFER_DOWNLOAD=0"; do
(cd tests && env $env LD_PRELOAD=$RT python - "$env" <<'PY' 2>&1 | grep -v "^\[hevc\|makecontext\|IRAP"
import sys
sys.path.insert(0, "..")
from oracle import pystream as ps
import test_stream_cpu as S
for name, th in (("ra_8b_ctb64", 4), ("ldb_8b", 8), ("intra_8b", 4)):
    aus, md5 = S.load_golden(name)
    print(sys.argv[1], "stream", name, "frame threads", th, "equal to the reference:", S.frames_md5(ps.decode_stream("hipemu_tsan", aus, th, 1)) == md5, flush=True)
PY
)
done
echo "== every 'WARNING: ThreadSanitizer' block above is a finding; none = no data race seen in the instrumented host layer"
