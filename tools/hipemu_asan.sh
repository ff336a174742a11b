#!/bin/bash
# Memory-safety pass over the device code WITHOUT a device: the kernel emulator's AddressSanitizer build (tests/hipemu, SAN=1).
# Device allocations and the tests' planes get red zones, so an out-of-bounds access of any kernel is a report with the kernel's
# source line.  Runs (1) the kernel / ctx / table parity tests, (2) every golden stream incl. thread modes, (3) the stream fuzzer.
#   tools/hipemu_asan.sh [fuzz_seconds] > profiles/<name>.txt
set -e
cd "$(dirname "$0")/.."
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0
export OHEVC_REF_WAIT_SECONDS=900      # frame threads: the emulated, instrumented device work of a reference picture can take minutes
export OHEVC_PICTURE_BATCH=0          # every device picture its own allocation: red zones around each (ctx.hip: take_piece)
make -s -j8 -C tests/hipemu SAN=1
make -s -C oracle hipemu_asan
echo "== kernel / ctx / table tests on libohevc_hip_emu_asan.so"
(cd tests && HIPEMU_ASAN=1 LD_PRELOAD=$RT python -m pytest test_hipemu_cpu.py -q -k "not stream" -p no:cacheprovider 2>&1 | tail -3)
echo "== golden streams on libopenhevc_hipemu_asan.so"
(cd tests && LD_PRELOAD=$RT python - <<'PY' 2>&1 | grep -v "^\[hevc\|makecontext"
import sys, time
sys.path.insert(0, "..")
from oracle import pystream as ps
import test_stream_cpu as S
from stream_cases import CASES
bad, t0 = 0, time.time()
for name in sorted(CASES):
    aus, md5 = S.load_golden(name)
    bad += S.frames_md5(ps.decode_stream("hipemu_asan", aus, 1, 1)) != md5
for name in ["wpp", "tiles", "ra_8b_ctb64"]:
    aus, md5 = S.load_golden(name)
    for tt in (1, 2, 3):
        bad += S.frames_md5(ps.decode_stream("hipemu_asan", aus, 4, tt)) != md5
print("golden streams", len(CASES), "+ 9 threaded decodes; mismatches", bad, "; seconds", round(time.time() - t0, 1))
PY
)
echo "== stream fuzzer on libopenhevc_hipemu_asan.so"
FUZZ_BACKEND=hipemu_asan LD_PRELOAD=$RT python tools/fuzz_streams.py "${1:-120}" 4711 2>&1 | grep -v "^\[hevc\|makecontext" | tail -3
echo "== no AddressSanitizer report above = no out-of-bounds access seen"
