#!/bin/bash
# Memory-safety pass over the TWO-LAYER (SHVC) path without a device: the two-layer fixtures, slice threads, decoder pairs opened and closed, and the
# two-layer fuzzer, through the hooked decoder linked against the AddressSanitizer build of the kernel emulator (see tools/hipemu_asan.sh).
#   tools/hipemu_asan_shvc.sh > profiles/<name>.txt
set -e
cd "$(dirname "$0")/.."
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0
export OHEVC_REF_WAIT_SECONDS=900
export OHEVC_PICTURE_BATCH=0          # every device picture its own allocation: red zones around each (ctx.hip: take_piece)
make -s -j6 -C tests/hipemu SAN=1
make -s -C oracle hipemu_asan
echo "== two-layer streams on libopenhevc_hipemu_asan.so"
(cd tests && LD_PRELOAD=$RT python - <<'PY' 2>&1 | grep -v "^\[hevc\|makecontext\|IRAP"
import sys, time
sys.path.insert(0, "..")
from oracle import pystream as ps
from shvc_cases import SHVC_CASES
import shvc_exec as X
t0 = time.time()
for name in sorted(SHVC_CASES):
    X.check_both_layers("hipemu_asan", name)
X.check_both_layers("hipemu_asan", "x2_wpp", threads=4, thread_type=2)
X.open_close_layer_pairs("hipemu_asan", ps._load("hipemu_asan"), rounds=4)
print("two-layer streams", len(SHVC_CASES), "+ slice threads + 4 open/close pairs: all equal to the reference; seconds", round(time.time() - t0, 1))
PY
)
echo "== two-layer fuzzer on libopenhevc_hipemu_asan.so"
FUZZ_BACKEND=hipemu_asan LD_PRELOAD=$RT python tools/fuzz_shvc.py 240 99 2>&1 | grep -v "^\[hevc\|makecontext\|IRAP" | tail -3
echo "== no AddressSanitizer report above = no out-of-bounds access seen"
