"""Out-of-bounds accesses of the device code, caught without a device and without a sanitizer: kernel parity tests once more over the
emulated device code (tests/test_hipemu_cpu.py), with every "device" buffer ending flush against an inaccessible page (HIPEMU_GUARD=1,
tests/gpu_util.py).  A kernel that writes past the end of a plane, a job array or the coefficient arena - or reads there a value it
uses - kills the run.

What it does NOT catch is the finding that prompted it: round 2's restructured deblocking loaded the samples of a line before it knew
the line's tc, and for the second segment of a chroma edge below the plane (chroma heights are multiples of 4, edges are 8 lines long)
read up to 4 rows past it.  The values were never used, so every parity test passed - and the host compiler sinks such dead loads
behind the test that makes them dead, so the emulated code does not perform them.  The AddressSanitizer build (tools/hipemu_asan.sh,
minutes) instruments the loads where the source has them and is what found it; run it after every change of a kernel's load structure."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("modules", ["test_filters_gpu test_dbk_maps_gpu", "test_intra_gpu test_mc_gpu test_shvc_gpu"])
def test_kernels_stay_inside_their_buffers(modules):
    env = dict(os.environ, HIPEMU_GUARD="1", HIPEMU_MODULES=modules, OHEVC_PICTURE_BATCH="0")      # (every device picture its own allocation, ending at its own guard page)
    env.pop("HIPEMU_ASAN", None)
    # (four worker processes: the repeat is pure CPU work, the suite's own run is serial)
    workers = ["-n", str(min(4, os.cpu_count() or 1))] if (os.cpu_count() or 1) > 1 else []
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_hipemu_cpu.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "not stream and not decoders"] + workers,      # (kernel, ctx and table tests: the whole-decoder ones run once, in the suite itself)
                       capture_output=True, text=True, env=env, timeout=1500, cwd=HERE)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, f"the emulated kernels of {modules} left their buffers (or a test failed):\n{tail}"
    assert " passed" in r.stdout, tail
