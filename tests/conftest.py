import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU tier (-m "not gpu") is ~800 tests, half of them whole decodes over the emulated device code: ~13 minutes on one core, ~4 on four.
    # When nobody asked for a number of workers, spread it over four pytest-xdist workers (the tests are independent processes' worth of work:
    # every session of this repository has run them with -n 8).  Never for the device tier (one GPU), never inside a worker;
    # OHEVC_TEST_WORKERS=0 switches it off, OHEVC_TEST_WORKERS=n picks another number.
    # (round 6: ~910 tests; six workers where the box has eight cores - 12 minutes on four, ~8 on six)
    want = os.environ.get("OHEVC_TEST_WORKERS", str(6 if (os.cpu_count() or 1) >= 8 else 4))
    if (hasattr(config, "workerinput") or not want.isdigit() or int(want) < 2 or (os.cpu_count() or 1) < 4
            or "not gpu" not in (getattr(config.option, "markexpr", "") or "") or getattr(config.option, "collectonly", False)):
        return
    if getattr(config.option, "numprocesses", None) or getattr(config.option, "dist", "no") != "no" or not config.pluginmanager.hasplugin("xdist"):
        return
    n = min(int(want), os.cpu_count() or 1)
    config.option.numprocesses = n           # what "-n 4" sets (xdist/plugin.py: pytest_cmdline_main); xdist's own pytest_configure runs after this one
    config.option.dist = "load"
    config.option.tx = ["popen"] * n


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle.load("oracle")


@pytest.fixture(scope="session")
def ref():
    from oracle import pyoracle
    lib = pyoracle.load("ref")
    if lib is None:
        pytest.skip("oracle/_ref/libhevcref.so not built (needs /root/reference)")
    return lib
