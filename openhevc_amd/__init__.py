"""openhevc_amd -- MI355X (gfx950) backend for openHEVC's pixel-reconstruction hot path.

The product is the C-ABI shared library ``libohevc_hip.so`` (include/ohevc_hip.h) built from
``openhevc_amd/csrc``; this package is only the thin Python binding used by tests and bench.py.
Importing :mod:`openhevc_amd.lib` fails loudly if the library has not been built -- there is no
CPU fallback anywhere in the package.
"""
from .lib import load_library, OhevcError  # noqa: F401
