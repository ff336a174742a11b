"""A/B of the up-sampling kernel's vertical pass on the device: matrix cores (variant 0) against dot products (variant 2), constant planes."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from openhevc_amd import lib as L
from oracle import pyoracle as po
import gpu_util as G
lib = L.load_library()
bd = 10
dt = G.pixdt(bd)
bw, bh, ew, eh = 208, 120, 416, 240
win = (0, 0, 0, 0)
up = po.shvc_params(bw, bh, ew, eh, win, phase_align=1)
prm = L.upsample_params(ew, eh, bw, bh, win, up, 0)
for c in (100, 300, 511, 512, 513, 640, 700, 1023):
    src = np.full((bh, bw), c, dt)
    outs = {}
    for variant in (2, 0):
        lib.ohevc_debug_set_upsample_variant(variant)
        cols, col_of, rows, sc, sr = L.upsample_maps(prm, 0)
        keep = [G.to_dev(a) for a in (cols, col_of, rows)]
        d_src, d_dst = G.to_dev(src), G.to_dev(np.zeros((eh, ew), dt))
        L.dev_upsample_plane(d_dst, d_src, bd, 0, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), sc, sr, G.stream())
        G.sync()
        outs[variant] = G.to_host(d_dst, dt).astype(np.int64)
    h = ((64 * c + 32768) % 65536) - 32768
    print("c", c, "h", h, "hi", h >> 8, "lo", h & 255, "dot2", np.unique(outs[2])[:4].tolist(), "mfma", np.unique(outs[0])[:6].tolist(), "mismatches", int((outs[0] != outs[2]).sum()))
