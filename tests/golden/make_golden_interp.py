"""Golden vectors of three_nn / three_interpolate / three_interpolate_grad generated FROM THE REFERENCE'S OWN CPU FUNCTIONS
(oracle/_ref/libtf_interpolate_ref.so, `make -C oracle ref`; needs /root/reference, so it runs in the build
container only).  Writes tests/golden/interp_ref.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import interp_oracle as I

assert I.ref_lib() is not None, "run `make -C oracle ref` first"
rng = np.random.default_rng(20260925)
out = {}
cases = {"rand": (2, 700, 150), "few": (1, 20, 2), "grid": (1, 300, 64)}
for name, (b, n, m) in cases.items():
    xyz1 = rng.uniform(-3, 3, (b, n, 3)).astype(np.float32)
    xyz2 = rng.uniform(-3, 3, (b, m, 3)).astype(np.float32)
    if name == "grid":      # lattice points: many exactly equal distances (tie-break by index order)
        xyz1 = np.round(xyz1); xyz2 = np.round(xyz2)
    dist, idx = I.ref_three_nn(xyz1, xyz2)
    pts = rng.normal(0, 1, (b, m, 37)).astype(np.float32)
    w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    interp = I.ref_three_interpolate(pts, idx, w)
    gout = rng.normal(0, 1, (b, n, 37)).astype(np.float32)
    gpts = I.ref_three_interpolate_grad(pts.shape, idx, w, gout)
    for k, v in dict(xyz1=xyz1, xyz2=xyz2, dist=dist, idx=idx, pts=pts, w=w, interp=interp, gout=gout, gpts=gpts).items():
        out["%s_%s" % (name, k)] = v
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "interp_ref.npz"), **out)
print("wrote", len(out), "arrays")
