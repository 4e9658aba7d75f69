"""Layer-1 ball query alone (3dssd.yaml: 16384 points -> 4096 D-FPS centres, radii 0.2 / 0.4 / 0.8, nsample 32 / 32 / 64, dilated),
timed with events over `reps` launches of `frames` frames:   python tools/bq_bench.py [frames] [data] [reps]
Library variants via SA3D_LIB (tools/build_variant.sh).  Prints the median; `check` as 4th argument compares idx / cnt with
the scan kernel (sa_query_ball_point_multi)."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module("3dssd_amd." + m)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    data = sys.argv[2] if len(sys.argv) > 2 else "default"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    check = len(sys.argv) > 4 and sys.argv[4] == "check"
    npts = int(sys.argv[5]) if len(sys.argv) > 5 else 16384
    dev = torch.device("cuda:0")
    syn, N = pkg("synthetic"), pkg("utils._native")
    S = pkg("utils.tf_ops.sampling.tf_sampling")
    lib = N.lib()
    x = torch.from_numpy(np.stack([syn.frame_of(data, f, npts) for f in range(frames)])).to(dev)
    xyz = x[:, :, :3].contiguous()
    n, m = npts, 4096
    new_xyz = S.gather_point(xyz, S.farthest_point_sample(m, xyz)).contiguous()
    radii, nss = [0.2, 0.4, 0.8], [32, 32, 64]
    nb = 3
    rmax = (ctypes.c_float * nb)(*radii)
    rmin = (ctypes.c_float * nb)(0.0, 0.2, 0.4)
    nsa = (ctypes.c_int * nb)(*nss)

    def outputs():
        idx = [torch.empty((frames, m, ns), dtype=torch.int32, device=dev) for ns in nss]
        cnt = [torch.empty((frames, m), dtype=torch.int32, device=dev) for _ in nss]
        return idx, cnt, (ctypes.c_void_p * nb)(*[t.data_ptr() for t in idx]), (ctypes.c_void_p * nb)(*[t.data_ptr() for t in cnt])

    idx, cnt, idxp, cntp = outputs()
    ws = torch.empty((lib.sa_query_ball_point_grid_ws_bytes(frames, n, m) + 3) // 4, dtype=torch.int32, device=dev)
    stream = N.current_stream()

    def run():
        N.check(lib.sa_query_ball_point_grid(frames, n, m, nb, rmin, rmax, nsa, 1, xyz.data_ptr(), new_xyz.data_ptr(), idxp, cntp,
                                             ws.data_ptr(), stream), "grid")
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    c2 = cnt[2].float()
    print("bq layer1 %d frames %s: median %.1f us (min %.1f)   hits in the widest band: mean %.1f, share of full balls %.3f" %
          (frames, data, ts[len(ts) // 2], ts[0], float(c2.mean()), float((c2 >= nss[2]).float().mean())))
    if check:
        idx2, cnt2, idxp2, cntp2 = outputs()
        N.check(lib.sa_query_ball_point_multi(frames, n, m, nb, rmin, rmax, nsa, 1, xyz.data_ptr(), new_xyz.data_ptr(), idxp2, cntp2,
                                              stream), "multi")
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip(idx, idx2)) and all(torch.equal(a, b) for a, b in zip(cnt, cnt2))
        print("   idx / cnt equal to the scan kernel: %s" % ok)
        if not ok:
            for i in range(nb):
                bad = (idx[i] != idx2[i]).any(-1) | (cnt[i] != cnt2[i])
                w = bad.nonzero()
                print("   band %d: %d queries differ; first %s" % (i, int(bad.sum()), w[:3].tolist()))
                if len(w):
                    f, q = int(w[0][0]), int(w[0][1])
                    print("      grid cnt %d scan cnt %d" % (int(cnt[i][f, q]), int(cnt2[i][f, q])))
                    print("      grid", idx[i][f, q].tolist()); print("      scan", idx2[i][f, q].tolist())


if __name__ == "__main__":
    main()
