"""VERDICT r3 item 4, the measured attempt: F-FPS WITHOUT the distance matrix at the layer-2 shape.  The multi-workgroup
on-the-fly sampler that exists (csrc/fps_coop.hip, fps_coop_kernel<67, 1>: 1024 threads x 1 point x 67 channels in
registers, cross-workgroup arg-max through sc1 slots) run on [frames, 4096, 67] -> 512 picks: G = 4 workgroups per frame,
i.e. 4 x frames CUs.  Its arithmetic is the raw-point form (sum of squared differences), not calc_square_dist's
norm - 2 dot form, so the PICKS are not the layer's; the TIME is what the item asks for: per pick the same 67-channel
update per point + the same exchange.  Printed next to the matrix path (distance matrix kernel + fps_dual launch) on
the same frames."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
S, syn, lu = pkg("utils.tf_ops.sampling.tf_sampling"), pkg("synthetic"), pkg("utils.layers_util")
dev = torch.device("cuda:0")
n, c1, m = 4096, 64, 512


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for frames in (8, 32, 64):
    rng = np.random.default_rng(frames)
    xyz = torch.from_numpy(np.stack([syn.frame_of("default", 40 + i, 16384)[:n, :3] for i in range(frames)])).to(dev).contiguous()
    feat = torch.from_numpy(rng.normal(0, 0.5, (frames, n, c1)).astype(np.float32)).to(dev)
    pts = torch.cat([xyz, feat], 2).contiguous()
    t_fly = timed(lambda: S.farthest_point_sample(m, pts))
    # the matrix path as the backbone issues it: 'FS' layer, F-FPS 512 + D-FPS 512 of the same range in one dual launch
    t_mat = timed(lambda: lu.sample_layer(xyz, feat, [-1], ["FS"], [m], None, None, [0.4], side_mode=5))
    t_d = timed(lambda: S.farthest_point_sample(m, xyz))
    # csrc/ffps_fly.hip: the same idea with the matrix arithmetic (the layer's own picks), two points per thread and one
    # packed FMA per channel, the squared norm of the candidate travelling with the exchange; + the D-FPS half behind it
    a = lu.sample_layer(xyz, feat, [-1], ["FS"], [m], None, None, [0.4], side_mode=5, ffps_fly=False)
    bfly = lu.sample_layer(xyz, feat, [-1], ["FS"], [m], None, None, [0.4], side_mode=5, ffps_fly=True)
    torch.cuda.synchronize()
    same = bool(torch.equal(a[0], bfly[0]) and torch.equal(a[1], bfly[1]))
    t_new = timed(lambda: lu.sample_layer(xyz, feat, [-1], ["FS"], [m], None, None, [0.4], side_mode=5, ffps_fly=True))
    print("frames %3d: ffps_fly.hip F-FPS + D-FPS half %.3f ms -> F part %.3f ms = %.2f us per pick, %.1f CU-ms (picks equal to the matrix "
          "path: %s)" % (frames, t_new, t_new - t_d, (t_new - t_d) * 1e3 / (m - 1), (t_new - t_d) * 4 * frames, same), flush=True)
    print("frames %3d: on-the-fly multi-workgroup F-FPS %.3f ms (%d CUs, %.2f us per pick, %.1f CU-ms) | matrix + dual sampler %.3f ms | "
          "D-FPS half alone %.3f ms" % (frames, t_fly, 4 * frames, t_fly * 1e3 / (m - 1), t_fly * 4 * frames, t_mat, t_d), flush=True)
