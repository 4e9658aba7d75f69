"""Parity STATISTICS over populations: numbers that single-case tests cannot give.

  * free-running agreement rate: the chain is chaotic in its index outputs -- one F-FPS pick that flips on a ~1e-5
    feature difference (split bf16 against the fp32 oracle) re-orders everything downstream -- so the free-running test
    of test_backbone_gpu.py compares "until a near tie flips".  Here: over 64 default + 16 rings64 frames, the fraction of
    frames whose indices equal the oracle's through ALL rows, where the others first differ, and the final-feature error
    of the frames that agree;
  * fp16 headroom: layers 3 / 4 run their scales in ONE fp16 pass against a 1e-3 bar (utils/weights.py); 8 weight seeds x
    {default, rings64, dense} frames, max |d| / max |ref| per scale (the bar's metric) and an element-relative figure.
Both write a JSON report next to the other GPU artefacts (gpurun_out/, copied to profiles/ by hand)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _report(name, obj):
    d = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump(obj, f, indent=1)


def test_free_running_agreement_rate_over_a_frame_population(gpu, oracle):
    cfgs, syn = pkg("configs"), pkg("synthetic")
    arch = cfgs.KITTI_3DSSD_ARCH
    params = syn.random_backbone_params(arch)
    net = pkg("backbone").SABackbone(arch, params, gpu, cfgs.KITTI_MAX_TRANSLATE_RANGE)
    report = {}
    for variant, nframes, first in (("default", 64, 2000), ("rings64", 16, 2000)):
        pts = np.stack([syn.frame_of(variant, first + f, 16384) for f in range(nframes)])
        xl, fl, il = net(torch.from_numpy(pts).to(gpu))
        torch.cuda.synchronize()
        net.raise_if_overflow()
        rxl, rfl, ril = oracle.sa_backbone(pts, arch, params, cfgs.KITTI_MAX_TRANSLATE_RANGE)
        # layer 1 is pure geometry: bit-exact in every frame
        assert np.array_equal(il[1].cpu().numpy(), ril[1]) and np.array_equal(xl[1].cpu().numpy(), rxl[1])
        agree = np.ones(nframes, bool)
        first_diff = [None] * nframes
        for li in range(1, len(ril)):
            if ril[li] is None:
                continue
            same = (il[li].cpu().numpy() == ril[li]).all(axis=1)
            for f in np.nonzero(agree & ~same)[0]:
                col = int(np.nonzero(il[li][f].cpu().numpy() != ril[li][f])[0][0])
                first_diff[f] = (arch[li - 1][12], col)
            agree &= same
        feat, ref = fl[-1].cpu().numpy(), rfl[-1]
        err = np.array([float(np.abs(feat[f] - ref[f]).max() / np.abs(ref[f]).max()) for f in range(nframes)])
        # the vote layer's centres are MLP outputs (floats): with the oracle's indices they differ by rounding only ...
        vote = np.array([float(np.abs(xl[-1][f].cpu().numpy() - rxl[-1][f]).max()) for f in range(nframes)])
        # ... but layer 4's ball query runs on them, and a point a few ulps from a radius flips a neighbour set: the
        # frames with the oracle's indices split into those whose last layer saw the same balls (error ~1e-4) and those
        # where a membership flipped (a pooled maximum changes discontinuously)
        same_balls = agree & (err < TOL)
        flipped = agree & ~(err < TOL)
        where = {}
        for d in first_diff:
            if d is not None:
                where[d[0]] = where.get(d[0], 0) + 1
        q = lambda a: None if len(a) == 0 else {"max": float(np.max(a)), "median": float(np.median(a))}
        report[variant] = {
            "frames": nframes, "frames_with_oracle_indices_through_all_rows": int(agree.sum()),
            "agreement_rate": round(float(agree.mean()), 4), "first_difference_by_layer": where,
            "first_difference_pick_positions": sorted(d[1] for d in first_diff if d is not None),
            "of_those_final_features_within_1e-3": int(same_balls.sum()),
            "of_those_with_a_layer4_ball_membership_flip": int(flipped.sum()),
            "final_feature_err_within_bar": q(err[same_balls]), "final_feature_err_membership_flip": q(err[flipped]),
            "vote_centre_abs_err_where_indices_agree": q(vote[agree]),
            "metric": "max |gpu - oracle| / max |oracle| of the [256,512] output per frame (the 1e-3 bar's metric)"}
        print("free-running agreement, %s: %d of %d frames keep the oracle's indices through every row (first differences %s); of "
              "those %d end within 1e-3 (max %.2e), %d have a layer-4 ball membership flip on a vote-shifted centre; centres differ by <= %.1e m"
              % (variant, int(agree.sum()), nframes, where, int(same_balls.sum()), err[same_balls].max() if same_balls.any() else 0.0,
                 int(flipped.sum()), vote[agree].max() if agree.any() else 0.0))
        assert (vote[agree] < 3e-3).all(), "with the oracle's indices the vote centres (xyz + offsets clipped to +-3 m) must agree within the 1e-3 bar: %g" % vote[agree].max()
        # a first difference is only ever an F-FPS pick (layers 2 / 3): D-FPS and the ball query are pure geometry on bit-identical centres
        assert set(where) <= {"layer2", "layer3"}, where
    report["note"] = ("oracle.sa_backbone (fp32, CPU) against SABackbone (default per-scale precision) from the raw cloud, no teacher "
                      "forcing; a frame 'agrees' when every fps_idx list of every row is bit-equal.  Frames that do not agree "
                      "differ first at an F-FPS pick whose two candidates are a near tie in the 67- / 131-channel distance; a "
                      "frame that agrees can still see a layer-4 neighbour set change, because that layer queries balls around "
                      "vote-shifted centres (floats from an MLP); every layer of every frame is covered by the teacher-forced test")
    _report("r06_free_running_agreement.json", report)
    assert report["default"]["agreement_rate"] > 0.25


def test_fp16_headroom_sweep_over_weight_seeds_and_frame_kinds(gpu, oracle):
    cfgs, syn = pkg("configs"), pkg("synthetic")
    lu = pkg("utils.layers_util")
    arch = cfgs.KITTI_3DSSD_ARCH
    rows = {"layer3": arch[2], "layer4": arch[5]}
    assert rows["layer3"][12] == "layer3" and rows["layer4"][12] == "layer4"
    worst, table = {}, []
    for seed in range(8):
        params = syn.random_backbone_params(arch, seed=syn.WEIGHT_SEED + 100 * seed)
        net = pkg("backbone").SABackbone(arch, params, gpu, cfgs.KITTI_MAX_TRANSLATE_RANGE)
        for variant in ("default", "rings64", "dense"):
            pts = np.stack([syn.frame_of(variant, 3000 + 7 * seed + f, 16384) for f in range(2)])
            lu.CONCAT_LOG = []
            try:
                xl, fl, il = net(torch.from_numpy(pts).to(gpu))
                torch.cuda.synchronize()
            finally:
                log, lu.CONCAT_LOG = lu.CONCAT_LOG, None
            net.raise_if_overflow()
            cpu = lambda ts: [None if t is None else t.cpu().numpy() for t in ts]
            xl, fl, il = cpu(xl), cpu(fl), cpu(il)
            for scope, concat, offs, widths, precs in log:
                if scope not in rows:
                    continue
                assert set(precs) == {"fp16"}, (scope, precs)       # the scales under test are the one-pass ones
                row = rows[scope]
                li = [r[12] for r in arch].index(scope)
                vote = xl[row[14]] if row[14] != -1 else None
                former = il[row[9]] if row[9] != -1 else None
                trace = []
                # teacher-forced: the oracle recomputes the row from the GPU's own inputs to it
                rx, _rf, ri = oracle.pointnet_sa_module_msg(xl[row[0][0]], fl[row[1][0]], row[2], row[3], row[4], row[5], row[6], row[7],
                                                            row[8], former, scope, row[13], params, vote_ctr=vote,
                                                            aggregation_channel=row[15], trace=trace)
                assert np.array_equal(rx, xl[li + 1]) and (ri is None or np.array_equal(ri, il[li + 1]))
                got = concat.cpu().numpy()
                for t in trace:
                    i = t["scale"]
                    g, r = got[:, :, offs[i]:offs[i] + widths[i]], t["pooled"]
                    d = np.abs(g - r)
                    e_max = float(d.max() / np.abs(r).max())
                    big = np.abs(r) >= 0.01 * np.abs(r).max()              # elements that carry signal
                    e_rel = float((d[big] / np.abs(r[big])).max())
                    e_p999 = float(np.quantile(d[big] / np.abs(r[big]), 0.999))
                    key = "%s scale %d" % (scope, i)
                    table.append({"seed": seed, "data": variant, "scale": key, "err_max_over_max": e_max,
                                  "elem_rel_max_where_ref_ge_1pct_of_max": e_rel, "elem_rel_p99.9": e_p999})
                    w = worst.setdefault(key, {"err_max_over_max": 0.0, "elem_rel_max": 0.0, "elem_rel_p99.9": 0.0})
                    w["err_max_over_max"] = max(w["err_max_over_max"], e_max)
                    w["elem_rel_max"] = max(w["elem_rel_max"], e_rel)
                    w["elem_rel_p99.9"] = max(w["elem_rel_p99.9"], e_p999)
    for k in sorted(worst):
        print("fp16 headroom %-16s max|d|/max|ref| %.2e (bar 1e-3)   element-relative max %.2e, p99.9 %.2e" %
              (k, worst[k]["err_max_over_max"], worst[k]["elem_rel_max"], worst[k]["elem_rel_p99.9"]))
    _report("r06_fp16_headroom.json", {"worst_per_scale": worst, "cases": table, "bar": TOL,
                                        "note": "8 weight seeds x {default, rings64, dense} x 2 frames; pooled output of every fp16 scale "
                                                "(before the aggregation layer) against the fp32 oracle's on the GPU's own inputs to the layer; "
                                                "err_max_over_max is the bar's metric (gating), the element-relative figures are reported only"})
    assert len(worst) == 5 and all(w["err_max_over_max"] < TOL for w in worst.values()), worst
