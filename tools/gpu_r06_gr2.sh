#!/bin/bash
# 2-row granules: bit identity against the 8-row plans on three kinds of frames, then the stage timings
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - > gpurun_out/gr2_ident.txt 2>&1 <<'PY'
import importlib, numpy as np, torch
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn, lu, B = pkg("configs"), pkg("synthetic"), pkg("utils.layers_util"), pkg("backbone")
gpu = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
net = B.SABackbone(arch, syn.random_backbone_params(arch), gpu, cfgs.KITTI_MAX_TRANSLATE_RANGE)
G2 = {4096: {0: 2, 1: 2}, 1024: {0: 2, 1: 2, 2: 4}, 512: {0: 2}}
for variant in ("default", "rings64", "dense"):
    pts = torch.from_numpy(np.stack([syn.frame_of(variant, f, 16384) for f in range(3)])).to(gpu)
    def run(flag):
        default = lu.MLP_GRANULE4
        lu.MLP_GRANULE4, lu.PLAN_LOG = flag, []
        try:
            xl, fl, il = net(pts); torch.cuda.synchronize()
            hd = [p[4][:4].cpu().tolist() for p in lu.PLAN_LOG]
        finally:
            lu.MLP_GRANULE4, lu.PLAN_LOG = default, None
        return [t.clone() for t in fl], hd
    f8, h8 = run(False)
    f2, h2 = run(G2)
    print(variant, [h[3] for h in h2], "equal", all(torch.equal(a, b) for a, b in zip(f8, f2)),
          "rows8", sum(h[0]*h[3] for h in h8), "rows2", sum(h[0]*h[3] for h in h2), "distinct", sum(h[2] for h in h2))
    print("  ", h2)
PY
cat gpurun_out/gr2_ident.txt
for d in default rings64; do
  python tools/stages_at.py 128 data=$d 2>&1 | tail -12 > gpurun_out/gr2_${d}_base.txt
  python tools/stages_at.py 128 data=$d "MLP_GRANULE4={4096: {0: 2, 1: 2}, 1024: {0: 2, 1: 2, 2: 4}, 512: {0: 2}}" 2>&1 | tail -12 > gpurun_out/gr2_${d}_all.txt
  python tools/stages_at.py 128 data=$d "MLP_GRANULE4={4096: {0: 2, 1: 2}, 1024: True}" 2>&1 | tail -12 > gpurun_out/gr2_${d}_l1.txt
  python tools/stages_at.py 128 data=$d "MLP_GRANULE4={4096: (0, 1), 1024: {0: 2, 1: 2, 2: 4}}" 2>&1 | tail -12 > gpurun_out/gr2_${d}_l2.txt
  python tools/stages_at.py 128 data=$d "MLP_GRANULE4={4096: (0, 1), 1024: True, 512: {0: 2}}" 2>&1 | tail -12 > gpurun_out/gr2_${d}_l3.txt
done
tail -n 12 gpurun_out/gr2_*_*.txt
