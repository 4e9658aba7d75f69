#!/bin/bash
# round 5: one poller per workgroup in fps_coop.hip -- parity (incl. the orphaned-partner test) and time per pick
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "coop or cooperative or multi_workgroup or 65536 or beyond_16384 or uncapturable or stress or ref_pin" -p no:cacheprovider 2>&1 | tail -2
python - <<'P'
import importlib, numpy as np, torch, time
S = importlib.import_module("3dssd_amd.utils.tf_ops.sampling.tf_sampling")
syn = importlib.import_module("3dssd_amd.synthetic")
dev = torch.device("cuda:0")
for (b, n, c, m) in ((16, 65536, 3, 4096), (32, 16384, 67, 4096), (8, 40000, 3, 2048)):
    if c == 3:
        p = torch.from_numpy(np.stack([syn.frame_of("default", f, n)[:, :3] for f in range(b)])).to(dev)
    else:
        p = torch.randn((b, n, c), device=dev)
    for _ in range(2): S.farthest_point_sample(m, p)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): S.farthest_point_sample(m, p)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    print("fps b=%d n=%d c=%d m=%d: %.3f ms per call" % (b, n, c, m, ms))
P
timeout 300 python bench.py --workload configs4 --no-cpu-baseline --no-other-executor --profile-iters 1 --verify 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs4', d['value'], d['roofline'].get('us_per_pick'), d['roofline'].get('avg_launch_ms'))"
timeout 300 python bench.py --workload configs2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs2', d['value'], d['roofline'].get('us_per_pick'))"
