#!/bin/bash
# round 5, GPU pass 5: point-carrying cell lists in the grid ball query, XCD-local column blocks in dense128
OUT=gpurun_out/r05_pass5; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x -k "ball or query or dense or backbone or pipeline or ref_pin or fuzz or aggregation or vote" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for d in default rings64; do echo "== stages at 128 frames, data=$d"; timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep -v amdgpu.ids | tee $OUT/stages_128_$d.txt | grep -i "ball\|dense\|total"; done
echo "== stages at 32 frames, data=dense"; timeout 300 python tools/stages_at.py 32 data=dense 2>&1 | grep -v amdgpu.ids | tee $OUT/stages_32_dense.txt | grep -i "ball\|total"
echo "== sustained write bandwidth (torch fill of 7.2 GB, 20 x)"
python - <<'P'
import torch
x = torch.empty(7_200_000_000 // 4, dtype=torch.float32, device="cuda")
for _ in range(3): x.fill_(1.0)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): x.fill_(2.0)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("fill 7.2 GB: %.3f ms = %.2f TB/s" % (ms, 7.2e9 / ms / 1e9))
y = torch.empty_like(x[: x.numel() // 2]); z = x[: x.numel() // 2]
for _ in range(3): y.copy_(z)
s.record()
for _ in range(20): y.copy_(z)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("copy 3.6 GB -> 3.6 GB: %.3f ms = %.2f TB/s (read + write)" % (ms, 7.2e9 / ms / 1e9))
P
bash tools/gpu_prof128.sh r05_prof128_default default 2>&1 | grep "dense\|ball\|pass total\|grouped"
echo "== done"
