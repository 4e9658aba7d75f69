#!/bin/bash
# round 5, GPU pass 4: smoke + the whole GPU suite, the 128-frame roofline profiles (default, rings64), one driver-command line
OUT=gpurun_out/r05_pass4; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
echo "== 128-frame profiles"
bash tools/gpu_prof128.sh r05_prof128_default default 2>&1 | tail -70
bash tools/gpu_prof128.sh r05_prof128_rings64 rings64 2>&1 | tail -45
echo "== the driver's command"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r05_pass4/bench_20steps.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v for k, v in list(d["config"].items())[:40]})
print("roofline", d["roofline"]["frac"], "mlp", d["roofline_grouped_mlp"]["frac"], d["roofline_grouped_mlp"]["nominal_frac"], "cpu", d["cpu_baseline"]["value"])
P
echo "== done"
