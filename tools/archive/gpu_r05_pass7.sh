#!/bin/bash
# round 5, GPU pass 7: cheaper submit path (pipeline tests + the driver's command), configs[4] line, grid ball query at layer 3
OUT=gpurun_out/r05_pass7; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "pipeline or bench" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q > $OUT/b20_$i.json 2> $OUT/b20_$i.err; python - $OUT/b20_$i.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("20 steps", d["value"], "window", c["timed_window_ms"], "probe", c["probe_window_ms"], "issue_total", c["host_issue_total_ms"], "top", c["host_issue_top_sites"], "pkg0_reached", c["pkg0_reached_ms"], "verify", d["verify"]["all_equal_eager"])
P
done
timeout 300 python bench.py $Q | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['config']['host_issue_total_ms'])"
timeout 600 python bench.py --workload configs4 --no-cpu-baseline --no-other-executor > $OUT/bench_configs4.json 2> $OUT/bench_configs4.err; python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r05_pass7/bench_configs4.json").read().strip().splitlines()[-1])
    print("configs4", d["value"], d["ms_per_step"], d["config"]["hip_graphs"], d["verify"]["all_equal_eager"], d["roofline"]["frac"], d["roofline"].get("us_per_pick"))
except Exception as e:
    print("configs4 failed", e, open("gpurun_out/r05_pass7/bench_configs4.err").read()[-600:])
P
for g in 2048 1024 512; do echo "== GRID_BALL_QUERY_MIN_N=$g"; timeout 300 python tools/stages_at.py 128 GRID_BALL_QUERY_MIN_N=$g 2>&1 | grep -i "ball\|total"; done
echo "== done"
