#!/bin/bash
# One gpurun call: smoke, the whole GPU suite, the bench lines that go to profiles/, the 128-frame roofline profiles.
# Everything lands in gpurun_out/$TAG/.   usage: gpurun --timeout 3000 -- 'bash tools/gpu_round_final_r06.sh r06_final'
TAG=${1:-r06_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
show() { python - <<P
import json
try:
    d = json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1]); c = d["config"]
    print("$1", d["value"], d["unit"], "ms/step", d["ms_per_step"], "window", c.get("timed_window_ms"), "probe", c.get("probe_window_ms"), "rehearsals", c.get("rehearsals"),
          "stall_max", c.get("host_stall_max_ms"), "throttled", c.get("cgroup_nr_throttled"), "verify", (d.get("verify") or {}).get("all_equal_eager"),
          "rows", (d.get("mlp_rows_per_step") or {}).get("evaluated_frac"), "roofline", (d.get("roofline") or {}).get("frac"),
          "mlp", (d.get("roofline_grouped_mlp") or {}).get("frac"), "other", c.get("other_executor_value"), "gather", (c.get("gather") or {}).get("ranks_seen"))
except Exception as e:
    print("$1 failed", e)
P
}
echo "== bench 20 steps (the driver's command), FIRST command of the lease"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps_cold.json 2> $OUT/bench_20steps_cold.err; show 20steps_cold
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --maxfail=50 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
echo "== bench 20 steps again (warm lease)"; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err; show 20steps
echo "== bench default"; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; show default
Q="--no-cpu-baseline --no-other-executor --extras-budget 0"
for i in 2 3; do timeout 300 python bench.py $Q --profile-iters 0 --verify 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default again', d['value'], d['ms_per_step'])"; done
timeout 600 python bench.py --host-input $Q --profile-iters 0 > $OUT/bench_host_input.json 2> $OUT/bench_host_input.err; show host_input
for v in dup10 dense rings64; do timeout 600 python bench.py --data $v $Q > $OUT/bench_$v.json 2> $OUT/bench_$v.err; show $v; done
timeout 600 python bench.py --gpus 2 --allow-shared-device --steps 64 --warmup 16 $Q > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; show 2ranks_shared
timeout 600 python bench.py --workload configs2 --no-cpu-baseline --extras-budget 0 > $OUT/bench_configs2.json 2> $OUT/bench_configs2.err; show configs2
timeout 600 python bench.py --workload configs4 $Q > $OUT/bench_configs4.json 2> $OUT/bench_configs4.err; show configs4
for b in 8 32 128; do timeout 600 python bench.py --workload group --batch $b > $OUT/bench_group_b$b.json 2> $OUT/bench_group_b$b.err; show group_b$b; done
timeout 600 python bench.py --workload group --batch 128 --share-grid > $OUT/bench_group_b128_shared_grid.json 2> $OUT/bench_group_b128_shared_grid.err; show group_b128_shared_grid
timeout 600 python bench.py --workload detector $Q > $OUT/bench_detector.json 2> $OUT/bench_detector.err; show detector
echo "== 128-frame roofline profiles"
bash tools/gpu_prof128.sh $TAG/prof128_default default 2>&1 | tail -34
bash tools/gpu_prof128.sh $TAG/prof128_rings64 rings64 2>&1 | tail -4
echo "== done"
