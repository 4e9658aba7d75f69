#!/bin/bash
# round 6, VERDICT r5 item 6: the matrix-free F-FPS inside the executor (a stage and a stream of its own) against the matrix path
TAG=${1:-r06_fly}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --extras-budget 0 --profile-iters 0"
show() { python - <<P
import json
try:
    d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1]); c = d["config"]
    print("$1", d["value"], "frames/s  ms/step", d["ms_per_step"], "verify", (d.get("verify") or {}).get("all_equal_eager"), "sclk", c.get("sclk_before"), c.get("sclk_after"), "streams", c.get("streams_used"), "hwq", c.get("hw_queues"))
except Exception as e:
    print("$1 failed", e); print(open("$OUT/$1.err").read()[-1500:])
P
}
for rep in 1 2; do
  for fly in 0 1; do
    for hq in 4 8; do
      n=fly${fly}_hq${hq}_$rep
      timeout 600 python bench.py $Q --ffps-fly $fly --hw-queues $hq --verify 32 > $OUT/$n.json 2> $OUT/$n.err; show $n
    done
  done
done
timeout 600 python bench.py $Q --ffps-fly 1 --steps 20 --warmup 5 > $OUT/fly1_20steps.json 2> $OUT/fly1_20steps.err; show fly1_20steps
timeout 600 python bench.py $Q --ffps-fly 0 --steps 20 --warmup 5 > $OUT/fly0_20steps.json 2> $OUT/fly0_20steps.err; show fly0_20steps
timeout 600 python bench.py $Q --ffps-fly 1 --data rings64 > $OUT/fly1_rings64.json 2> $OUT/fly1_rings64.err; show fly1_rings64
timeout 600 python bench.py $Q --ffps-fly 0 --data rings64 > $OUT/fly0_rings64.json 2> $OUT/fly0_rings64.err; show fly0_rings64
