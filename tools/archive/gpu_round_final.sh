#!/bin/bash
# One gpurun call at the end of a round: smoke, the whole GPU suite, the bench lines that go to profiles/, the op
# microbench under rocprofv3, and the rocprofv3 kernel-stats + PMC passes of a single-stream run.  Everything lands in
# gpurun_out/$TAG/.      usage: gpurun --timeout 2400 -- 'bash tools/gpu_round_final.sh r03'
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=50 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
show() { python - <<P
import json
try:
    d = json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["unit"], "ms/step", d["ms_per_step"], "lat", d.get("single_stream_batch_latency_ms"), "n_gpus", d["n_gpus"],
          "verify", (d.get("verify") or {}).get("all_equal_eager"), "rows", (d.get("mlp_rows_per_step") or {}).get("evaluated_frac"),
          "fps_eval", (d.get("roofline") or {}).get("evaluated_frac"), "roofline", (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("$1 failed", e)
P
}
echo "== bench default"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; show default
for i in 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-uncoalesced --profile-iters 0 --verify 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default again', d['value'], d['ms_per_step'])"; done
echo "== bench 20 steps"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err; show 20steps
for v in dup10 dense; do timeout 600 python bench.py --data $v --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err; show $v; done
timeout 600 python bench.py --gpus 2 --allow-shared-device --steps 64 --warmup 16 --no-cpu-baseline > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; show 2ranks_shared
timeout 600 python bench.py --workload configs2 --no-cpu-baseline > $OUT/bench_configs2.json 2> $OUT/bench_configs2.err; show configs2
timeout 600 python bench.py --workload configs4 --no-cpu-baseline > $OUT/bench_configs4.json 2> $OUT/bench_configs4.err; show configs4
timeout 600 python bench.py --workload group > $OUT/bench_group.json 2> $OUT/bench_group.err; show group
echo "== op microbench under rocprofv3"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/ops_trace -o ops -- python $GRAFT_REPO_ROOT/tools/bench_ops.py > $GRAFT_REPO_ROOT/$OUT/ops_microbench.jsonl 2> $GRAFT_REPO_ROOT/$OUT/ops_microbench.err)
python - <<P
import csv, glob
for f in glob.glob("$OUT/ops_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("$OUT/ops_rocprof_summary.txt", "w") as o:
        for r in rows[:30]:
            n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
            o.write("%-80s calls %6s  avg %10.1f us  total %10.1f us\n" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
print(open("$OUT/ops_rocprof_summary.txt").read()[:1500])
P
find $OUT/ops_trace -name "*kernel_trace.csv" -size +10M -delete
echo "== rocprof single stream"; bash tools/gpu_prof.sh $TAG/prof > $OUT/prof.log 2>&1; tail -5 $OUT/prof.log
echo "== slots x batches per replay"; bash tools/gpu_sweep_coalesce.sh 16:1 16:2 16:4 16:8 12:4 8:4 > $OUT/sweep_coalesce.txt 2>&1; cat $OUT/sweep_coalesce.txt
echo "== marginal cost of the kernel classes"; bash tools/ablate.sh > $OUT/ablation.txt 2>&1; cat $OUT/ablation.txt
echo "== done"
