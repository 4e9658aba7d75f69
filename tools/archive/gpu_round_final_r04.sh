#!/bin/bash
# One gpurun call: smoke, the whole GPU suite, the bench lines that go to profiles/, the op microbench under rocprofv3,
# the rocprofv3 kernel-stats + PMC passes of a single-stream run at the product's launch shape, the package-size sweep
# and the ablation.  Everything lands in gpurun_out/$TAG/.   usage: gpurun --timeout 3000 -- 'bash tools/gpu_round_final_r04.sh r04'
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --maxfail=50 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
show() { python - <<P
import json
try:
    d = json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1]); c = d["config"]
    print("$1", d["value"], d["unit"], "ms/step", d["ms_per_step"], "alone", c.get("one_package_alone_ms"), "n_gpus", d["n_gpus"], "hwq", c.get("hw_queues"),
          "verify", (d.get("verify") or {}).get("all_equal_eager"), "rows", (d.get("mlp_rows_per_step") or {}).get("evaluated_frac"),
          "roofline", (d.get("roofline") or {}).get("frac"), "other", (c.get("other_executor") or {}).get("value"), "gather", (c.get("gather") or {}).get("ranks_seen"))
except Exception as e:
    print("$1 failed", e)
P
}
echo "== bench 20 steps (the driver's command)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err; show 20steps
echo "== bench default"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; show default
for i in 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default again', d['value'], d['ms_per_step'])"; done
timeout 600 python bench.py --host-input --no-cpu-baseline --no-other-executor --profile-iters 0 > $OUT/bench_host_input.json 2> $OUT/bench_host_input.err; show host_input
for v in dup10 dense rings64; do timeout 600 python bench.py --data $v --no-cpu-baseline --no-other-executor > $OUT/bench_$v.json 2> $OUT/bench_$v.err; show $v; done
timeout 600 python bench.py --gpus 2 --allow-shared-device --steps 64 --warmup 16 --no-cpu-baseline --no-other-executor > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; show 2ranks_shared
timeout 600 python bench.py --workload configs2 --no-cpu-baseline > $OUT/bench_configs2.json 2> $OUT/bench_configs2.err; show configs2
timeout 600 python bench.py --workload configs4 --no-cpu-baseline --no-other-executor > $OUT/bench_configs4.json 2> $OUT/bench_configs4.err; show configs4
for b in 8 32 128; do timeout 600 python bench.py --workload group --batch $b > $OUT/bench_group_b$b.json 2> $OUT/bench_group_b$b.err; show group_b$b; done
echo "== on-the-fly F-FPS probe"; timeout 300 python tools/ffps_fly_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/ffps_fly_probe.txt; cat $OUT/ffps_fly_probe.txt
echo "== L2 stream rates"; (cd tools/microbench && for a in "1441792 24 64 16 8" "1441792 48 64 8 8" "1441792 96 64 4 8"; do ./l2_stream $a; done) > $OUT/l2_stream.txt 2>&1; cat $OUT/l2_stream.txt
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
try: print(open("$OUT/ops_rocprof_summary.txt").read()[:1500])
except Exception as e: print("no ops summary", e)
P
find $OUT/ops_trace -name "*kernel_trace.csv" -size +10M -delete
echo "== rocprof single stream, 128 frames per launch"; bash tools/gpu_prof.sh $TAG/prof > $OUT/prof.log 2>&1; tail -5 $OUT/prof.log
echo "== packages x batches per package (staged)"; EXECUTOR=staged bash tools/gpu_sweep_coalesce.sh 4:8 4:16 4:24 3:16 6:16 > $OUT/sweep_coalesce.txt 2>&1; cat $OUT/sweep_coalesce.txt
echo "== marginal cost of the kernel classes (staged executor)"; ABLATE_MODE=staged ABLATE_COALESCE=16 ABLATE_STREAMS=4 bash tools/ablate.sh > $OUT/ablation.txt 2>&1; cat $OUT/ablation.txt
echo "== done"
