#!/bin/bash
# One gpurun call at the end of a round: build check, smoke, the whole GPU suite, the bench lines that go to profiles/,
# the 16-stream stage trace / saturation tables and the rocprofv3 passes.  Everything lands in gpurun_out/$TAG/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round_final.sh r02'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== build check"; python -c "import __graft_entry__ as g; g.build(); print('build ok')" 2>&1 | tail -2
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=50 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
echo "== bench default"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json | head -c 400; echo
for i in 2 3; do timeout 300 python bench.py --no-cpu-baseline --profile-iters 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default again', d['value'], d['ms_per_step'])"; done
echo "== bench 20 steps"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err; python -c "import json; d=json.loads(open('$OUT/bench_20steps.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== configs2 / configs4"; timeout 600 python bench.py --workload configs2 --no-cpu-baseline > $OUT/bench_configs2.json 2> $OUT/bench_configs2.err; timeout 600 python bench.py --workload configs4 --no-cpu-baseline > $OUT/bench_configs4.json 2> $OUT/bench_configs4.err
python - <<P
import json
for w in ("configs2", "configs4"):
    try:
        d = json.loads(open("$OUT/bench_%s.json" % w).read().strip().splitlines()[-1]); print(w, d["value"], d["unit"], d["ms_per_step"])
    except Exception as e:
        print(w, "failed", e)
P
echo "== stage trace / saturation"; timeout 300 python tools/stage_trace.py > $OUT/stage_trace.txt 2> $OUT/stage_trace.err; head -3 $OUT/stage_trace.txt
timeout 300 python tools/saturation.py > $OUT/saturation.txt 2> $OUT/saturation.err; tail -2 $OUT/saturation.txt
(cd tools/microbench && GPU_MAX_HW_QUEUES=16 timeout 100 ./dispatch_contention 8 8) > $OUT/dispatch_contention.txt 2>&1
echo "== rocprof"; bash tools/gpu_prof.sh $TAG/prof > $OUT/prof.log 2>&1; tail -5 $OUT/prof.log
echo "== done"
