#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
rocm-smi --showproductname > $OUT/gpu.txt 2>&1 | head -20
nproc > $OUT/host.txt; grep -m1 'model name' /proc/cpuinfo >> $OUT/host.txt
echo "== build check"; python -c "import __graft_entry__ as g; g.build(); print('build ok')" 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=200 -p no:cacheprovider -rf > $OUT/pytest_gpu.log 2>&1; tail -40 $OUT/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 24 --warmup 6 > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
echo "== bench 1 stream"; timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > $OUT/bench_s1.json 2> $OUT/bench_s1.err; tail -c 600 $OUT/bench_s1.json
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --profile-iters 1 > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -30 $f; done
# keep only the summaries (traces can be large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"
