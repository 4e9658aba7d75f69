#!/bin/bash
# rocprofv3 kernel trace of the HEADLINE configuration (16 streams, hipGraph replay) + concurrency summary.
# usage: gpurun -- 'bash tools/gpu_trace16.sh [tag]'   -> gpurun_out/<tag>/{trace16_summary.txt,kernel_stats.csv}
TAG=${1:-t16}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 24 --no-cpu-baseline --profile-iters 0 > $OUT/bench_under_trace.json 2> $OUT/trace.log
cd $GRAFT_REPO_ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python tools/concurrency.py $T > $OUT/trace16_summary.txt 2>&1; cat $OUT/trace16_summary.txt
cp $S $OUT/kernel_stats.csv 2>/dev/null
find $OUT/trace -name "*kernel_trace.csv" -size +20M -delete
