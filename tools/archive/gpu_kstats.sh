#!/bin/bash
# per-kernel average durations of one single-stream bench run (rocprofv3 --kernel-trace --stats); $1 = tag, $2 = grep filter
TAG=${1:-ks}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-uncoalesced --profile-iters 1 --verify 0 ${BENCH_EXTRA}"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<P
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:40]:
        n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:90]
        print("%-90s calls %5s avg %9.1f us  %5.1f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
P
find $OUT -name "*kernel_trace.csv" -size +20M -delete
