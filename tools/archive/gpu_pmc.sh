#!/bin/bash
# rocprofv3 PMC passes (kernel-trace + counters, nothing else) of one single-stream bench run; per-kernel means for the
# kernels matching $2 (regex).  usage: bash tools/gpu_pmc.sh TAG 'mlp_gemm' "CTR1 CTR2 ..." "CTR3 ..."
TAG=$1; FILT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-uncoalesced --profile-iters 0 --verify 0 ${BENCH_EXTRA}"
cd /tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -o bench -- $CMD > $OUT/pmc_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<P
import csv, glob, collections, re
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        if not re.search(r"$FILT", k):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for k in sorted(agg):
        print("%-52s n=%3d " % (k, cnt[k]) + " ".join("%s=%.4g" % (c, v / cnt[k]) for c, v in sorted(agg[k].items())))
P
find $OUT -name "*.csv" -size +20M -delete
