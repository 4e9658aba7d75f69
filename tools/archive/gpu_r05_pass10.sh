#!/bin/bash
# round 5, GPU pass 10: where do the waves of bq_grid_query_kernel spend their cycles?  extra SQ counters, rings64 + default
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_pass10; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt; wc -l $OUT/sq_counters.txt
for d in rings64 default; do
  i=0
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SENDMSG SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${d}_$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_128f.py $OUT/calls_$d $d 2 > $OUT/${d}_$i.log 2>&1; tail -1 $OUT/${d}_$i.log
  done
  python - $OUT $d <<'P'
import csv, glob, sys, collections
root, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (root, d), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bq_grid_query" in k and r["Grid_Size"] and int(r["Grid_Size"]) > 20000000:      # the layer-1 launch (131072 workgroups x 256 threads)
            agg["L1"][r["Counter_Name"]] += float(r["Counter_Value"]); n["L1"][r["Counter_Name"]] += 1
for k in agg:
    print(d, k, " ".join("%s=%.4g" % (c, v / n[k][c]) for c, v in sorted(agg[k].items())))
P
done
find $OUT -name "*.csv" -size +5M -delete
echo "== done"
