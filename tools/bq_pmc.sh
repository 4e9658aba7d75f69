#!/bin/bash
# SQ counters of the grid ball query kernels over 128-frame backbone passes: bash tools/bq_pmc.sh TAG [data] [lib]
TAG=${1:-bq_pmc}; DATA=${2:-rings64}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
[ -n "$3" ] && export SA3D_LIB=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants/$3
CMD="python $GRAFT_REPO_ROOT/tools/bq_bench.py ${FRAMES:-128} $DATA 3"
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc_$i.log 2>&1; tail -1 $OUT/pmc_$i.log
done
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "bq_grid" not in kn: continue
        key = (kn.split("(")[0][-40:], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/summary.txt", "w") as fo:
    for key in sorted(agg):
        d = sorted(dur[key]); line = "%s grid=%s  launches=%d  median %.1f us" % (key[0], key[1], max(cnt[key].values()), d[len(d)//2] / 1e3)
        print(line); fo.write(line + "\n")
        for c in sorted(agg[key]):
            line = "    %-24s %14.0f per launch" % (c, agg[key][c] / cnt[key][c]); print(line); fo.write(line + "\n")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
