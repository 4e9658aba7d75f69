#!/bin/bash
# rocprofv3 kernel trace + PMC passes over FULL 128-frame launches only (tools/prof_128f.py), joined per C-ABI call by
# tools/summarize_128f.py.  usage: bash tools/gpu_prof128.sh TAG [data]
TAG=${1:-r05_prof128}; DATA=${2:-default}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CMD="python $GRAFT_REPO_ROOT/tools/prof_128f.py $OUT $DATA 6"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p128 -- $CMD > $OUT/trace.log 2>&1; tail -1 $OUT/trace.log
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$n -o p128 -- $CMD > $OUT/pmc_$n.log 2>&1; tail -1 $OUT/pmc_$n.log
done
cd $GRAFT_REPO_ROOT
python tools/summarize_128f.py $OUT > $OUT/rooflines_128f.txt 2> $OUT/summarize.err; head -60 $OUT/rooflines_128f.txt; tail -3 $OUT/summarize.err
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
