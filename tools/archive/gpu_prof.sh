#!/bin/bash
# rocprofv3 kernel stats + PMC passes of one single-stream bench run.  Summaries land in gpurun_out/$TAG/
TAG=${1:-p}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT && python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
# one slot on one stream, packages of 16 batches = 128 frames: the launch shape of the default (staged) executor, kernels
# strictly one after the other
CMD="python $GRAFT_REPO_ROOT/bench.py --executor slots --streams 1 --coalesce ${PROF_COALESCE:-16} --steps 32 --warmup 16 --no-cpu-baseline --no-other-executor --profile-iters 1 --verify 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$n -o bench -- $CMD > $OUT/pmc_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*.csv" | head -30
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | head -80
find $OUT -name "*kernel_trace.csv" -size +30M -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
