#!/bin/bash
# copy the summaries of a tools/gpu_round_final_r04.sh pass (gpurun_out/$1) into profiles/ under the round prefix $2
S=gpurun_out/$1; P=profiles; R=${2:-r04}
for n in 20steps default host_input dup10 dense rings64 2ranks_shared configs2 configs4 group_b8 group_b32 group_b128; do
  [ -s $S/bench_$n.json ] && grep '^{' $S/bench_$n.json | tail -1 > $P/${R}_bench_$n.json
done
cp $S/pytest_gpu.log $P/${R}_pytest_gpu.log
cp $S/prof/summary.txt $P/${R}_rocprof_summary.txt
cp $S/prof/traffic.json $P/${R}_traffic.json
f=$(find $S/prof/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_kernel_stats_single_stream.csv
cp $S/ops_rocprof_summary.txt $P/${R}_ops_rocprof_summary.txt 2>/dev/null
[ -s $S/ops_microbench.jsonl ] && cp $S/ops_microbench.jsonl $P/${R}_ops_microbench.jsonl
cp $S/ablation.txt $P/${R}_ablation.txt
cp $S/sweep_coalesce.txt $P/${R}_sweep_packages.txt
cp $S/l2_stream.txt $P/${R}_l2_stream.txt
cp $S/ffps_fly_probe.txt $P/${R}_ffps_fly_probe.txt
ls -la $P | grep " ${R}_" | awk '{print $5, $9}'
