#!/bin/bash
# copy the summaries of a tools/gpu_round_final_r06.sh pass (gpurun_out/$1) into profiles/ under the round prefix $2
S=gpurun_out/$1; P=profiles; R=${2:-r06}
for n in 20steps_cold 20steps default host_input dup10 dense rings64 2ranks_shared configs2 configs4 group_b8 group_b32 group_b128 group_b128_shared_grid detector; do
  [ -s $S/bench_$n.json ] && grep '^{' $S/bench_$n.json | tail -1 > $P/${R}_bench_$n.json
done
cp $S/pytest_gpu.log $P/${R}_pytest_gpu.log
cp $S/prof128_default/rooflines_128f.txt $P/${R}_rooflines_128f.txt
cp $S/prof128_default/kernel_stats_128f.csv $P/${R}_kernel_stats_128f.csv
cp $S/prof128_default/traffic.json $P/${R}_traffic.json
cp $S/prof128_rings64/rooflines_128f.txt $P/${R}_rooflines_128f_rings64.txt
f=$(find $S/prof128_default/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_rocprofv3_kernel_stats.csv
ls -la $P | grep " ${R}_" | awk '{print $5, $9}'
