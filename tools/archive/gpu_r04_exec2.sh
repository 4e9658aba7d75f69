#!/bin/bash
# round 4: staged executor -- package size, linear graphs, hardware queues; 20 and 512 steps
O=gpurun_out/r04_exec2; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fp16" 2>&1 | tail -3
B="python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 16"
val() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'window', c.get('timed_window_ms'), 'alone', c.get('one_package_alone_ms'), 'hwq', c.get('hw_queues'), 'inflight', c.get('steps_in_flight_mean'), 'verify', (d.get('verify') or {}).get('all_equal_eager'))
    for r in c['timed_packages_ms']['rows'][:5]: print('    ', r)
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
P
}
run() { n=$1; shift; $B "$@" > $O/$n.json 2> $O/$n.err; val $O/$n.json; }
for lin in 0 1; do
 for c in 8 16 20; do
  for q in 0 8; do
    hq=""; [ $q != 0 ] && hq="--hw-queues $q"
    run staged_c${c}_lin${lin}_q${q}_s20 --executor staged --coalesce $c --linear-graphs $lin --steps 20 --warmup 5 $hq
    run staged_c${c}_lin${lin}_q${q}_s512 --executor staged --coalesce $c --linear-graphs $lin --steps 512 --warmup 64 $hq
  done
 done
done
run slots16_c4_lin1_s512 --executor slots --streams 16 --coalesce 4 --linear-graphs 1 --steps 512 --warmup 64
run slots16_c4_lin1_s20 --executor slots --streams 16 --coalesce 4 --linear-graphs 1 --steps 20 --warmup 5
