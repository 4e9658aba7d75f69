#!/bin/bash
# round 4: the staged executor against the 16-slot one, at 2 / 4 / 16 hardware queues, 20 and 512 steps
O=gpurun_out/r04_exec; mkdir -p $O
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
B="python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
val() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'window', c.get('timed_window_ms'), 'alone', c.get('one_package_alone_ms'), 'hwq', c.get('hw_queues'), 'inflight', c.get('steps_in_flight_mean'), 'sclk', c.get('sclk_mhz'), 'prime', (c.get('priming') or {}).get('wall_ms'))
    for r in c['timed_packages_ms']['rows'][:8]: print('    ', r)
except Exception as e: print(sys.argv[1], 'ERR', e)
P
}
run() { # name, extra args
  n=$1; shift
  $B "$@" > $O/$n.json 2> $O/$n.err; val $O/$n.json
}
for st in 20 512; do
  w=5; [ $st = 512 ] && w=64
  run staged_c8_s$st --executor staged --coalesce 8 --steps $st --warmup $w
  run staged_c4_s$st --executor staged --coalesce 4 --steps $st --warmup $w
  run staged_c5_p5_s$st --executor staged --coalesce 5 --streams 5 --steps $st --warmup $w
  run slots16_c4_s$st --executor slots --streams 16 --coalesce 4 --steps $st --warmup $w
done
for q in 2 4 16; do
  run staged_c8_s20_q$q --executor staged --coalesce 8 --steps 20 --warmup 5 --hw-queues $q
  run staged_c8_s512_q$q --executor staged --coalesce 8 --steps 512 --warmup 64 --hw-queues $q
  run slots16_c4_s20_q$q --executor slots --streams 16 --coalesce 4 --steps 20 --warmup 5 --hw-queues $q
  run slots16_c4_s512_q$q --executor slots --streams 16 --coalesce 4 --steps 512 --warmup 64 --hw-queues $q
done
