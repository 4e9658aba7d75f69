#!/bin/bash
# round 4, experiment 1: what decides the 20-step number -- hardware queues of this process, and a neighbour's queues
O=gpurun_out/r04_queues; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-uncoalesced --profile-iters 0 --verify 0"
val() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1], d['value'], d['ms_per_step'], d.get('timed_window_ms'), (d.get('overlap') or {}).get('steps_in_flight_mean'))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
}
for q in 2 4 8 16; do GPU_MAX_HW_QUEUES=$q $B > $O/q$q.json 2> $O/q$q.err; val $O/q$q.json; done
for k in 4 8 12 24; do
  python tools/cotenant.py $k 25 > $O/co$k.log 2>&1 &
  P=$!; sleep 9
  $B > $O/co$k.json 2> $O/co$k.err; val $O/co$k.json
  wait $P
done
python tools/cotenant.py 8 25 busy > $O/cobusy8.log 2>&1 &
P=$!; sleep 9; $B > $O/cobusy8.json 2> $O/cobusy8.err; val $O/cobusy8.json; wait $P
