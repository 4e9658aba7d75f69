#!/bin/bash
# round 4, pass 4: big frames through graphs, the new gather path, group workload as one graph, L2 stream rates
O=gpurun_out/r04_pass4; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "beyond or eager_pipeline" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gather or group_and" 2>&1 | tail -3
val() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'window', c.get('timed_window_ms'), 'alone', c.get('one_package_alone_ms'), 'graphs', c.get('hip_graphs'), 'verify', (d.get('verify') or {}).get('all_equal_eager'), 'roofline', (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
P
}
for c in 1 2 4; do
  timeout 600 python bench.py --workload configs4 --coalesce $c --no-cpu-baseline --no-other-executor > $O/configs4_c$c.json 2> $O/configs4_c$c.err; val $O/configs4_c$c.json
done
timeout 600 python bench.py --workload configs4 --executor slots --no-cpu-baseline --no-other-executor > $O/configs4_slots_eager.json 2> $O/configs4_slots_eager.err; val $O/configs4_slots_eager.json
for b in 8 32 128; do
  timeout 600 python bench.py --workload group --batch $b --no-cpu-baseline > $O/group_b$b.json 2> $O/group_b$b.err
  python - $O/group_b$b.json <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], 'frac(events)', r['frac'], 'graph', r.get('sequence_as_one_graph'), 'group_only', r['group_point_only'], 'bq_ms', r['ball_query_only_ms'])
P
done
cd tools/microbench && for a in "1441792 24 64 16 8" "1441792 48 64 8 8" "1441792 96 64 4 8" "1441792 24 64 16 4" "4194304 24 64 16 8" "262144 24 64 16 8"; do ./l2_stream $a; done 2>&1 | tee ../../$O/l2_stream.txt
