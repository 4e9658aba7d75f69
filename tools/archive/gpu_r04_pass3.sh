#!/bin/bash
# round 4, pass 3: full GPU suite on the new executor / gather / ball-query kernels, then the bench lines
O=gpurun_out/r04_pass3; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
val() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'window', c.get('timed_window_ms'), 'alone', c.get('one_package_alone_ms'), 'hwq', c.get('hw_queues'), 'verify', (d.get('verify') or {}).get('all_equal_eager'), 'roofline', (d.get('roofline') or {}).get('frac'), 'other', (c.get('other_executor') or {}).get('value'))
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
P
}
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; val $O/bench_20.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; val $O/bench_default.json
for d in dup10 dense rings64; do
  python bench.py --data $d --no-cpu-baseline --no-other-executor > $O/bench_$d.json 2> $O/bench_$d.err; val $O/bench_$d.json
done
python bench.py --workload group > $O/bench_group.json 2> $O/bench_group.err; tail -c 1500 $O/bench_group.json | head -c 900; echo
python bench.py --workload group --batch 32 > $O/bench_group32.json 2> $O/bench_group32.err; tail -c 1500 $O/bench_group32.json | head -c 900; echo
for ms in 1 2; do
  python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 --main-streams $ms --steps 512 --warmup 64 > $O/ms$ms.json 2> $O/ms$ms.err; val $O/ms$ms.json
  python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 --main-streams $ms --steps 20 --warmup 5 > $O/ms${ms}_s20.json 2> $O/ms${ms}_s20.err; val $O/ms${ms}_s20.json
done
