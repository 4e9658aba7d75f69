#!/bin/bash
O=gpurun_out/r04_pass5; mkdir -p $O
python tools/ffps_fly_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/ffps_fly_probe.txt
val() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'window', c.get('timed_window_ms'), 'alone', c.get('one_package_alone_ms'), 'graphs', c.get('hip_graphs'), 'verify', (d.get('verify') or {}).get('all_equal_eager'), 'roofline', (d.get('roofline') or {}).get('frac'), 'gather', c.get('gather'))
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
P
}
for c in 2 4; do
  timeout 600 python bench.py --workload configs4 --coalesce $c --no-cpu-baseline --no-other-executor > $O/configs4_c$c.json 2> $O/configs4_c$c.err; val $O/configs4_c$c.json
done
timeout 600 python bench.py --gpus 2 --allow-shared-device --steps 64 --warmup 16 --no-cpu-baseline --no-other-executor --profile-iters 0 > $O/two_ranks_shared.json 2> $O/two_ranks_shared.err; val $O/two_ranks_shared.json
timeout 300 python -m pytest tests/test_pipeline_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "beyond or cooperative" 2>&1 | tail -3
