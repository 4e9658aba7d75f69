export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "ball or query or group" -p no:cacheprovider 2>&1 | tail -3
for B in 4 16; do
python bench.py --workload group --batch $((B*8)) --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/group_share_b$B.json 2> gpurun_out/group_share_b$B.err
python - <<PY
import json
d=json.loads(open('gpurun_out/group_share_b$B.json').read().strip().splitlines()[-1])
print('group frames', $B*8, 'value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('achieved'))
for s in d.get('stages',[]):
    if 'ball' in s.get('label',''): print('   ', s['label'], s['avg_ms'], s.get('calls'))
print({k:v for k,v in d.items() if 'graph' in k or 'frac' in k})
PY
done
