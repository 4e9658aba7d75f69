B="python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', d['value'], d['ms_per_step'], 'graphs', c['hip_graphs'], 'host_issue', c['host_issue_ms_per_step'], 'verify', d['verify']['all_equal_eager'])"; }
$B --graphs 0 --steps 512 --warmup 64 2>/dev/null | p "eager staged 512"
$B --graphs 0 --steps 20 --warmup 5 2>/dev/null | p "eager staged 20"
$B --coalesce 32 --steps 512 --warmup 64 2>/dev/null | p "coalesce 32 512"
$B --coalesce 32 --pool 512 --steps 512 --warmup 64 2>/dev/null | p "coalesce 32 pool 512"
