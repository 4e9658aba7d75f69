for ss in 1 2; do for st in "20:5" "20:5" "512:64"; do K=${st%%:*}; W=${st##*:}
python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8 --sampler-streams $ss --steps $K --warmup $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sampler streams $ss steps $K:', d['value'], d['ms_per_step'], d['config']['streams_used'], d['verify']['all_equal_eager'], d['config']['timed_packages_ms']['rows'][:3])"
done; done
for c in "10" "12" "20"; do
python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 --sampler-streams 2 --coalesce $c --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sampler streams 2 coalesce $c steps 20:', d['value'], d['ms_per_step'], d['config']['timed_packages_ms']['rows'][:3])"
done
