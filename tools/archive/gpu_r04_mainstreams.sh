for ms in 2 3 4; do for p in 4 6; do
python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 --main-streams $ms --streams $p --steps 512 --warmup 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('main $ms packages $p 512:', d['value'], d['ms_per_step'], d['config']['hw_queues'], d['config']['streams_used'])"
done; done
for ms in 3 4; do
GPU_MAX_HW_QUEUES=8 python bench.py --gpus 1 --no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0 --main-streams $ms --streams 6 --steps 512 --warmup 64 --hw-queues 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q8 main $ms packages 6 512:', d['value'], d['ms_per_step'], d['config']['hw_queues'])"
done
