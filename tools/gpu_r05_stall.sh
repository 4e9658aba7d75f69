#!/bin/bash
# where does a 20-step window lose tens of ms on the HOST?  Run as the FIRST command of a fresh lease:
#   gpurun --timeout 900 -- 'bash tools/gpu_r05_stall.sh'
# Every line = `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command) under one disturbance, once with the
# round-4 behaviour (--rehearse 0) and once with the rehearsed region; the flat host diagnostics of config are printed.
OUT=gpurun_out/r05_stall; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0"
show() { python - "$1" <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_stall/%s.json" % tag).read().strip().splitlines()[-1]); c = d["config"]
    keys = ["timed_window_ms", "probe_window_ms", "rehearsal_ms_first", "rehearsal_ms_last", "host_issue_total_ms", "host_stall_max_ms", "host_stall_at",
            "host_issue_median_us", "majflt", "minflt", "nvcsw", "nivcsw", "cpu_user_ms", "cpu_sys_ms", "cgroup_nr_throttled", "cgroup_throttled_us",
            "ioctl_calls", "ioctl_max_us", "pkg0_reached_ms", "pkg0_done_ms", "pkg1_reached_ms", "pkg1_done_ms", "sclk_before", "sclk_after", "load_avg_1min"]
    print("%-26s %9.1f f/s | " % (tag, d["value"]) + " ".join("%s=%s" % (k.replace("host_", "h_"), c.get(k)) for k in keys if c.get(k) is not None))
except Exception as e:
    print(tag, "failed", e)
P
}
run() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; }
echo "== facts"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/self/cgroup | head -3; cat /proc/loadavg; which rocm-smi amd-smi strace perf 2>/dev/null
grep -c processor /proc/cpuinfo; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
echo "== 1. cold: first command of the lease, round-4 behaviour (no rehearsal), ioctl interposer"
LD_PRELOAD=$PWD/tools/microbench/libioctl_trace.so run cold_norehearse python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse 0 $Q
echo "== 2. rehearsed (default), ioctl interposer"
LD_PRELOAD=$PWD/tools/microbench/libioctl_trace.so run warm_default_ioctl python bench.py --gpus 1 --steps 20 --warmup 5 $Q
run warm_default python bench.py --gpus 1 --steps 20 --warmup 5 $Q
run warm_norehearse python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse 0 $Q
echo "== 3. a concurrent SMI poll loop (the driver's harness runs a sampler)"
( while true; do rocm-smi --showuse --showmemuse --showpower --showclocks --json > /dev/null 2>&1; amd-smi metric --json > /dev/null 2>&1; done ) & SMI=$!
for i in 1 2 3; do run smi_norehearse_$i python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse 0 $Q; done
for i in 1 2 3; do run smi_default_$i python bench.py --gpus 1 --steps 20 --warmup 5 $Q; done
kill $SMI; wait $SMI 2>/dev/null
echo "== 4. every host core busy"
PIDS=""; for i in $(seq $(nproc)); do ( exec timeout 60 sh -c 'while :; do :; done' ) & PIDS="$PIDS $!"; done
sleep 1; cat /proc/loadavg
run hog_norehearse python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse 0 $Q
run hog_default python bench.py --gpus 1 --steps 20 --warmup 5 $Q
run hog_default_pinned python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-affinity auto $Q
kill $PIDS 2>/dev/null; wait 2>/dev/null
echo "== 5. the whole process on ONE core (taskset), then on two"
run one_core_norehearse taskset -c 3 python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse 0 $Q
run one_core_default taskset -c 3 python bench.py --gpus 1 --steps 20 --warmup 5 $Q
run two_cores_default taskset -c 3,4 python bench.py --gpus 1 --steps 20 --warmup 5 $Q
echo "== 6. page cache dropped (if permitted)"
sync; if echo 3 > /proc/sys/vm/drop_caches 2>/dev/null; then echo dropped; else echo "drop_caches not permitted"; fi
run dropped_norehearse python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse 0 $Q
sync; echo 3 > /proc/sys/vm/drop_caches 2>/dev/null
run dropped_default python bench.py --gpus 1 --steps 20 --warmup 5 $Q
echo "== 7. the driver's exact command (CPU baseline + other executor), SMI loop running"
( while true; do rocm-smi --showuse --json > /dev/null 2>&1; sleep 0.2; done ) & SMI=$!
run driver_cmd_smi python3 bench.py --gpus 1 --steps 20 --warmup 5
kill $SMI; wait $SMI 2>/dev/null
echo "== done"
