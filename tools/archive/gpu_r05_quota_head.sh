#!/bin/bash
# the driver's command at the final HEAD under a CPU quota of 1.0 and 0.5 cores (default path: eager launches chosen by bench.py itself)
OUT=gpurun_out/r05_quota_head; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 8"
B="python bench.py --gpus 1 --steps 20 --warmup 5"
show() { python - "$1" <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_quota_head/%s.json" % tag).read().strip().splitlines()[-1]); c = d["config"]
    keys = ["hip_graphs", "timed_window_ms", "probe_window_ms", "rehearsals", "host_stall_max_ms", "cgroup_cpu_quota_cores", "cgroup_nr_throttled_since_priming"]
    print("%-16s %9.1f f/s verify=%s | " % (tag, d["value"], d["verify"]["all_equal_eager"]) + " ".join("%s=%s" % (k, c.get(k)) for k in keys))
except Exception as e:
    print(tag, "failed", e)
P
}
CG=/sys/fs/cgroup
mkdir $CG/rest 2>/dev/null && for p in $(cat $CG/cgroup.procs); do echo $p > $CG/rest/cgroup.procs 2>/dev/null; done
echo "+cpu" > $CG/cgroup.subtree_control 2>/dev/null || { echo "cgroup child not permitted"; exit 0; }
for q in 100000 50000; do
  mkdir -p $CG/q$q; echo "$q 100000" > $CG/q$q/cpu.max
  for i in 1 2 3; do
    timeout 900 sh -c "echo \$\$ > $CG/q$q/cgroup.procs; exec $B $Q" > $OUT/q${q}_$i.json 2> $OUT/q${q}_$i.err; show q${q}_$i
  done
done
