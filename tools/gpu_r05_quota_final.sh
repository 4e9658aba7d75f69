#!/bin/bash
# round 5, final code: the driver's command under CPU quotas (child cgroup), sleeping waits (default) against spinning
OUT=gpurun_out/r05_quota_final; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0"
B="python bench.py --gpus 1 --steps 20 --warmup 5"
show() { python - "$1" <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_quota_final/%s.json" % tag).read().strip().splitlines()[-1]); c = d["config"]
    keys = ["timed_window_ms", "probe_window_ms", "rehearsals", "rehearsal_ms_min", "rehearsal_ms_max", "host_stall_max_ms", "cgroup_cpu_quota_cores", "cgroup_nr_throttled", "cgroup_nr_throttled_since_priming", "process_cpu_cores_since_priming"]
    print("%-24s %9.1f f/s | " % (tag, d["value"]) + " ".join("%s=%s" % (k, c.get(k)) for k in keys))
except Exception as e:
    print(tag, "failed", e)
P
}
CG=/sys/fs/cgroup
mkdir $CG/rest 2>/dev/null && for p in $(cat $CG/cgroup.procs); do echo $p > $CG/rest/cgroup.procs 2>/dev/null; done
echo "+cpu" > $CG/cgroup.subtree_control 2>/dev/null || { echo "cgroup child not permitted"; exit 0; }
for q in 50000 100000 130000; do
  mkdir -p $CG/q$q; echo "$q 100000" > $CG/q$q/cpu.max
  for mode in sleep spin; do
    for i in 1 2 3 4 5; do
      flag=""; [ $mode = spin ] && flag="--blocking-wait 0"
      timeout 900 sh -c "echo \$\$ > $CG/q$q/cgroup.procs; exec $B $flag $Q" > $OUT/q${q}_${mode}_$i.json 2> $OUT/q${q}_${mode}_$i.err; show q${q}_${mode}_$i
    done
  done
done
echo "== done"
