#!/bin/bash
# round 5: the staged executor WITHOUT hipGraphs (--graphs 0: eager launches on the same three streams) -- throughput, host
# CPU, and behaviour under a CPU quota (the runtime thread that spins while graph work is pending does not exist then)
OUT=gpurun_out/r05_eager; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 16"
show() { python - "$1" <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_eager/%s.json" % tag).read().strip().splitlines()[-1]); c = d["config"]
    keys = ["hip_graphs", "timed_window_ms", "probe_window_ms", "rehearsals", "rehearsal_ms_max", "host_issue_total_ms", "host_stall_max_ms", "cpu_user_ms", "cpu_sys_ms", "cgroup_cpu_quota_cores", "cgroup_nr_throttled", "process_cpu_cores_since_priming"]
    print("%-24s %9.1f f/s | " % (tag, d["value"]) + " ".join("%s=%s" % (k, c.get(k)) for k in keys) + " verify=%s" % (d.get("verify") or {}).get("all_equal_eager"))
except Exception as e:
    print(tag, "failed", e)
P
}
run() { tag=$1; shift; timeout 600 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; }
for i in 1 2 3; do run g1_20_$i python bench.py --gpus 1 --steps 20 --warmup 5 $Q; run g0_20_$i python bench.py --gpus 1 --steps 20 --warmup 5 --graphs 0 $Q; done
for i in 1 2; do run g1_512_$i python bench.py $Q; run g0_512_$i python bench.py --graphs 0 $Q; done
run g0_rings64 python bench.py --graphs 0 --data rings64 $Q
CG=/sys/fs/cgroup
mkdir $CG/rest 2>/dev/null && for p in $(cat $CG/cgroup.procs); do echo $p > $CG/rest/cgroup.procs 2>/dev/null; done
echo "+cpu" > $CG/cgroup.subtree_control 2>/dev/null || { echo "cgroup child not permitted"; exit 0; }
for q in 30000 50000 100000; do
  mkdir -p $CG/q$q; echo "$q 100000" > $CG/q$q/cpu.max
  for i in 1 2 3 4 5; do
    timeout 900 sh -c "echo \$\$ > $CG/q$q/cgroup.procs; exec python bench.py --gpus 1 --steps 20 --warmup 5 --graphs 0 $Q" > $OUT/q${q}_eager_$i.json 2> $OUT/q${q}_eager_$i.err; show q${q}_eager_$i
  done
done
echo "== done"
