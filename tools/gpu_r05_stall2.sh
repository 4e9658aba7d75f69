#!/bin/bash
# VERDICT r4 item 1, second pass: the driver's command under a CPU QUOTA (cgroup v2 child of this container's cgroup, if it
# can be created), with per-process SMI queries running, and sleeping vs spinning waits.  Lines as in gpu_r05_stall.sh.
OUT=gpurun_out/r05_stall2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0"
show() { python - "$1" <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_stall2/%s.json" % tag).read().strip().splitlines()[-1]); c = d["config"]
    keys = ["timed_window_ms", "probe_window_ms", "rehearsals", "rehearsal_ms_min", "rehearsal_ms_max", "rehearsal_ms_last", "blocking_wait", "host_issue_total_ms",
            "host_stall_max_ms", "host_stall_at", "nvcsw", "nivcsw", "cpu_user_ms", "cpu_sys_ms", "cgroup_nr_throttled", "cgroup_throttled_us",
            "ioctl_calls", "ioctl_max_us", "ioctl_top", "pkg0_reached_ms", "pkg1_done_ms", "sclk_before", "sclk_after"]
    print("%-28s %9.1f f/s | " % (tag, d["value"]) + " ".join("%s=%s" % (k.replace("host_", "h_"), c.get(k)) for k in keys if c.get(k) is not None))
except Exception as e:
    print(tag, "failed", e, open("gpurun_out/r05_stall2/%s.err" % tag).read()[-300:])
P
}
run() { tag=$1; shift; timeout 600 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; }
B="python bench.py --gpus 1 --steps 20 --warmup 5"
echo "== 0. defaults of this round (rehearsed, sleeping waits) vs spinning waits vs the round-4 region; ioctl interposer on the first"
LD_PRELOAD=$PWD/tools/microbench/libioctl_trace.so run default_ioctl $B $Q
run default $B $Q
run spin $B --blocking-wait 0 $Q
run r4_region $B --blocking-wait 0 --rehearse 0 $Q
run default_512 python bench.py $Q
run spin_512 python bench.py --blocking-wait 0 $Q
echo "== 1. per-process SMI queries in a loop (rocm-smi --showpids / amd-smi process read KFD's per-process sysfs)"
( while true; do rocm-smi --showpids --showpidgpus > /dev/null 2>&1; amd-smi process --json > /dev/null 2>&1; amd-smi monitor -w 1 -i 1 > /dev/null 2>&1; done ) & SMI=$!
for i in 1 2 3; do run smipid_r4_$i $B --blocking-wait 0 --rehearse 0 $Q; done
for i in 1 2; do run smipid_default_$i $B $Q; done
kill $SMI; wait $SMI 2>/dev/null
echo "== 2. CPU quota: a child cgroup with cpu.max = 1 core / 2 cores / 4 cores"
CG=/sys/fs/cgroup
ls -ld $CG; cat $CG/cgroup.controllers 2>/dev/null; cat $CG/cgroup.subtree_control 2>/dev/null; mount | grep cgroup | head -3
ok=1
mkdir $CG/rest 2>/dev/null || ok=0
if [ $ok = 1 ]; then
  for p in $(cat $CG/cgroup.procs); do echo $p > $CG/rest/cgroup.procs 2>/dev/null; done
  echo "+cpu" > $CG/cgroup.subtree_control 2>/dev/null || ok=0
fi
if [ $ok = 1 ]; then
  for q in 100000 200000 400000; do
    mkdir -p $CG/q$q; echo "$q 100000" > $CG/q$q/cpu.max || ok=0
    runq() { tag=$1; shift; timeout 900 sh -c "echo \$\$ > $CG/q$q/cgroup.procs; exec $*" > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; grep -h "nr_throttled\|throttled_usec" $CG/q$q/cpu.stat | tr '\n' ' '; echo; }
    runq quota${q}_r4 $B --blocking-wait 0 --rehearse 0 $Q
    runq quota${q}_spin $B --blocking-wait 0 $Q
    runq quota${q}_default $B $Q
  done
else
  echo "cgroup child not permitted here (read-only or no delegation): quota not emulated"
fi
echo "== done"
