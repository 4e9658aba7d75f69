#!/bin/bash
# round 5, GPU pass 3: 4-row granules (parity + A/B timing), CPU burn while the GPU works, quota runs with the sleeping wait
OUT=gpurun_out/r05_pass3; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== MLP / backbone / pipeline tests"
timeout 1500 python -m pytest tests -m gpu -q -x -k "mlp or backbone or pipeline or granule or dense or vote or abi" -p no:cacheprovider -rf > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for d in default rings64; do for g in True False; do echo "== stages at 128 frames, data=$d MLP_GRANULE4=$g"; timeout 300 python tools/stages_at.py 128 data=$d MLP_GRANULE4=$g 2>&1 | grep -v amdgpu.ids | tee $OUT/stages_128_${d}_gr4_$g.txt | grep -i "mlp\|total"; done; done
echo "== CPU while the GPU works"
for m in eager graph; do timeout 120 python tools/cpu_while_gpu_busy.py $m 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cpu_while_busy.txt; done
for v in "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=0" "AMD_DIRECT_DISPATCH=0" "HIP_FORCE_QUEUE_PROFILING=0"; do echo "-- $v"; env $v timeout 120 python tools/cpu_while_gpu_busy.py graph 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cpu_while_busy.txt; done
echo "== bench lines"
Q="--no-cpu-baseline --no-other-executor --profile-iters 0 --verify 0"
show() { python - "$1" <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_pass3/%s.json" % tag).read().strip().splitlines()[-1]); c = d["config"]
    keys = ["timed_window_ms", "probe_window_ms", "rehearsals", "rehearsal_ms_min", "rehearsal_ms_max", "host_issue_total_ms", "host_stall_max_ms",
            "cpu_user_ms", "cpu_sys_ms", "cgroup_cpu_quota_cores", "cgroup_nr_throttled", "cgroup_nr_throttled_since_priming", "cgroup_throttled_us_since_priming", "process_cpu_cores_since_priming"]
    print("%-28s %9.1f f/s | " % (tag, d["value"]) + " ".join("%s=%s" % (k.replace("host_", "h_"), c.get(k)) for k in keys if c.get(k) is not None))
except Exception as e:
    print(tag, "failed", e, open("gpurun_out/r05_pass3/%s.err" % tag).read()[-300:])
P
}
run() { tag=$1; shift; timeout 600 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; }
B="python bench.py --gpus 1 --steps 20 --warmup 5"
run b20 $B $Q; run b20_again $B $Q; run b512 python bench.py $Q; run b512_rings64 python bench.py --data rings64 $Q
CG=/sys/fs/cgroup; ok=1
mkdir $CG/rest 2>/dev/null || ok=0
if [ $ok = 1 ]; then for p in $(cat $CG/cgroup.procs); do echo $p > $CG/rest/cgroup.procs 2>/dev/null; done; echo "+cpu" > $CG/cgroup.subtree_control 2>/dev/null || ok=0; fi
if [ $ok = 1 ]; then
  for q in 100000 150000 200000; do
    mkdir -p $CG/q$q; echo "$q 100000" > $CG/q$q/cpu.max
    runq() { tag=$1; shift; timeout 900 sh -c "echo \$\$ > $CG/q$q/cgroup.procs; exec $*" > $OUT/$tag.json 2> $OUT/$tag.err; show $tag; }
    for i in 1 2 3; do runq quota${q}_spin_$i $B --blocking-wait 0 $Q; done
    for i in 1 2 3; do runq quota${q}_sleep_$i $B $Q; done
  done
else echo "cgroup child not permitted"; fi
echo "== done"
