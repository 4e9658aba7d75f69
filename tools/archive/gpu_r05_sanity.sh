export TMPDIR=/tmp PYTHONUNBUFFERED=1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/sanity_default.json 2> gpurun_out/sanity_default.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/sanity_default.json").read().strip().splitlines()[-1]); c = d["config"]
print("default", d["value"], c["hip_graphs"], c["graphs_note"], c["timed_window_ms"], d["cpu_baseline"]["value"], d["roofline"]["frac"])
P
CG=/sys/fs/cgroup
mkdir $CG/rest 2>/dev/null && for p in $(cat $CG/cgroup.procs); do echo $p > $CG/rest/cgroup.procs 2>/dev/null; done
echo "+cpu" > $CG/cgroup.subtree_control 2>/dev/null
mkdir -p $CG/q1; echo "100000 100000" > $CG/q1/cpu.max
sh -c "echo \$\$ > $CG/q1/cgroup.procs; exec python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-executor --profile-iters 0" > gpurun_out/sanity_q1.json 2> gpurun_out/sanity_q1.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/sanity_q1.json").read().strip().splitlines()[-1]); c = d["config"]
print("quota 1.0", d["value"], c["hip_graphs"], c["graphs_note"], c["timed_window_ms"], c["cgroup_nr_throttled"], d["verify"]["all_equal_eager"])
P
