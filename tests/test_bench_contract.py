"""The bench lines committed under profiles/ (outputs of `python bench.py` on the GPU box, round 4) carry every field
of the driver's contract, the round-3 additions (distinct frames, verification against eager, recorded environment,
latency-bound roofline for FPS, executed-flop MFMA fraction, batches per package), the round-4 ones (the executor and
what decides a short run inside `config`, where the driver's record keeps them: queues, priming, package timeline), and
their algorithmic work model reproduces SURVEY.md 8d: 30.93 GFLOP of MLP per frame.  Stages / rooflines describe the
calls as the pipeline issues them: one pass over config.frames_per_launch frames."""
import json
import os

import pytest

from conftest import ROOT

MLP_CALLS = ("sa_group_mlp_max", "sa_group_mlp_max_layer")


def _line(name="r04_bench_default.json"):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r04_bench_default.json", "r04_bench_20steps.json", "r04_bench_dup10.json", "r04_bench_dense.json",
                                  "r04_bench_rings64.json", "r04_cold_2.json", "r04_cold_3.json"])
def test_bench_line_has_the_contract_fields(name):
    d = _line(name)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "env_knobs", "verify", "timed_window_ms", "ramp_dominated"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    c = d["config"]
    assert c["executor"] == "staged" and "SAPipeline" in c["executor_note"] and c["pool_frames_per_gpu"] >= 128
    # the staged executor: three streams on ROCm's default four hardware queues, linear graphs, nothing set in the environment
    assert c["streams_used"] == 3 and c["hw_queues"] == 4 and c["linear_graphs"] is True and c["hip_graphs"] is True
    assert "GPU_MAX_HW_QUEUES" not in d["env_knobs"]
    # what decides a short run travels inside `config` (the driver's record keeps `config` and `roofline` verbatim)
    for k in ("timed_window_ms", "one_package_alone_ms", "ramp_dominated", "host_issue_ms_per_step", "steps_in_flight_mean",
              "priming", "sclk_mhz", "timed_packages_ms", "package_sizes"):
        assert k in c, k
    assert c["priming"]["wall_ms"] >= 150 and c["priming"]["packages"] >= 2 * c["slots"]     # independent of --warmup
    rows = c["timed_packages_ms"]["rows"]
    assert rows and sum(r[1] for r in rows) >= min(d["steps"], sum(r[1] for r in rows))
    for _slot, _fill, reached, a_done, done in rows:
        assert 0 <= reached <= a_done <= done <= c["timed_window_ms"] + 0.5
    assert d["config"]["frames_per_launch"] == d["config"]["frames_per_step_per_gpu"] * d["config"]["batches_per_replay"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # value = frames of all ranks / wall time of the timed steps
    frames = d["config"]["frames_per_step_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert abs(d["timed_window_ms"] - d["ms_per_step"] * d["steps"]) < 0.05 * d["timed_window_ms"]
    # every pipeline output of the verification pass equalled the eager result of the same batch
    v = d["verify"]
    assert v["all_equal_eager"] is True and v["output_sha1_replay"] == v["output_sha1_eager"] and v["batches"] >= 16
    assert v["batches_per_replay"] == d["config"]["batches_per_replay"]
    assert not [k for k in d["env_knobs"] if k.startswith("SA_")]


def test_default_line_reports_the_other_executor_beside_the_headline():
    for name in ("r04_bench_default.json", "r04_bench_20steps.json", "r04_cold_2.json"):
        d = _line(name)
        u = d["config"]["other_executor"]
        assert u["executor"]["executor"] == "slots" and u["hw_queues"] == 16 and u["unit"] == d["unit"] and u["steps"] == d["steps"]
        assert abs(u["value"] - d["config"]["frames_per_step_per_gpu"] / (u["ms_per_step"] * 1e-3)) / u["value"] < 1e-3
        # the 3-stream executor is at least the 16-slot one, in the short run and in the long one
        assert d["value"] >= 0.98 * u["value"]


def test_cold_first_command_runs_agree_with_the_warm_one():
    # `python bench.py --gpus 1 --steps 20 --warmup 5` as the first and only command of a fresh lease
    warm = _line("r04_bench_20steps.json")["value"]
    for name in ("r04_cold_2.json", "r04_cold_3.json"):
        d = _line(name)
        assert d["steps"] == 20 and d["warmup"] == 5 and abs(d["value"] - warm) / warm < 0.15


def test_default_line_cpu_baseline_and_ramp_flag():
    d = _line()
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "effective_parallelism"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    assert d["ramp_dominated"] is False and d["steps"] >= 100
    assert _line("r04_bench_20steps.json")["ramp_dominated"] is True      # the driver's 20-step run says so itself


def test_fps_roofline_is_latency_bound_with_evaluated_pairs():
    d = _line()
    r, fpl = d["roofline"], d["config"]["frames_per_launch"]
    assert r["bound"] == "latency" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert r["device_kernel"].startswith("fps3_wave_bucket_kernel")
    assert r["reference_pair_evaluations"] == fpl * 4095 * 16384
    assert 0 < r["evaluated_pairs"] < r["reference_pair_evaluations"] and abs(r["evaluated_frac"] - r["evaluated_pairs"] / r["reference_pair_evaluations"]) < 1e-4
    assert abs(r["achieved"] - r["reference_pair_evaluations"] * 11 / (r["avg_launch_ms"] * 1e9)) < 0.01
    assert abs(r["cycles_per_pick"] - r["us_per_pick"] * r["clock_mhz"]) < 1.0 and r["cus_used"] == fpl   # one CU per frame
    assert abs(r["algorithmic_bytes"] / fpl / 1e6 - 0.213) < 0.001   # 213 KB algorithmic per frame for layer-1 FPS
    assert r["traffic"] is not None and r["traffic"] < 2 * r["algorithmic_bytes"]      # counter traffic of the same launch shape


def test_algorithmic_mlp_work_matches_survey_8d():
    d = _line()
    frames = d["config"]["frames_per_launch"]              # the stages are the calls of one replay
    mlp = sum(s["gflop"] * s["calls_per_step"] for s in d["stages"] if s["kernel"] in MLP_CALLS + ("sa_dense", "sa_vote_tail"))
    assert abs(mlp / frames - 30.93) < 0.01            # 30.93 GFLOP per frame through the backbone
    fps = [s for s in d["stages"] if s["label"].startswith("fps n=16384")]
    assert len(fps) == 1 and fps[0]["calls_per_step"] == 1
    g = d["roofline_grouped_mlp"]
    rows = d["mlp_rows_per_step"]
    # frac is on EXECUTED flops, the nominal figure stands beside it
    ms = g["kernel_ms"] + g["plan_ms"]
    assert abs(g["achieved"] - rows["gflop_evaluated"] / ms) < 0.5 and abs(g["frac"] - g["achieved"] / g["peak"]) < 1e-4
    assert abs(g["nominal_tflops"] - rows["gflop_nominal"] / ms) < 1.0 and g["nominal_frac"] > g["frac"]
    assert rows["frames"] == frames
    # the north star's ">= 30 % MFMA utilisation on the grouped MLP": the hardware counter over the MLP kernels of the
    # committed PMC pass (same launch shape), not the flop model
    assert g["pmc"]["mfma_util"] >= 0.30 and "_traffic.json" in g["pmc"]["source"]


def test_data_variants_move_the_data_dependent_counters():
    base, dense, dup = _line(), _line("r04_bench_dense.json"), _line("r04_bench_dup10.json")
    rings = _line("r04_bench_rings64.json")
    assert rings["config"]["data"] == "rings64" and dense["value"] < rings["value"] < base["value"]
    assert base["mlp_rows_per_step"]["evaluated_frac"] < rings["mlp_rows_per_step"]["evaluated_frac"] < dense["mlp_rows_per_step"]["evaluated_frac"]
    assert dense["config"]["data"] == "dense" and dup["config"]["data"] == "dup10"
    assert dense["mlp_rows_per_step"]["evaluated_frac"] > 0.95 > 0.5 > base["mlp_rows_per_step"]["evaluated_frac"]
    assert dense["value"] < base["value"]               # the uniform box is the slow case, and the line shows it


# ---- round 6: the driver's one command carries the other claims as flat scalars ---------------------------------
R06_EXTRA_KEYS = ["steady512_frames_s", "rings64_frames_s", "detector_frames_s", "configs4_frames_s", "configs2_frames_s",
                  "group_b128_hbm_frac", "group_b32_hbm_frac", "group_b8_hbm_frac", "dense_frames_s", "rccl_smoke"]


@pytest.mark.parametrize("name", ["r06_bench_20steps_cold.json", "r06_bench_20steps.json", "r06_cold_1.json", "r06_cold_2.json", "r06_cold_3.json"])
def test_r06_headline_line_carries_the_secondary_measurements_in_its_first_twenty_config_keys(name):
    d = _line(name)
    c = d["config"]
    first = list(c)[:20]
    assert first[0] == "workload" and "configs[1]" in c["workload"] and d["steps"] == 20 and d["warmup"] == 5
    for k in R06_EXTRA_KEYS + ["cgroup_cpu_quota_cores", "cgroup_nr_throttled", "timed_window_ms", "probe_window_ms"]:
        assert k in first, (k, first)
    for k in R06_EXTRA_KEYS[:-1]:
        assert isinstance(c[k], (int, float)) and c[k] > 0, (k, c[k])
    assert c["rccl_smoke"] == "ok"                                            # backend "nccl" (RCCL) initialised and ran its collectives on the box
    assert d["extras"]["runs"]["rccl_smoke"]["gather_check"]["backend"] == "nccl (RCCL)"
    # the secondary figures are what the stand-alone runs of the same lease say (within run-to-run spread)
    assert c["steady512_frames_s"] > 1.2 * d["value"] and c["rings64_frames_s"] < c["steady512_frames_s"]
    assert 0.9 < c["detector_frames_s"] / c["steady512_frames_s"] <= 1.02    # points -> boxes costs ~1 % over the backbone
    assert c["group_b128_hbm_frac"] >= 0.40 > c["group_b8_hbm_frac"]         # the north star's 40 % of HBM: met where calls are large
    assert d["extras"]["seconds"] <= d["extras"]["budget_s"] + 15 and not d["extras"]["skipped"]
    # the contract fields are untouched
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["verify"]["all_equal_eager"] is True


def test_r06_detector_line():
    d = _line("r06_bench_detector.json")
    assert "points -> boxes" in d["metric"] and d["config"]["workload"].startswith("detector")
    v = d["verify"]
    assert v["all_equal_eager"] is True and set(v["tail_outputs_compared"]) >= {"pred_3d_bbox", "pred_3d_score", "nms_idx", "nms_cnt"}
    assert d["detections"]["max_output_num"] == 100
    names = [s["kernel"] for s in d["stages"]]
    assert names[-3:] == ["sa_decode_anchor_free", "sa_nms_bev", "sa_nms_gather"]           # the tail is C-ABI launches, nothing else
