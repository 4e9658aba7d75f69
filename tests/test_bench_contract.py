"""The bench line committed under profiles/ (the output of `python bench.py` on the GPU box) carries every field of
the driver's contract, and its algorithmic work model reproduces SURVEY.md 8d: 30.93 GFLOP of MLP per frame."""
import json
import os

from conftest import ROOT


def _line():
    with open(os.path.join(ROOT, "profiles", "r01_bench_s16_graphs.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_bench_line_has_the_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    # value = frames of all ranks / wall time of the timed steps
    frames = d["config"]["frames_per_step_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def test_algorithmic_mlp_work_matches_survey_8d():
    d = _line()
    frames = d["config"]["frames_per_step_per_gpu"]
    mlp = sum(s["gflop"] * s["calls_per_step"] for s in d["stages"] if s["kernel"] in ("sa_group_mlp_max", "sa_dense"))
    assert abs(mlp / frames - 30.93) < 0.01            # 30.93 GFLOP per frame through the backbone
    fps = [s for s in d["stages"] if s["label"].startswith("fps n=16384")]
    assert len(fps) == 1 and fps[0]["calls_per_step"] == 1
    assert abs(fps[0]["mbytes"] / frames - 0.213) < 0.001   # 213 KB algorithmic per frame for layer-1 FPS
