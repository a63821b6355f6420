"""GPU: the benchmark entry itself.  `python bench.py --gpus 2` must start its own two ranks (the
driver's scaling command has exactly that form), run the data-parallel step and print ONE JSON line;
rehearsed here with gloo carrying the collective and both ranks on the one visible MI355X (the
multi-GPU node runs the same code over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + "\n---\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"], {"TDX_DIST_BACKEND": "gloo"})
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 512 and res["scaling"] == "weak"
    assert res["value"] > 0 and res["steps"] == 2 and res["warmup"] == 1
    assert abs(res["images_per_s_per_gpu"] * 2 - res["value"]) <= 0.2
    assert res["weak_scaling"]["single_rank_images_per_s"] > 0
    assert res["parity_check"]["eps_mse_vs_oracle"] < 1e-5   # first step checked against the oracle at B=256


def test_bench_single_rank_line_shape():
    res = _run(["--steps", "3", "--warmup", "1", "--no-extras"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["dtype"] == "f32" and res["vs_baseline"] is None
    assert res["parity_check"]["batch"] == 256 and res["parity_check"]["eps_rel_mse_vs_oracle"] < 1e-9


def test_bench_two_rank_line_carries_sample_latency_and_per_rank_roofline():
    """The N > 1 line in full (BASELINE's metric names the 1000-step sample latency at every N): each rank runs
    its own replica chains at n = 16 / 64 - no collective in the path, MAX over ranks reported - and the
    kernel-level roofline is taken on every rank's GPU.  (Both ranks share the one visible GPU here, so the
    figures themselves mean nothing; the shape of the line and the absence of a hang do.)"""
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"TDX_DIST_BACKEND": "gloo"}, timeout=900)
    assert res["n_gpus"] == 2
    s = res["sample"]
    assert 0 < s["n16_fastest_rank"] <= s["n16"] and 0 < s["n64_fastest_rank"] <= s["n64"]
    assert "no collective" in s["unit"]
    roof = res["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1
    assert [r["rank"] for r in roof["per_rank"]] == [0, 1]
    assert all(0 < r["frac"] < 1 for r in roof["per_rank"])
