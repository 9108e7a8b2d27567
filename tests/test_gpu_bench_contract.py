"""`bench.py` prints ONE JSON line with the driver's contract fields (a small, fast configuration of the same code path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_carries_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-slam",
                          "--no-cpu-baseline", "--gaussians", "200000", "--intrinsics", "fr3_office"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE line on stdout; progress goes to stderr
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "Mpix/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["workload"].startswith("custom")      # not the C5 configuration
    assert d["value"] > 0 and abs(d["value"] - 640 * 480 / 1e6 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert r["traffic"] is None                          # the stamped PMC files belong to the C5 workload only
    # the line says what ran BEFORE its timed region, without a look into `config`: the exact first step, the settle steps
    # (default 60, the first 20 of them timed as the cold figure) and the warm-up steps
    assert d["warmup_effective"] == 1 + d["config"]["settle_steps"] + (d["warmup"] - 1) == 63
    assert d["cold_steps"] == 20 and d["value_cold"] > 0
    assert abs(d["value_cold"] - 640 * 480 / 1e6 / (d["ms_per_step_cold"] * 1e-3)) < 0.01 * d["value_cold"]
    assert d["blend_in_timed_region"]["launches_timed"] <= 8       # the probes inside the timed region stay few
