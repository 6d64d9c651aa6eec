"""The driver's contract for bench.py: ONE JSON line with the agreed keys, the metric string of BASELINE.json, the roofline
and cpu_baseline objects, bit-exact parity.  Run small here; the driver runs the default sizes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--reads", "60000", "--global-tasks", "20000", "--sw-tasks", "20000", "--seed-reads", "0",
                        "--cpu-seconds", "1"], capture_output=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    if os.path.exists(os.path.join(ROOT, "BASELINE.json")):
        assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    # value = reads per second over exactly the timed steps (reads without any extension task -- ~2 % -- are not counted)
    per_step = d["config"]["reads_per_gpu"] / (d["ms_per_step"] * 1e-3)
    assert 0.9 * per_step < d["value"] <= per_step * (1 + 1e-9)
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["parity"].startswith("bit-exact")
    assert d["global_alignment"]["parity"].startswith("bit-exact") and d["mate_rescue_sw"]["parity"].startswith("bit-exact")
