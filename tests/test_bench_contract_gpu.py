"""The driver's contract for bench.py: ONE JSON line with the agreed keys, the metric string of BASELINE.json, the roofline
and cpu_baseline objects, bit-exact parity.  Run small here; the driver runs the default sizes (10 M pairs per step)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def _contract(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    if os.path.exists(os.path.join(ROOT, "BASELINE.json")):
        assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
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


def test_bench_prints_one_contract_line_on_the_metric_config():
    """configs[2] shape, whole DP path in the step (fused seed extension -> global -> rescue), scaled down."""
    d = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--pairs", "60000", "--chunk-reads", "40000", "--cpu-seconds", "1",
              "--pipeline-reads", "20000", "--pipeline-genome", "500000"])
    _contract(d, 3, 1)
    assert "configs[2]" in d["config"]["workload"] and d["config"]["chunks"] == 3
    # value = reads per second over exactly the timed steps; a pair counts as two reads
    per_step = d["config"]["reads_per_gpu"] / (d["ms_per_step"] * 1e-3)
    assert d["config"]["reads_per_gpu"] == 120000 and abs(d["value"] - per_step) <= 1e-6 * per_step
    st = d["stages_ms_per_step"]
    assert all(st[k] > 0 for k in ("seed_extension", "global_alignment", "mate_rescue_sw"))
    ks = d["roofline"]["kernels"]
    assert any("round R1" in k["kernel"] for k in ks) and any("global_lane_kernel<64" in k["kernel"] for k in ks)
    assert any(k["kernel"].startswith("extend_lane_kernel<") and k["ms"] > 0 for k in ks)  # the extension kernels themselves, timed per bin
    # the dominant kernel is the single kernel with the largest total time, whichever family it is in
    singles = [k for k in ks if k["single_kernel"]]
    assert d["roofline"]["kernel"] == max(singles, key=lambda k: k["ms"])["kernel"]
    assert d["config"]["ksw_extend2_calls_per_gpu"] > d["config"]["seeded_reads_per_gpu"]  # left AND right flanks
    pb = d["cpu_baseline_pipeline"]
    if "skipped" not in pb:  # needs oracle/_ref (travels with the tree)
        assert pb["kind"] == "reference" and pb["value"] > 0 and pb["dut_value"] > 0 and pb["sam_identical"] is True
        assert pb["dut_detail"]["chunks"] and "phase 2 (marking, pairing, global alignments, SAM)" in pb["dut_detail"]["chunks"][0]


def test_bench_round1_line_still_runs():
    d = _run(["--workload", "se1m", "--gpus", "1", "--steps", "3", "--warmup", "1", "--reads", "60000", "--global-tasks", "20000",
              "--sw-tasks", "20000", "--seed-reads", "0", "--cpu-seconds", "1"])
    _contract(d, 3, 1)
    assert d["config"]["workload"].startswith("se1m")
    assert d["global_alignment"]["parity"].startswith("bit-exact") and d["mate_rescue_sw"]["parity"].startswith("bit-exact")


def test_bench_two_ranks_on_one_device_and_the_host_fed_step():
    """`bench.py --gpus 2` must start by itself (no RANK in the environment): two rank processes, here both mapped onto the one visible
    device (--oversubscribe: gloo for the barrier and the report), static shard per rank, one line with n_gpus 2 and one time per rank;
    and the host-fed steps (uploads one chunk ahead on a copy stream, results downloaded, inside the timed region) report `value_streamed`
    with the same results as the resident steps (the line's parity covers host-fed == resident)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "2", "--warmup", "1", "--pairs", "60000",
                        "--chunk-reads", "40000", "--no-cpu-baseline", "--no-pipeline-baseline"], capture_output=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert d["parity"].startswith("bit-exact") and d["value"] > 0
    # weak scaling: value counts both ranks' reads over the slower rank's time
    assert abs(d["value"] - 2 * d["config"]["reads_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    st = d["streamed"]
    assert d["value_streamed"] == st["value_streamed"] > 0 and st["h2d_bytes_per_step"] > 0 and st["d2h_bytes_per_step"] > 0
    assert st["h2d_GBps_copy_stream"] and st["h2d_GBps_copy_stream"] > 0.1
    assert d["value_streamed"] <= d["value"] * 1.25  # the fed step cannot be meaningfully faster than the resident one
