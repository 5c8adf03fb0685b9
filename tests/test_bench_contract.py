"""bench.py's output contract, checked on the committed round-2 bench line (CPU only), and its loud failure
without a GPU."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    with open(os.path.join(ROOT, "profiles", "r02_bench.json")) as fh:
        line = [ln for ln in fh.read().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "u64" and d["n_gpus"] == 1
    assert "Connect4(6,7,4)" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["rewards_to_host"] is True  # the metric ends with the rewards in a host array
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0 < roof["frac"] <= 1.0  # a physical fraction: bytes the kernel really moves over its duration
    assert roof["pcie"]["frac"] <= 1.0
    cpu = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1
    assert cpu["parity_with_host_rewards"] is True
    assert d["value"] > 1e9  # the north star's floor was 1e9 env-steps/s on EIGHT GPUs
    assert d["device_resident"]["value"] >= d["value"] * 0.9
    assert d["value"] >= 0.8 * d["device_resident"]["value"]  # hand-over within 20 % of the device-resident rate


def test_counters_file_names_its_build():
    with open(os.path.join(ROOT, "profiles", "r02_rollout_counters.json")) as fh:
        c = json.load(fh)
    assert len(c["build_id"]) == 16 and c["valu_wave_instructions_per_launch"] > 1e6
    assert 2.0 < c["mix_cycles_per_instruction"] < 6.0


def test_bench_starts_ranks_itself_and_relays_their_failure():
    """`python bench.py --gpus 2` with no launcher: two child ranks are started (before the parent touches any GPU API);
    without a GPU each of them refuses, and the parent relays that as its own exit code and prints no line."""
    sys.path[:0] = [os.path.join(ROOT, "board-game-simulator-python_amd")]
    from simulator.game import _abi

    if _abi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode == 2 and not proc.stdout.strip()
    assert proc.stderr.count("no GPU visible") == 2 and "rank(s) failed" in proc.stderr


def test_bench_refuses_to_run_without_a_gpu():
    sys.path[:0] = [os.path.join(ROOT, "board-game-simulator-python_amd")]
    from simulator.game import _abi

    if _abi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 2
    assert "no GPU" in proc.stderr and not proc.stdout.strip()
