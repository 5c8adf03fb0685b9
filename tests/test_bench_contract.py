"""bench.py's output contract, checked on the committed bench lines of rounds 4 and 5 (CPU only), and its loud failure
without a GPU."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        return json.loads([ln for ln in fh.read().splitlines() if ln.startswith("{")][-1])


def test_committed_bench_line_has_every_contract_field():
    d = _line("r04_bench.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "u64" and d["n_gpus"] == 1
    assert "Connect4(6,7,4)" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["rewards_to_host"] is True  # the metric ends with the rewards in a host array
    assert d["config"]["gather"] == "none" and d["config"]["loop"].startswith("native")
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    # the binding resource is VALU instruction issue (round-2 review): wave-instructions per second against the SIMD-32 peak
    assert roof["bound"] == "valu_issue" and roof["unit"] == "Ginstr/s" and roof["peak"] == pytest.approx(1228.8)
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.3 < roof["frac"] <= 1.0
    assert roof["kernel"] == "k_connect_rollout_opened" and roof["traffic"] > 1e7
    assert 0 < roof["hbm"]["frac"] < 0.2 and roof["pcie"]["frac"] <= 1.0 and roof["algorithmic"]["bytes_per_env_step"] == 32
    cpu = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "single_game_latency_us"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and "restatement" in cpu["sample"]
    assert cpu["kind_note"].startswith("restatement") and cpu["all_cores"]["cores"] >= cpu["cores"] and cpu["all_cores"]["value"] > 0
    assert cpu["parity_with_host_rewards"] is True
    # round 4: the chip's own busy counters beside the instruction-rate fraction, and one launch at a time beside `value`
    busy = roof["valu_busy"]
    assert busy["counters_file"].startswith("r04_") and 0.85 < busy["valu_busy_frac"] < 1.0 == (roof["valu_busy_frac"] > 0)
    assert abs(sum(busy["wave_time_split"].values()) - 1.0) < 0.02 and 0 < busy["frac_of_quad_cycles_with_two_valu_issued"] < 0.5
    assert roof["valu_busy_from_rate"] == pytest.approx(roof["achieved"] / 614.4)
    assert d["solo"]["rewards_to_host"] is False and 0.3 * d["value"] < d["solo"]["value"] < d["value"]
    assert d["value"] > 1e9  # the north star's floor was 1e9 env-steps/s on EIGHT GPUs
    assert d["value"] >= 0.9 * d["device_resident"]["value"]  # the hand-over costs less than 10 % of the device-resident rate
    assert len(d["values_of_3"]) == 3 and d["value"] == d["values_of_3"][0]
    assert min(d["values_of_3"]) <= d["value_median_of_3"] <= max(d["values_of_3"])
    for name, floor in (("connect_12x13x5", 1e11), ("bounce_default", 5e9)):   # BASELINE configs 3 and 4 in the same line
        o = d["other_configs"][name]
        assert o["parity_with_oracle"] is True and o["value"] > floor and o["solo"]["value"] > 0
        assert o["valu_issue"]["frac"] is not None and o["valu_issue"]["counters_file"].startswith("r04_")
        assert o["valu_busy"]["counters_file"].startswith("r04_")
    assert d["other_configs"]["bounce_default"]["value"] > 1.2e10   # round 3: 1.05e10
    assert d["grids_to_host"]["value"] > 5e9 and d["grids_to_host"]["host_grids_equal_device_grids"] is True


def test_round5_bench_line():
    """The round-5 line (`python bench.py`, 200 steps): the contract's fields, the new RNG contract named in the workload,
    counters gated per kernel unit, the CPU baseline as the best of a few thread teams, and the round's targets."""
    d = _line("r05_bench.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["n_gpus"] == 1 and d["config"]["rewards_to_host"] is True and "model" not in d["config"]
    assert set(d["config"]["kernel_unit_ids"]) == {"connect", "bounce", "generic"}
    roof = d["roofline"]
    assert roof["bound"] == "valu_issue" and roof["peak"] == pytest.approx(1228.8) and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["counters_file"].startswith("r05_") and "connect kernel unit" in roof["basis"]
    assert roof["wave_instr_per_launch"] < 19.5e6          # round 4: 20.68 M, round 1: 27.8 M
    assert roof["valu_busy"]["counters_file"].startswith("r05_") and 0.8 < roof["valu_busy_frac"] < 1.0
    assert d["value"] > 7.9e11 and d["value_median_of_3"] > 8.0e11   # round 4: 7.56e11; the review asked for >= 8.1e11 at 200 steps
    assert d["value"] >= 0.9 * d["device_resident"]["value"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["parity_with_host_rewards"] is True
    assert str(cpu["cores"]) in cpu["by_threads"] and cpu["value"] == max(cpu["by_threads"].values())   # the best team carries `value`
    assert cpu["gpu_share_of_host"]["cores"] >= 1 and cpu["single_thread_value"] > 0 and "first touched" in cpu["sample"]
    bounce = d["other_configs"]["bounce_default"]
    assert bounce["parity_with_oracle"] is True and bounce["value"] > 1.65e10     # round 4: 1.53e10; review target 1.75e10
    assert bounce["valu_issue"]["wave_instr_per_launch"] <= 2.85e8 and bounce["valu_issue"]["counters_file"].startswith("r05_")
    big = d["other_configs"]["connect_12x13x5"]
    assert big["parity_with_oracle"] is True and big["value"] > 2.3e11


def test_round5_short_run_and_api_levels():
    """The driver's 20-step run against the 200-step one, and what each API level delivers: the documented Python loop
    (RolloutPipeline.run) within 5 % of the native executor on configs 2 and 4."""
    long, short = _line("r05_bench.json"), _line("r05_bench_steps20.json")
    assert short["steps"] == 20 and short["value_median_of_3"] > 0.86 * long["value_median_of_3"]
    assert short["value_median_of_3"] > 7.2e11     # round 4: 6.95e11; the review asked for >= 7.4e11
    with open(os.path.join(ROOT, "profiles", "r05_api_rates.json")) as fh:
        api = json.load(fh)["configs"]
    for name in ("connect_6x7x4", "bounce_default"):
        assert api[name]["pipeline_over_executor"] > 0.95, name
        assert api[name]["naive"]["value"] < api[name]["pipeline"]["value"]
    assert api["connect_12x13x5"]["pipeline_over_executor"] > 0.93


def test_round5_rehearsals_name_the_communicators_rank_count():
    """BASELINE config 5 rehearsed at 2, 4 and 6 processes sharing the GPU (full shards, both hand-overs verified): the line
    says how many ranks the COMMUNICATOR counts; the full-size 8-rank gather ran as 4 processes x 2 ranks."""
    for n in (2, 4, 6):
        d = _line(f"r05_dist_{n}_ranks_standin.json")
        assert d["n_gpus"] == n and d["config"]["global_batch"] == n << 20 and d["config"]["gathers_measured"] == ["shm", "rccl"]
        assert d["gather_shm"]["gathered_rewards_verified"] is True and d["gather_rccl"]["gathered_rewards_verified"] is True
        info = d["gather_rccl"]["gather_info"]
        assert info["ranks"] == n and info["rank"] == 0 and info["transport_check"] == "passed" and "libfake_rccl" in info["transport"]
    with open(os.path.join(ROOT, "profiles", "r05_gather_8_ranks_full_size_standin.txt")) as fh:
        text = fh.read()
    for r in range(8):
        assert f"FULL_OK rank {r} of 8 verified 14 steps of 8 x 1048576 games" in text
    assert text.count("'ranks': 8") == 8


def test_short_run_stays_close_to_the_long_one():
    """The driver times 20 steps: pre-warm, the native loop and the polled last delivery keep that figure within 20 % of
    the 200-step one (round 2: 13 % below at 5.7e11, with outliers to -40 %), and its three regions within 8 % of each other."""
    long, short = _line("r04_bench.json"), _line("r04_bench_steps20.json")
    # (the median of the three regions: on the shared test hosts any single 0.7 ms region can be hit by a neighbour)
    assert short["steps"] == 20 and short["value_median_of_3"] > 0.88 * long["value_median_of_3"]
    assert short["value"] > 0.8 * long["value"]


def test_counters_file_names_its_build():
    with open(os.path.join(ROOT, "profiles", "r04_rollout_counters.json")) as fh:
        c = json.load(fh)
    assert len(c["build_id"]) == 16 and c["valu_wave_instructions_per_launch"] > 1e6
    assert 2.0 < c["mix_cycles_per_instruction"] < 6.0


def test_counters_are_gated_per_kernel_unit(tmp_path):
    """Round-4 review, weak #8: counters are quoted when the id of the KERNEL UNIT they were taken on matches the running
    library's (bgs_kernel_unit_id), not the global build id -- a Bounce-only edit leaves the Connect counters quotable
    and the other way round; files from before round 5 (no unit id) are still matched on the global build id."""
    sys.path[:0] = [ROOT]
    import bench

    def write(name, **fields):
        with open(tmp_path / name, "w") as fh:
            json.dump(fields, fh)

    write("r05_rollout_counters.json", build_id="b" * 16, unit="connect", unit_id="c" * 16, valu_wave_instructions_per_launch=2.0e7,
          hbm_bytes_per_launch=2.0e7)
    write("r05_bounce.json", build_id="b" * 16, unit="bounce", unit_id="d" * 16, valu_wave_instructions_per_launch=3.0e8,
          kernels={"k_bounce_rollout_pieces<16, 256>(...)": {"SQ_INSTS_VALU": 2.9e8}})
    write("r04_k2c.json", build_id="a" * 16, kernels={"void k_connect_rollout_lds<...>": {"SQ_INSTS_VALU": 2.8e7}})
    write("r05_valu_busy.json", build_id="b" * 16, cases={
        "k2o_3deep": {"what": "x", "unit": "connect", "unit_id": "c" * 16, "kernels": {"k": {"derived": {"valu_busy_frac": 0.9}}}},
        "k3p_8x": {"what": "y", "unit": "bounce", "unit_id": "d" * 16, "kernels": {"k": {"derived": {"valu_busy_frac": 0.6}}}}})
    ids = {"build": "z" * 16, "units": {"connect": "c" * 16, "bounce": "d" * 16, "generic": "e" * 16}}
    got, why = bench.committed_counters(ids, "rollout_counters", profiles_dir=str(tmp_path))
    assert why is None and got["valu_wave_instructions_per_launch"] == 2.0e7 and got["file"] == "r05_rollout_counters.json"
    assert bench.committed_counters(ids, "bounce", "k_bounce_rollout", profiles_dir=str(tmp_path))[0]["unit_id"] == "d" * 16
    assert bench.busy_block(ids, "k2o_3deep", profiles_dir=str(tmp_path))["valu_busy_frac"] == 0.9
    # a Bounce-only edit: the Bounce unit's id (and the global build id) move, the Connect unit's does not
    edited = {"build": "y" * 16, "units": dict(ids["units"], bounce="f" * 16)}
    assert bench.committed_counters(edited, "rollout_counters", profiles_dir=str(tmp_path))[0] is not None
    assert bench.busy_block(edited, "k2o_3deep", profiles_dir=str(tmp_path)) is not None
    got, why = bench.committed_counters(edited, "bounce", "k_bounce_rollout", profiles_dir=str(tmp_path))
    assert got is None and "bounce unit" in why and "not quoted" in why
    assert bench.busy_block(edited, "k3p_8x", profiles_dir=str(tmp_path)) is None
    # ... and a Connect edit the other way round
    edited = {"build": "x" * 16, "units": dict(ids["units"], connect="0" * 16)}
    assert bench.committed_counters(edited, "rollout_counters", profiles_dir=str(tmp_path))[0] is None
    assert bench.committed_counters(edited, "bounce", "k_bounce_rollout", profiles_dir=str(tmp_path))[0] is not None
    # a file from before round 5 names the global build only
    assert bench.committed_counters(ids, "k2c", "_lds", profiles_dir=str(tmp_path))[0] is None
    old = dict(ids, build="a" * 16)
    assert bench.committed_counters(old, "k2c", "_lds", profiles_dir=str(tmp_path))[0]["valu_wave_instructions_per_launch"] == 2.8e7


def test_sharded_loops_with_one_rank_over_rccl_are_within_reach_of_the_plain_loop():
    """The N > 1 code paths over the real RCCL with a one-rank world (all a one-GPU box can run): the shared host array
    costs nothing; the in-library RCCL gather -- a communicator, a communication thread and a second stream between the
    rollout and the sink, with nothing to transport for one rank -- stays within 35 % of it (0.7-0.93 from box to box)."""
    plain, shm, rccl = _line("r03_bench.json"), _line("r03_dist_shm.json"), _line("r03_dist_rccl.json")
    assert shm["config"]["gather"] == "shm" and rccl["config"]["gather"] == "rccl"
    assert shm["config"]["gathered_rewards_verified"] is True and rccl["config"]["gathered_rewards_verified"] is True
    typical = lambda line: line["value_median_of_3"]   # (one region in three may be hit by a neighbour on the host)
    assert typical(shm) > 0.95 * typical(plain) and typical(rccl) > 0.65 * typical(shm)


def test_round4_gather_machinery_is_off_the_step():
    """Round 4: one run measures BOTH hand-overs of the N > 1 path (`--gather both`); over the real RCCL library with a
    one-rank world the in-library gather -- whose machinery is now per group of steps, and absent when there is no peer --
    stays within 5 % of the shared host array (round 3: 0.66-0.88), at 200 and at 20 steps; with three ranks sharing the GPU
    over the tests' stand-in transport both hand-overs verify every rank's rows."""
    typical = lambda values: sorted(values)[1]
    for name in ("r04_dist_one_rank.json", "r04_dist_one_rank_steps20.json"):
        d = _line(name)
        assert d["config"]["gathers_measured"] == ["shm", "rccl"] and d["config"]["gather"] == "shm"
        shm, rccl = d["gather_shm"], d["gather_rccl"]
        assert shm["gathered_rewards_verified"] is True and rccl["gathered_rewards_verified"] is True
        assert rccl["gather_info"]["transport"].startswith("librccl") and rccl["gather_info"]["transport_check"] == "none"
        assert typical(rccl["values_of_3"]) > 0.95 * typical(shm["values_of_3"]), name
    d = _line("r04_dist_three_ranks_standin.json")
    assert d["n_gpus"] == 3 and d["gather_shm"]["gathered_rewards_verified"] is True
    rccl = d["gather_rccl"]
    assert rccl["gathered_rewards_verified"] is True and "libfake_rccl" in rccl["gather_info"]["transport"]
    assert rccl["gather_info"]["transport_check"] == "passed" and rccl["gather_info"]["direct"] is False


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


def test_round6_lines():
    """Round 6: the line names its RNG contract and carries the other contract's rate beside `value` (the strict contract
    costs < 10 %), Bounce's compile-time geometry shows in BASELINE config 4 (pipelined > 2.1e10, alone > 4.8e9, both queue
    settings of eight in flight, <= 2.4e8 instructions a launch), the counters are this round's own."""
    long, short, strict = _line("r06_bench.json"), _line("r06_bench_steps20.json"), _line("r06_bench_per_ply.json")
    assert long["steps"] == 200 and short["steps"] == 20 and long["value_median_of_3"] > 8.0e11 and short["value_median_of_3"] > 7.3e11
    assert long["config"]["rng"].startswith("per-block") and strict["config"]["rng"].startswith("per-ply")
    for d in (long, short, strict):
        other = d["rng_other"]
        assert other["parity_with_oracle_first_4096"] is True and other["rewards_to_host"] is True
        assert d["cpu_baseline"]["parity_with_host_rewards"] is True   # (the oracle under the line's own contract)
    assert long["rng_other"]["rng"].startswith("per-ply") and 0.88 < long["rng_other"]["over_value"] < 1.0
    assert strict["rng_other"]["rng"].startswith("per-block") and 1.0 < strict["rng_other"]["over_value"] < 1.15
    assert long["rccl_ranks"] is None and long["ms_per_step_fastest_rank"] == pytest.approx(long["ms_per_step"])
    roof = long["roofline"]
    assert roof["counters_file"] == "r06_rollout_counters.json" and 0.55 < roof["frac"] < 0.62 and roof["valu_busy"]["counters_file"] == "r06_valu_busy.json"
    bounce = long["other_configs"]["bounce_default"]
    assert bounce["parity_with_oracle"] is True and bounce["value"] > 2.1e10 and bounce["solo"]["value"] > 4.8e9
    assert bounce["valu_issue"]["counters_file"] == "r06_bounce.json" and bounce["valu_issue"]["wave_instr_per_launch"] <= 2.4e8
    eight = bounce["eight_in_flight"]
    assert eight["hardware_queues_16"]["value"] > 1.8e10 > eight["hardware_queues_4"]["value"] > 1.0e10
    assert eight["hardware_queues_4"]["parity_with_oracle"] is True and eight["hardware_queues_16"]["parity_with_oracle"] is True
    assert long["other_configs"]["connect_12x13x5"]["value"] > 2.3e11


def test_round6_rehearsals_value_is_the_rccl_gathers():
    """BASELINE config 5 rehearsed at 2, 4 and 6 processes sharing the GPU over the stand-in: since round 6 `value` is the RCCL
    gather's (the north-star's collective), `rccl_ranks` = what the communicator counts, both hand-overs verified, nothing failed."""
    for n in (2, 4, 6):
        d = _line(f"r06_dist_{n}_ranks_standin.json")
        assert d["n_gpus"] == n and d["config"]["global_batch"] == n << 20 and d["config"]["gather"] == "rccl" and d["rccl_ranks"] == n
        assert d["value"] == pytest.approx(d["gather_rccl"]["value"]) and "failed_handovers" not in d
        assert d["gather_shm"]["gathered_rewards_verified"] is True and d["gather_rccl"]["gathered_rewards_verified"] is True
        assert 0 < d["ms_per_step_fastest_rank"] <= d["ms_per_step_slowest_rank"]
