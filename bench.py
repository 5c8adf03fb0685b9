#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched random rollout, Connect4(6,7,4), 2^20 boards per GPU.

One "step" = one pass of the hot path over one batch: every board of the batch is played from
Config.sample_initial_state() to its terminal state with uniformly sampled actions (enumerate -> sample ->
transition -> k-in-a-row / draw -> reward), fused in one HIP launch (k_connect_rollout).  env-steps are the
transitions applied to running boards (masked no-ops are not counted); they are counted on the device.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, rank r owns global game ids [r * 2^20, (r+1) * 2^20) (RNG streams are keyed by global
game id, so the shards reproduce the unsharded run); the only collective per step is the reward gather (RCCL
all-gather of 2-bit outcome codes, 256 KiB per rank, expanded to int8[N * 2^20, 2] rewards on rank 0) plus one
all-reduce of the step counters after the timed region.  Weak scaling.

Steps run on `--inflight` (default 2) batches / HIP streams in rotation: a rollout is bound by VALU instruction
issue, and its drain (the last game of every lane) leaves SIMDs idle that the next launch fills.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic HBM bytes (32 B per env-step: both 8-byte planes in and out) over the
                 rollout kernel's mean launch duration (HIP events on the launch stream) against 8 TB/s, plus
                 `valu_issue`, the resource that actually binds;
  cpu_baseline : the CPU oracle (plain C restatement, OpenMP) timed on this host on a bounded sample of the same
                 workload -- a reported baseline, not the target.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 0x0123456789ABCDEF
HEIGHT, WIDTH, COUNT = 6, 7, 4
BATCH_PER_GPU = 1 << 20
BYTES_PER_STEP = 32          # 2 planes x 8 B read + 2 planes x 8 B written per env-step
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
EVENT_STRIDE = 10            # HIP-event pairs bracket every 10th rollout launch of the timed region


def cpu_baseline(torch, last_seed, device_reward_head):
    """Time the oracle on this host's cores on a bounded sample of the same workload, and use the same run to
    cross-check the device's rewards for the first games of the last timed step."""
    import numpy as np

    from oracle import oracle

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("BGS_CPU_THREADS", "16")))  # a 1-GPU box's CPU share is 16 cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    n = 1 << 20
    reps = 4
    orc = oracle.ConnectOracle(HEIGHT, WIDTH, COUNT, n)
    orc.rollout(SEED, max_plies=4)  # touch the pages, start the thread team
    total, elapsed = 0, 0.0
    parity = None
    for r in range(reps):
        seed = last_seed if r == 0 else SEED + 1000 + r
        orc.reset()
        t0 = time.perf_counter()
        total += orc.rollout(seed)
        elapsed += time.perf_counter() - t0
        if r == 0 and device_reward_head is not None:
            parity = bool(np.array_equal(orc.reward[: device_reward_head.shape[0]], device_reward_head))
    # the same oracle on ONE thread (the reference's own loop is single-threaded under the GIL)
    single = None
    try:
        import ctypes

        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        small = oracle.ConnectOracle(HEIGHT, WIDTH, COUNT, 1 << 18)
        t0 = time.perf_counter()
        steps1 = small.rollout(SEED + 77)
        single = steps1 / (time.perf_counter() - t0)
        gomp.omp_set_num_threads(cores)
    except OSError:
        pass
    return {
        "value": total / elapsed,
        "unit": "env-steps/s",
        "cores": cores,
        "single_thread_value": single,
        "kind": "port",
        "sample": f"{reps} x 2^20 Connect4(6,7,4) games from the initial state ({total} env-steps), CPU oracle "
        f"(oracle/bgs_oracle.c, OpenMP, {cores} threads); the reference's own core is not buildable offline",
        "parity_with_device_rewards": parity,
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="boards per GPU (default 2^20)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight per GPU: step i runs on batch i %% D / HIP stream i %% D, so the drain of one "
                    "rollout (and, with N > 1, its reward gather) overlaps the start of the next (default 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-compare", action="store_true",
                    help="after the timed region, also time the same launches one at a time on one stream and report it "
                    "as `one_launch_at_a_time` (off by default so that a profile of this command sees only the timed "
                    "region's launches)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs the torch.distributed.run launcher (see docstring)", file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the rollout engine has no CPU fallback", file=sys.stderr)
        return 2

    # one rank per GPU; BGS_DIST_BACKEND=gloo is a rehearsal mode (ranks may then share a GPU, rewards are gathered
    # through host copies) used to exercise the N > 1 code path where RCCL cannot run (e.g. a one-GPU box)
    backend = os.environ.get("BGS_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from simulator.batch import ConnectBatch, expand_outcomes
    from simulator.sharding import gather_outcomes, shard_range, sum_steps

    n = args.batch
    if world > 1 and n % 4:
        print("bench.py: --batch must be a multiple of 4 when sharded (4 outcome codes per byte)", file=sys.stderr)
        return 2
    depth = max(1, args.inflight)
    os.environ.setdefault("BGS_ROLLOUT_WPS", "2")  # waves per SIMD per launch; `depth` launches share the chip
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(depth)] if depth > 1 else [torch.cuda.current_stream()]
    # per in-flight batch: the batch, its packed outcome codes (what crosses xGMI: 0.25 B per game), the gathered codes
    # of all ranks and -- on rank 0, the owner of "the one array" -- the expanded rewards int8[world * n, 2]
    batches, packed, all_packed, gathered = [], [], [], []
    gpu = torch.device("cuda", local_rank)
    for s in streams:
        with torch.cuda.stream(s):
            b = ConnectBatch(HEIGHT, WIDTH, COUNT, n, device=local_rank, use_torch=True)  # binds to stream s
            b.set_first_game(shard_range(n * world, rank, world)[0])
            batches.append(b)
            packed.append(torch.empty((n + 3) // 4, dtype=torch.uint8, device=gpu) if world > 1 else None)
            all_packed.append(torch.empty(world * ((n + 3) // 4), dtype=torch.uint8, device=gpu) if world > 1 else None)
            gathered.append(torch.empty((world * n, 2), dtype=torch.int8, device=gpu) if world > 1 and rank == 0 else None)
    device = gpu if backend == "nccl" else torch.device("cpu")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    pending = [None] * depth  # the in-flight reward gather of each batch slot

    def finish_gather(k):
        """Slot k's gather is due: make stream k wait for it and, on rank 0, expand the codes into THE reward array."""
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
            if rank == 0:
                expand_outcomes(all_packed[k], world * n, gathered[k])

    def one_step(i, ev=None):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            if dist is not None:
                finish_gather(k)  # of step i - depth: it has had `depth` rollouts of time to cross xGMI
            if ev is not None:
                ev[0].record(streams[k])
            batches[k].rollout(SEED + i, from_initial=True)
            if ev is not None:
                ev[1].record(streams[k])
            if dist is not None:
                # the path's only exchange: every rank's outcomes into one reward array on rank 0 (RCCL over xGMI).
                # Ranks ship 2-bit outcome codes (a reward pair is a function of the code); the collective is
                # asynchronous, so the stream goes straight on to its next rollout.
                batches[k].outcomes_tensor(packed[k])
                if backend == "nccl":
                    _, pending[k] = gather_outcomes(dist, packed[k], all_packed[k], async_op=True)
                else:
                    all_packed[k].copy_(gather_outcomes(dist, packed[k].cpu()))
                    if rank == 0:
                        expand_outcomes(all_packed[k], world * n, gathered[k])

    def drain():
        for k in range(depth):
            with torch.cuda.stream(streams[k]):
                finish_gather(k)

    for i in range(args.warmup):
        one_step(i)
    drain()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for b in batches:
        b.reset_steps()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # event pairs around every EVENT_STRIDE-th launch: each record is a marker packet on the stream, and
        # bracketing every launch would cost more than it measures
        one_step(args.warmup + i, events[i] if i % EVENT_STRIDE == 0 else None)
    drain()  # every step's rewards are in rank 0's array before the clock stops
    barrier()
    elapsed = time.perf_counter() - t0

    steps_local = sum(b.steps for b in batches)
    timed = [events[i] for i in range(args.steps) if i % EVENT_STRIDE == 0]
    kernel_ms = sum(s.elapsed_time(e) for s, e in timed) / max(len(timed), 1)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        steps_total = sum_steps(dist, steps_local, device)
    else:
        steps_total = steps_local

    gather_ok = None
    if dist is not None:
        # the gathered array must hold every rank's rewards in global game order: rank 0 re-plays the first games of
        # the LAST rank's shard on its own GPU (RNG streams are keyed by global game id) and compares
        last = args.warmup + args.steps - 1
        if rank == 0:
            probe = ConnectBatch(HEIGHT, WIDTH, COUNT, 4096, device=local_rank, use_torch=True)
            probe.set_first_game((world - 1) * n)
            probe.rollout(SEED + last, from_initial=True)
            want = probe.reward
            got = gathered[last % depth][(world - 1) * n : (world - 1) * n + 4096].cpu().numpy()
            gather_ok = bool((want == got).all())

    if rank == 0:
        value = steps_total / elapsed
        steps_per_launch = steps_local / max(args.steps, 1)
        achieved = steps_per_launch * BYTES_PER_STEP / (kernel_ms * 1e-3) / 1e9
        traffic, valu = None, None
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            with open(traffic_file) as fh:
                counters = json.load(fh).get("k_connect_rollout", {})
            traffic = counters.get("hbm_bytes_per_launch")
            if counters.get("valu_wave_instructions_per_launch") and n == BATCH_PER_GPU:
                # the resource that actually binds: one wave64 VALU instruction per 4 cycles per SIMD (measured,
                # tools/ubench.hip + SQ_ACTIVE_INST_VALU); instruction count per launch from the committed PMC pass
                instr = counters["valu_wave_instructions_per_launch"]
                peak = 256 * 4 * 2.4e9 / 4 / 1e9
                rate = instr / (elapsed / max(args.steps, 1)) / 1e9
                valu = {"wave_instr_per_launch": instr, "achieved_Ginstr_per_s": rate, "peak_Ginstr_per_s": peak,
                        "frac": rate / peak, "basis": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction; "
                        "launches overlap, so the rate uses ms_per_step"}
        out = {
            "metric": "env-steps/sec, Connect4(6,7,4) random rollout, batch="
            + ("2^20" if n == BATCH_PER_GPU else str(n)) + " per GPU",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"Connect4({HEIGHT},{WIDTH},{COUNT}) uniform-random rollout from the initial state to terminal, "
                f"batch={n} boards per GPU, seed 0x{SEED:016X}+step, philox4x32-10 keyed by global game id",
                "batch_per_gpu": n,
                "global_batch": n * world,
                "env_steps_per_step": steps_total / max(args.steps, 1),
                "sharding": f"game ids split over {world} rank(s); per step {'RCCL' if backend == 'nccl' else backend} all-gather of 2-bit outcome codes ({(n + 3) // 4} B per rank), expanded to int8 rewards [{world * n}, 2] on rank 0" if world > 1 else "single GPU",
                "gathered_rewards_verified": gather_ok,
                "inflight_batches": depth,
                "waves_per_simd_per_launch": int(os.environ["BGS_ROLLOUT_WPS"]),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_connect_rollout",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": steps_per_launch * BYTES_PER_STEP,
                "kernel_ms_per_launch": kernel_ms,
                "launches_in_flight": depth,
                "valu_issue": valu,
                "note": "algorithmic = 32 B per env-step (SURVEY 8d); the fused rollout keeps boards in registers, "
                "so real HBM traffic (traffic) is far smaller and the kernel is VALU-issue bound; with "
                "launches_in_flight > 1 a launch shares the chip with its neighbours, so its own duration is longer "
                "than ms_per_step",
            },
        }
        last = args.warmup + args.steps - 1
        head = batches[last % depth].reward[:65536] if (world == 1 and not args.no_cpu_baseline) else None
        if world == 1 and depth > 1 and args.serial_compare:
            # for comparison, outside the timed region above: the same launches strictly one after the other on one
            # stream (what a caller sees who waits for each batch before starting the next)
            solo = batches[0]
            reps = min(args.steps, 40)
            solo.reset_steps()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                for i in range(reps):
                    solo.rollout(SEED + 5000 + i, from_initial=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            out["one_launch_at_a_time"] = {"value": solo.steps / dt, "unit": "env-steps/s", "ms_per_step": dt / reps * 1e3,
                                           "steps": reps}
        if head is not None:
            out["cpu_baseline"] = cpu_baseline(torch, SEED + last, head)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
