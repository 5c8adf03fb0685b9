#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched random rollout, Connect4(6,7,4), 2^20 boards per GPU, rewards delivered
to a HOST array.

One "step" = one pass of the hot path over one batch: every board of the batch is played from
Config.sample_initial_state() to its terminal state with uniformly sampled actions (enumerate -> sample ->
transition -> k-in-a-row / draw -> reward), fused in one HIP launch (k_connect_rollout_opened), and the step's
rewards int8[batch, 2] are handed over to a host array ("... to rewards resident in one host array";
the reference returns `reward` as a host ndarray, connect.cpp:41).  env-steps are the transitions applied to running
boards (masked no-ops are not counted); they are counted on the device.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Hand-over (`--handover`):
  codes  (default) the device packs 2-bit outcome codes (0.25 B per game), one asynchronous copy moves them into a
         page-locked slot and host worker threads expand them into the int8 pairs of the step's host array
         (bgs_sink_*); 256 KiB instead of 2 MiB cross PCIe per step;
  pairs  one asynchronous copy of the int8[batch, 2] reward buffer into a page-locked host array
         (bgs_rollout_to_host);
  none   rewards stay on the device (the round-1 measurement; reported as `device_resident` beside `value` otherwise).

N > 1: one process per GPU, rank r owns global game ids [r * 2^20, (r+1) * 2^20) (RNG streams are keyed by global
game id, so the shards reproduce the unsharded run).  The only exchange is the hand-over into THE one host array
int8[N * 2^20, 2] (`--gather`): by default that array lives in shared memory mapped by every rank of the node and each
rank's own sink delivers its rows -- the one-GPU loop on every rank, every GPU on its own PCIe link, no collective in the
data path (simulator/sharding.py: SharedRewardRing); `--gather rccl` gathers the ranks' 2-bit outcome codes to rank 0's
GPU over RCCL (256 KiB per rank over xGMI), whose sink copies them to the host and expands them all.  Plus one
all-reduce of the step counters after the timed region.  Weak scaling.

Environment (experiments): BGS_BENCH_SLOT_FACTOR (host arrays / sink slots per stream, default 3), BGS_BENCH_TRACE=1
(where the timed region's time goes), BGS_FORCE_DIST=1 / BGS_DIST_BACKEND=gloo (the N > 1 loops on a one-GPU box).

Steps run on `--inflight` batches / HIP streams in rotation: a rollout is bound by VALU instruction issue, its drain
(the last game of every lane) leaves SIMDs idle that the next launch fills, and the copy engine and the host workers
deliver step i while steps i+1.. play.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : bound "hbm" = the bytes the rollout kernel really moves per launch over its mean duration (HIP events
                 on the launch stream) against 8 TB/s -- small by design: boards live in registers -- next to the
                 SURVEY 8d algorithmic figure it avoids, the VALU-issue rate that actually binds (against the guide's
                 SIMD-32 peak and the measured ceiling of this instruction mix) and the PCIe share of the hand-over;
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
STORED_BYTES_PER_GAME = 19   # what the fused rollout really writes per finished game: 2 x 8 B planes + status + reward pair
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0         # PCIe Gen5 x16 spec (MI355X_MICROARCH.md)
VALU_PEAK_SIMD32 = 256 * 4 * 2.4e9 / 2 / 1e9  # G wave64-instr/s: 256 CUs x 4 SIMD-32, 2 cycles per wave64 instruction
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r02_rollout_counters.json")


def cpu_baseline(last_seed, host_reward_head):
    """Time the oracle on this host's cores on a bounded sample of the same workload, and use the same run to
    cross-check the rewards the last timed step delivered (first games of the host array)."""
    import numpy as np

    from oracle import oracle

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cap = int(os.environ.get("BGS_CPU_THREADS", "16"))  # a 1-GPU box's CPU share is 16 cores
    cores = min(avail, cap)
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
        if r == 0 and host_reward_head is not None:
            parity = bool(np.array_equal(orc.reward[: host_reward_head.shape[0]], host_reward_head))
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
    capped = f"; {avail} cores visible, capped at BGS_CPU_THREADS={cap}" if avail > cap else ""
    return {
        "value": total / elapsed,
        "unit": "env-steps/s",
        "cores": cores,
        "single_thread_value": single,
        "kind": "port",
        "sample": f"{reps} x 2^20 Connect4(6,7,4) games from the initial state ({total} env-steps), CPU oracle "
        f"(oracle/bgs_oracle.c, OpenMP, {cores} threads{capped}); the reference's own core is not buildable offline",
        "parity_with_host_rewards": parity,
    }


def committed_counters(build_id):
    """Per-launch PMC figures of the rollout kernel, valid only for the build they were measured on."""
    if not os.path.exists(COUNTERS_FILE):
        return None, "no counters file"
    with open(COUNTERS_FILE) as fh:
        c = json.load(fh)
    if c.get("build_id") != build_id:
        return None, f"counters were taken on build {c.get('build_id')}, this is {build_id}: not quoted"
    return c, None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="boards per GPU (default 2^20)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight per GPU: step i runs on batch i %% D / HIP stream i %% D, so the drain of one "
                    "rollout, its copy to the host and (N > 1) its reward gather overlap the next rollouts (default 3)")
    ap.add_argument("--handover", choices=("codes", "pairs", "none"), default="codes",
                    help="how a step's rewards reach the host array (see the module docstring)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="worker threads of the reward sink (--handover codes); 0 = 6 on one GPU, min(12, 4 + 2 N) on N "
                    "(rank 0 expands N x 2 MiB of rewards per step: tools/sink_rate.py)")
    ap.add_argument("--gather", default="shm",
                    help="N > 1: how the ranks' rewards reach the one host array. shm (default): the array is in shared "
                         "memory and every rank's own sink delivers its rows (no collective, every GPU uses its own PCIe "
                         "link); rccl: outcome codes gathered to rank 0's GPU over RCCL, rank 0's sink expands them all")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-device-resident", action="store_true",
                    help="skip the extra, separately timed pass without hand-over that fills `device_resident`")
    args = ap.parse_args()
    if args.gather not in ("shm", "rccl"):
        print("bench.py: --gather must be shm or rccl", file=sys.stderr)
        return 2

    import numpy as np
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

    # one rank per GPU; BGS_DIST_BACKEND=gloo is a rehearsal mode (ranks may then share a GPU, codes are gathered
    # through host copies) used to exercise the N > 1 code path where RCCL cannot run (e.g. a one-GPU box)
    backend = os.environ.get("BGS_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    # BGS_FORCE_DIST=1 takes the N > 1 code path (process group; shared host array or per-step RCCL gather) with whatever world
    # size the launcher gave, 1 included: the way to run that path over RCCL on a one-GPU box
    sharded = world > 1 or os.environ.get("BGS_FORCE_DIST") == "1"
    if sharded:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from simulator.batch import ConnectBatch, HostEvent, PinnedArray, RewardSink, expand_outcomes_host
    from simulator.game import _abi
    from simulator.sharding import SharedRewardRing, gather_outcomes_to, shard_range, sum_steps

    # N > 1, default: one host array in shared memory, every rank delivers its own rows with its own sink
    ring_mode = sharded and args.gather == "shm" and args.handover == "codes"
    if args.host_threads <= 0:
        args.host_threads = 6 if world == 1 else 4 if ring_mode else min(12, 4 + 2 * world)
    n = args.batch
    if sharded and n % 4:
        print("bench.py: --batch must be a multiple of 4 when sharded (4 outcome codes per byte)", file=sys.stderr)
        return 2
    handover = args.handover
    if sharded and handover == "pairs":
        handover = "codes"  # ranks exchange codes; int8 pairs would put 8x the bytes on xGMI and on rank 0's PCIe link
    depth = max(1, args.inflight)
    os.environ.setdefault("BGS_ROLLOUT_WPS", "2")  # waves per SIMD per launch; `depth` launches share the chip
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(depth)] if depth > 1 else [torch.cuda.current_stream()]
    gpu = torch.device("cuda", local_rank)
    code_bytes = (n + 3) // 4
    owner = rank == 0  # rank 0 owns "the one host array"
    # per in-flight slot: the batch; (N > 1) its packed outcome codes and, on rank 0, the gathered codes of all ranks;
    # the HOST array the step's rewards end in: int8[world * n, 2] on rank 0 (N = 1: int8[n, 2])
    # One GPU: the hand-over pipeline is deeper than the GPU's (three times as many host arrays / sink slots as streams),
    # so the launching thread waits for the delivery of step i - 3 * depth, not i - depth, before it enqueues step i:
    # waiting on the previous step of the SAME stream would leave the GPU one batch short for the length of the delivery.
    # (Measured, tools/slots_sweep.sh: 3 per stream is 3 % faster than 2 on a 20-step run -- every stream always has a
    # launch queued behind the running one, whatever the host does for ~100 us -- and 4 is slower: more arrays than the
    # caches hold.)
    host_slots = depth if sharded and not ring_mode else int(os.environ.get("BGS_BENCH_SLOT_FACTOR", "3")) * depth
    ring = None
    if ring_mode:
        try:
            ring = SharedRewardRing(dist, n, host_slots)
        except RuntimeError as exc:  # raised on every rank or on none: all ranks fall back together
            if rank == 0:
                print(f"bench.py: {exc}; falling back to --gather rccl", file=sys.stderr)
            ring_mode = False
            host_slots = depth
            if args.host_threads == 4:
                args.host_threads = min(12, 4 + 2 * world)
    batches, packed, packed_buf, all_packed, host_rewards, events = [], [], [], [], [], []
    for s in streams:
        with torch.cuda.stream(s):
            b = ConnectBatch(HEIGHT, WIDTH, COUNT, n, device=local_rank, use_torch=True)  # ordered onto stream s
            b.set_first_game(shard_range(n * world, rank, world)[0])
            batches.append(b)
            packed_buf.append(torch.zeros((n + 63) // 64 * 16, dtype=torch.uint8, device=gpu) if sharded and not ring_mode else None)
            packed.append(packed_buf[-1][:code_bytes] if sharded and not ring_mode else None)
            all_packed.append(torch.empty(world * code_bytes, dtype=torch.uint8, device=gpu) if sharded and owner and not ring_mode else None)
    for slot in range(host_slots):
        if ring_mode:
            host_rewards.append(ring.mine(slot))  # this rank's rows of the shared array (touched by the ring already)
            events.append(None)
        elif handover == "pairs":
            host_rewards.append(PinnedArray((n, 2), np.int8))
            events.append(HostEvent(local_rank))
        elif handover == "codes" and owner:
            # written by the sink's worker threads; filled here so that every page is mapped before the clock starts
            host_rewards.append(np.full((world * n, 2), 0x55, dtype=np.int8))
            events.append(None)
        else:
            host_rewards.append(None)
            events.append(None)
    if ring_mode:
        sink = RewardSink(n, slots=host_slots, threads=max(1, args.host_threads), device=local_rank)
    else:
        sink = RewardSink(world * n, slots=host_slots, threads=max(1, args.host_threads), device=local_rank) \
            if handover == "codes" and owner else None
    device = gpu if backend == "nccl" else torch.device("cpu")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    tickets = [None] * host_slots  # the sink ticket of the step last delivered into each host array
    ticket_step = [0] * host_slots  # (shared array: the step that ticket belongs to, published once it is delivered)
    pending = [None] * depth       # the in-flight reward gather of each stream slot (N > 1)

    def settle(k, final=False):
        """Bring slot k's earlier steps one stage further before the slot is reused.  The stages of a step are
        rollout -> (N > 1: gather of the codes to rank 0 -> hand the gathered codes to the sink) -> rewards in
        host_rewards[k]; the wait for a stage happens one turn of the slot later, so the host never blocks on work it
        has only just enqueued.  final=True completes everything (end of the timed region)."""
        if tickets[k] is not None:      # the step submitted to the sink one turn ago: its rewards are in the host array
            sink.wait(tickets[k])
            tickets[k] = None
            if ring is not None:
                ring.publish(ticket_step[k])  # this rank's rows of that step are in the shared array
        if k < depth and pending[k] is not None:      # the gather started one turn ago
            pending[k].wait()           # (only makes stream k wait for the collective)
            pending[k] = None
            if owner:                   # the gathered codes are on rank 0's device: the sink takes them to the host
                tickets[k] = sink.submit_packed(all_packed[k], world * n, host_rewards[k], stream=streams[k].cuda_stream)
        if final and tickets[k] is not None:
            sink.wait(tickets[k])
            tickets[k] = None
        if handover == "pairs" and events[k] is not None and events[k].armed:
            events[k].synchronize()
            events[k].armed = False

    for e in events:
        if e is not None:
            e.armed = False

    def one_step(i, with_handover, ev=None):
        k = i % depth
        b = batches[k]
        if dist is None or ring_mode:
            # one GPU, or N GPUs delivering into the shared host array (every rank runs the one-GPU loop on its rows):
            # library calls only (each batch is bound to its own stream), no torch stream switching
            h = i % host_slots
            if with_handover:
                settle(h)
            if ev is not None:
                ev[0].record(streams[k])
            if not with_handover or handover == "none":
                b.rollout(SEED + i, from_initial=True)
            elif handover == "codes":
                tickets[h] = sink.rollout(b, host_rewards[h], SEED + i, from_initial=True)
                ticket_step[h] = i
            else:
                b.rollout_to_host(host_rewards[h], SEED + i, from_initial=True, codes=False, event=events[h])
                events[h].armed = True
            if ev is not None:
                ev[1].record(streams[k])  # (with a hand-over the bracket includes the pack kernel or the copy: the
                # rollout kernel's own duration is taken from the device-resident pass then)
            return
        with torch.cuda.stream(streams[k]):
            if with_handover:
                settle(k)
            if ev is not None:
                ev[0].record(streams[k])
            if not with_handover or handover == "none":
                b.rollout(SEED + i, from_initial=True)
                if ev is not None:
                    ev[1].record(streams[k])
                return
            # N > 1: the path's only exchange -- every rank's outcome codes to rank 0 (RCCL over xGMI), asynchronous, so
            # the stream goes straight on to its next rollout; rank 0's sink takes the codes to the host one turn later.
            # The rollout kernel writes the codes into the send buffer itself (bgs_rollout_pack)
            b.rollout_outcomes_tensor(packed_buf[k], SEED + i, from_initial=True)
            if ev is not None:
                ev[1].record(streams[k])
            if backend == "nccl":
                pending[k] = gather_outcomes_to(dist, packed[k], all_packed[k] if owner else None, dst=0, async_op=True)
            else:
                got = gather_outcomes_to(dist, packed[k].cpu(), torch.empty(world * code_bytes, dtype=torch.uint8) if owner else None, dst=0)
                if owner:
                    expand_outcomes_host(got.numpy(), world * n, host_rewards[k])

    def drain():
        for k in range(host_slots):
            with torch.cuda.stream(streams[k % depth]):
                settle(k, final=True)

    def timed_region(first_step, count, with_handover, stride):
        evs = {}
        for b in batches:
            b.reset_steps()
        barrier()
        t0 = time.perf_counter()
        for i in range(count):
            ev = None
            if stride and i % stride == 0:
                ev = evs[i] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            one_step(first_step + i, with_handover, ev)
        t_enqueued = time.perf_counter()
        if with_handover:
            drain()  # every step's rewards are in their host array before the clock stops
        t_drained = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        if os.environ.get("BGS_BENCH_TRACE"):
            print(f"[trace] {count} steps, handover={with_handover}: enqueued at {(t_enqueued - t0) * 1e3:.3f} ms, "
                  f"rewards on the host at {(t_drained - t0) * 1e3:.3f} ms, device idle at {dt * 1e3:.3f} ms", file=sys.stderr)
        steps_local = sum(b.steps for b in batches)
        kernel_ms = sum(s.elapsed_time(e) for s, e in evs.values()) / max(len(evs), 1) if evs else None
        return dt, steps_local, kernel_ms

    for i in range(args.warmup):
        one_step(i, True)
    drain()
    # HIP-event pairs bracket a sample of the launches (each record is a marker packet on the stream): about 32 pairs
    stride = max(2, args.steps // 32)
    elapsed, steps_local, kernel_ms = timed_region(args.warmup, args.steps, True, stride)
    last = args.warmup + args.steps - 1

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        steps_total = sum_steps(dist, steps_local, device)
    else:
        steps_total = steps_local

    # the host array of the LAST timed step (a copy: the extra passes below reuse the slots)
    final_host = None
    if owner and handover != "none":
        slot = last % host_slots
        final_host = np.array(ring.array(slot) if ring is not None else
                              host_rewards[slot].array if handover == "pairs" else host_rewards[slot])

    gather_ok = None
    if dist is not None and owner:
        # the host array must hold every rank's rewards in global game order: rank 0 re-plays the first games of the
        # LAST rank's shard on its own GPU (RNG streams are keyed by global game id) and compares
        probe = ConnectBatch(HEIGHT, WIDTH, COUNT, 4096, device=local_rank, use_torch=True)
        probe.set_first_game((world - 1) * n)
        probe.rollout(SEED + last, from_initial=True)
        gather_ok = bool((probe.reward == final_host[(world - 1) * n : (world - 1) * n + 4096]).all())
        probe.close()

    # the same launches without the hand-over (rewards stay on the device), timed separately: what the hand-over costs
    device_resident = None
    if not args.no_device_resident and handover != "none":
        reps = min(args.steps, 100)
        dt, st_local, k_ms = timed_region(last + 1, reps, False, max(2, reps // 32))
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            st_local = sum_steps(dist, st_local, device)
        device_resident = {"value": st_local / dt, "unit": "env-steps/s", "ms_per_step": dt / reps * 1e3, "steps": reps,
                           "kernel_ms_per_launch": k_ms}
        if dist is None:
            kernel_ms = k_ms  # the hand-over brackets include the pack kernel / the copy

    if rank == 0:
        value = steps_total / elapsed
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        steps_per_launch = steps_local / max(args.steps, 1)
        build = _abi.build_id()
        counters, why_not = committed_counters(build) if n == BATCH_PER_GPU else (None, "counters are for batch 2^20")
        stored = STORED_BYTES_PER_GAME * n  # by construction: every game is written exactly once, when it ends
        achieved = stored / (kernel_ms * 1e-3) / 1e9
        valu = {"note": why_not}
        if counters:
            instr = counters["valu_wave_instructions_per_launch"]
            rate = instr / (ms_per_step * 1e-3) / 1e9
            mix_peak = VALU_PEAK_SIMD32 * 2.0 / counters["mix_cycles_per_instruction"]
            valu = {
                "wave_instr_per_launch": instr,
                "achieved_Ginstr_per_s": rate,
                "peak_simd32_Ginstr_per_s": VALU_PEAK_SIMD32,
                "frac_of_simd32_peak": rate / VALU_PEAK_SIMD32,
                "mix_ceiling_Ginstr_per_s": mix_peak,
                "frac_of_mix_ceiling": rate / mix_peak,
                "mix_cycles_per_instruction": counters["mix_cycles_per_instruction"],
                "basis": "SIMD-32 peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction "
                "(MI355X_MICROARCH.md); mix ceiling = the same with this kernel's measured cycles per instruction "
                "(tools/valu_mix.py: loop-body instruction mix x tools/ubench.hip issue costs); instruction count from "
                f"rocprofv3 SQ_INSTS_VALU on build {build}; launches overlap, so the rate uses ms_per_step",
            }
        to_host = {"none": 0, "pairs": 2 * n, "codes": code_bytes * world}[handover]
        out = {
            "metric": "env-steps/sec, Connect4(6,7,4) random rollout, batch="
            + ("2^20" if n == BATCH_PER_GPU else str(n)) + " per GPU",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
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
                "rewards_to_host": handover != "none",
                "handover": {"codes": f"2-bit outcome codes ({code_bytes * world} B per step over PCIe) -> page-locked slot -> "
                             f"{args.host_threads} host threads expand into int8[{world * n}, 2]",
                             "pairs": f"int8[{n}, 2] reward buffer ({2 * n} B per step over PCIe) -> page-locked host array",
                             "none": "rewards stay on the device"}[handover],
                "sharding": (f"game ids split over {world} rank(s); no data-path collective: the host array int8[{world * n}, 2] is "
                             f"in shared memory and every rank's own sink delivers its rows ({code_bytes} B of codes per step "
                             f"over the rank's own PCIe link, {args.host_threads} host threads per rank)" if ring_mode else
                             f"game ids split over {world} rank(s); per step {'RCCL' if backend == 'nccl' else backend} gather of "
                             f"2-bit outcome codes ({code_bytes} B per rank) to rank 0" if sharded else "single GPU"),
                "gathered_rewards_verified": gather_ok,
                "inflight_batches": depth,
                "waves_per_simd_per_launch": int(os.environ["BGS_ROLLOUT_WPS"]),
                "build_id": build,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_connect_rollout_opened",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": counters["hbm_bytes_per_launch"] if counters else None,
                "bytes_per_launch": stored,
                "kernel_ms_per_launch": kernel_ms,
                "event_pairs": len(range(0, args.steps, stride)),
                "launches_in_flight": depth,
                "algorithmic_bytes_per_launch": steps_per_launch * BYTES_PER_STEP,
                "algorithmic_GBps_avoided": steps_per_launch * BYTES_PER_STEP / (kernel_ms * 1e-3) / 1e9,
                "valu_issue": valu,
                "pcie": {"bytes_per_step": to_host, "achieved_GBps": to_host / (ms_per_step * 1e-3) / 1e9,
                         "peak_GBps": PCIE_PEAK_GBS, "frac": to_host / (ms_per_step * 1e-3) / 1e9 / PCIE_PEAK_GBS},
                "note": "achieved = bytes the kernel really moves (19 B per finished game: planes, status, reward; boards "
                "live in registers from first to last ply) / its mean duration: the kernel is NOT HBM-bound, the fraction "
                "is small by design. algorithmic_* is SURVEY 8d's 32 B per env-step model, i.e. the per-ply traffic the "
                "fusion avoids (it may exceed the HBM peak and is not a roofline). The binding resource is VALU issue "
                "(valu_issue). With launches_in_flight > 1 a launch shares the chip with its neighbours, so its own "
                "duration is longer than ms_per_step.",
            },
        }
        if device_resident is not None:
            device_resident["host_over_device"] = value / device_resident["value"]
            out["device_resident"] = device_resident
        if world == 1 and not args.no_cpu_baseline:
            head = final_host[:65536] if final_host is not None else batches[last % depth].reward[:65536]
            out["cpu_baseline"] = cpu_baseline(SEED + last, head)
        print(json.dumps(out), flush=True)

    if sink is not None:
        sink.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
