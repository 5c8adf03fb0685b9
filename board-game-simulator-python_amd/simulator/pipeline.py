"""Random-rollout pipeline: the loop bench.py measures, as a reusable object.

A single `ConnectBatch.rollout()` followed by `.reward` leaves the GPU idle while the rewards cross PCIe and the host
idle while the GPU plays.  `RolloutPipeline` keeps `depth` batches in flight on their own HIP streams and hands every
step's rewards to a `RewardSink` (2-bit outcome codes into page-locked slots, host threads expand them), with three times
as many host arrays as streams so that the launching thread never waits for the step it has just enqueued:

    with RolloutPipeline(ConnectBatch, (6, 7, 4), n=1 << 20) as pipe:
        for step, rewards in pipe.run(seeds=range(1000)):   # rewards: int8[n, 2], valid until 3 * depth steps later
            consume(rewards)

Every step plays all `n` boards from the initial state to the end with uniformly sampled moves; step s uses RNG seed
`seeds[s]` and global game ids `first_game + [0, n)`, so the results do not depend on depth, streams or devices
(reference: the README's `random.choice(state.actions)` loop, /root/reference README.md:57-65, batched)."""

from __future__ import annotations

import ctypes
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np

from .batch import RewardSink, _reward_destination
from .game import _abi


class RolloutExecutor:
    """The rollout loop as ONE library call per burst of steps (`bgs_pipeline_*`, csrc/bgs_pipeline.hip): step s plays
    every board of `batches[s % depth]` from the initial state to the end with seed `seed0 + s` on that batch's stream,
    and -- with a hand-over -- the step's rewards go to `host_arrays[j % len(host_arrays)]`, j counting hand-overs,
    through `sink` (a RewardSink) or `gather` (a sharding.RewardGather; ranks other than 0 pass None entries).

    `enqueue(count)` returns when `count` more steps are enqueued (it blocks only while the host array a step is about to
    reuse is still being delivered); `drain()` when every enqueued step's rewards are in their host arrays.  What a Python
    loop of `sink.rollout(...)` calls costs the launching thread per step (~27 us) is most of a 2^20-board Connect4 step
    (~34 us): this executor spends a kernel launch and an event record."""

    def __init__(self, batches: Sequence, sink=None, gather=None, host_arrays: Sequence = (), seed0: int = 0,
                 max_plies: int = 2**31 - 1, from_initial: bool = True):
        if sink is not None and gather is not None:
            raise ValueError("hand over through a sink OR a gather")
        self.batches, self.sink, self.gather = list(batches), sink, gather
        if len(self.batches) > 4 and _abi.hardware_queues() < len(self.batches):
            import warnings

            warnings.warn(f"RolloutExecutor: {len(self.batches)} batches in flight on {_abi.hardware_queues()} hardware queues: the HIP "
                          "runtime runs at most that many streams side by side (simulator.pipeline.request_hardware_queues() before "
                          "the first HIP call asks for more)", RuntimeWarning, stacklevel=2)
        self.host = list(host_arrays)  # (kept alive: worker threads write them after enqueue() has returned)
        n_games = self.batches[0].n * (gather.world if gather is not None else 1)
        ptrs = (ctypes.c_void_p * max(len(self.host), 1))()
        for k, a in enumerate(self.host):
            ptrs[k] = None if a is None else (sink._destination(a, n_games) if sink is not None else _reward_destination(a, n_games))
        handles = (_abi.c_handle * len(self.batches))(*[b._handle for b in self.batches])
        self._handle = _abi.c_handle()
        _abi.check(_abi.lib().bgs_pipeline_create(
            handles, len(self.batches), sink._handle if sink is not None else None, gather._handle if gather is not None else None,
            ptrs if self.host else None, len(self.host), ctypes.c_uint64(seed0), ctypes.c_int32(max_plies),
            ctypes.c_uint32(_abi.ROLLOUT_FROM_INITIAL if from_initial else 0), ctypes.byref(self._handle)))
        self.seed0 = int(seed0)
        self._ring = None

    def set_ring(self, ring, consumer: bool, lag: int, timeout: float = 60.0) -> None:
        """N ranks, one shared host array (`sharding.SharedRewardRing`; `ring.attach(sink)` first): hand-overs wait for
        the consumer's release of the array they overwrite, and on the consumer rank the loop itself consumes hand-over
        j - lag (waits for every rank's delivery, releases it) before it enqueues hand-over j."""
        _abi.check(_abi.lib().bgs_pipeline_set_ring(self._handle, ctypes.c_void_p(ring._words), 8, ring.world,
                                                    ctypes.c_void_p(ring._words + 64 * ring.world), 1 if consumer else 0,
                                                    int(lag), int(timeout * 1000)))
        self._ring = ring

    def enqueue(self, count: int, handover: bool = True, time_stride: int = 0) -> None:
        _abi.check(_abi.lib().bgs_pipeline_enqueue(self._handle, int(count), 1 if handover else 0, int(time_stride)))

    def enqueue_seeds(self, seeds, handover: bool = True) -> None:
        """`enqueue` with the seed of every step given (a C-contiguous uint64 array), instead of seed0 + step index."""
        arr = np.ascontiguousarray(seeds, dtype=np.uint64)
        _abi.check(_abi.lib().bgs_pipeline_enqueue_seeds(self._handle, ctypes.c_void_p(arr.ctypes.data), int(arr.size),
                                                         1 if handover else 0))

    def feed(self, seeds) -> None:
        """Hand the seeds of further steps to the pipeline's own feeder thread (`bgs_pipeline_feed`): it enqueues a step as
        soon as the host array the step lands in has been released -- `release(j)` after `wait_handover(j)`."""
        arr = np.ascontiguousarray(seeds, dtype=np.uint64)
        _abi.check(_abi.lib().bgs_pipeline_feed(self._handle, ctypes.c_void_p(arr.ctypes.data), int(arr.size)))

    def release(self, index: int) -> None:
        """The caller is done with hand-over `index`'s host array (and with every earlier one)."""
        _abi.check(_abi.lib().bgs_pipeline_release(self._handle, int(index)))

    def wait_handover(self, index: int) -> None:
        """Until hand-over number `index` is in its host array `host_arrays[index % len(host_arrays)]`."""
        _abi.check(_abi.lib().bgs_pipeline_wait(self._handle, int(index)))

    def drain(self) -> None:
        _abi.check(_abi.lib().bgs_pipeline_drain(self._handle))

    def kernel_ms(self) -> Tuple[Optional[float], int]:
        """(mean duration in ms of the launches bracketed since the last call, how many) -- after `drain()`."""
        ms, pairs = ctypes.c_double(0.0), ctypes.c_int(0)
        _abi.check(_abi.lib().bgs_pipeline_kernel_ms(self._handle, ctypes.byref(ms), ctypes.byref(pairs)))
        return (ms.value if pairs.value else None), pairs.value

    def timeline(self, capacity: int = 1024):
        """[(start_ms, end_ms)] of the bracketed launches, relative to the first one's start -- after `drain()` and before
        `kernel_ms()` (which resets the brackets)."""
        a, z = (ctypes.c_float * capacity)(), (ctypes.c_float * capacity)()
        n = ctypes.c_int(0)
        _abi.check(_abi.lib().bgs_pipeline_timeline(self._handle, a, z, capacity, ctypes.byref(n)))
        return [(a[k], z[k]) for k in range(n.value)]

    @property
    def steps(self) -> int:
        """Steps enqueued so far; the next step's seed is seed0 + steps."""
        v = ctypes.c_int64(0)
        _abi.check(_abi.lib().bgs_pipeline_progress(self._handle, ctypes.byref(v), None))
        return v.value

    @property
    def handovers(self) -> int:
        v = ctypes.c_int64(0)
        _abi.check(_abi.lib().bgs_pipeline_progress(self._handle, None, ctypes.byref(v)))
        return v.value

    def last_host_array(self):
        """The host array the most recent hand-over went to (valid after `drain()`)."""
        return self.host[(self.handovers - 1) % len(self.host)]

    def close(self) -> None:
        if self._handle:
            _abi.lib().bgs_pipeline_destroy(self._handle)
            self._handle = _abi.c_handle()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_depth(batch_cls, config_args: tuple = ()) -> int:
    """Batches in flight that saturate the chip.  A one-word Connect rollout (6x7x4) is a ~50 us launch bound by VALU
    issue at two waves per SIMD: 3 in flight fill the drain of one launch with the next, more only add to the tail.  A
    multi-word Connect board (12x13x5: the LDS-staged kernel, ONE wave per SIMD per launch, each issuing one VALU per
    ~9 cycles when alone) wants more waves per SIMD than three launches give: 8 in flight, 2.45 against 2.18 x 10^11
    env-steps/s (round 3, r3_k2c_depth.sh in the git history; 12-16 with larger chunks reach 2.5-2.56).  A Bounce rollout is a 5-15 ms launch
    whose tail is a handful of long games: 20 in flight on 32 hardware queues (16 / 20 / 24 / 28 / 32 in flight read
    10.1 / 11.0 / 10.1 / 7.9 / 5.9 x 10^9 env-steps/s at 2^18 boards: beyond 24 the hardware queues thrash), 7x one launch
    at a time."""
    if getattr(batch_cls, "game", 0) == _abi.GAME_BOUNCE:
        return 20
    if len(config_args) >= 2 and int(config_args[1]) * (int(config_args[0]) + 1) > 64:
        return 8
    return 3


def request_hardware_queues(wanted: int = _abi.HW_QUEUES_WANTED) -> int:
    """Ask for `wanted` hardware queues while the HIP runtime has not started (see `_abi.request_hardware_queues`: opt-in,
    process-wide); returns what the process has or will have."""
    return _abi.request_hardware_queues(wanted)


def usable_depth(depth: int, explicit: bool, what: str = "RolloutPipeline") -> int:
    """More than 4 batches in flight need more than the HIP runtime's default 4 hardware queues, and the runtime reads
    GPU_MAX_HW_QUEUES only once, when it initialises.  Ask for them if that is still possible; if the moment has passed, a
    depth the CALLER chose raises (it cannot be delivered), a DEFAULT depth falls back to what the process has, with a
    warning (round-3 advisor: a default-constructed pipeline must not start failing with the import order)."""
    if depth <= 4:
        return depth
    have = request_hardware_queues(max(_abi.HW_QUEUES_WANTED, depth))
    if have >= depth:
        return depth
    hint = ("call simulator.pipeline.request_hardware_queues() (or set GPU_MAX_HW_QUEUES) before the first torch.cuda / HIP call")
    if explicit:
        raise RuntimeError(f"{what}: {depth} batches in flight need GPU_MAX_HW_QUEUES >= {depth}, this process has {have}: the HIP "
                           f"runtime was initialised before more could be asked for; {hint}, or pass depth <= {max(have, 4)}.")
    import warnings

    warnings.warn(f"{what}: the default of {depth} batches in flight needs {depth} hardware queues, this process has {have} "
                  f"(the HIP runtime was up already); running {max(have, 4)} deep instead -- {hint} for the full depth.",
                  RuntimeWarning, stacklevel=3)
    return max(have, 4)


class RolloutPipeline:
    def __init__(self, batch_cls, config_args: tuple, n: int, depth: Optional[int] = None, host_threads: int = 6, device: int = 0,
                 first_game: int = 0, max_plies: int = 2**31 - 1, host_arrays=None, arrays_per_stream: int = 3):
        """`batch_cls(*config_args, n, device=..., use_torch=True)` is built `depth` times, each bound to its own stream
        (default: `default_depth(batch_cls, config_args)` -- 3 for one-word Connect boards, 8 for larger ones, 20 for Bounce).
        `host_arrays`: optional list of arrays_per_stream * depth C-contiguous int8[n, 2] destinations (e.g. rows of a shared array,
        `SharedRewardRing.mine(slot)`); by default the pipeline allocates (and pre-faults) its own."""
        explicit = depth is not None
        if depth is None:
            depth = default_depth(batch_cls, config_args)
        if depth < 1:
            raise ValueError("depth must be >= 1")
        depth = usable_depth(depth, explicit)   # (before torch touches the GPU: the queues can still be asked for)
        if not explicit:
            # Bounce: deliveries complete in step order and a step's duration varies with its longest games, so the launching
            # thread needs room to run past a slow stream (bench: 2 / 3 / 8 arrays per stream = 1.24 / 1.35 / 1.37 x 10^10)
            if getattr(batch_cls, "game", 0) == _abi.GAME_BOUNCE:
                arrays_per_stream = max(arrays_per_stream, 6)
            arrays_per_stream = min(arrays_per_stream, max(1, 256 // depth))
        import torch

        self.n, self.depth, self.max_plies = int(n), int(depth), int(max_plies)
        self._torch = torch
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.depth)]
        self.batches = []
        for s in self.streams:
            with torch.cuda.stream(s):  # the batch binds to the stream that is current when it is created
                b = batch_cls(*config_args, self.n, device=device, use_torch=True)
                b.set_first_game(first_game)
                b.set_launches_in_flight(self.depth)
                self.batches.append(b)
        self.slots = max(1, int(arrays_per_stream)) * self.depth
        if host_arrays is None:
            host_arrays = [np.full((self.n, 2), 0, dtype=np.int8) for _ in range(self.slots)]  # (np.full: pages mapped now)
        if len(host_arrays) != self.slots:
            raise ValueError(f"need {self.slots} host arrays (arrays_per_stream x depth)")
        self.host = list(host_arrays)
        self.sink = RewardSink(self.n, slots=self.slots, threads=max(1, host_threads), device=device)
        # the loop itself is native: the executor enqueues a BURST of steps per library call (bgs_pipeline_enqueue_seeds)
        self._exe = RolloutExecutor(self.batches, sink=self.sink, host_arrays=self.host, seed0=0, max_plies=self.max_plies)
        self._next = 0        # steps enqueued

    # ---- one step at a time -----------------------------------------------------------------------
    def submit(self, seed: int) -> int:
        """Enqueue one step (all n boards, initial state to terminal) and return its step index.  Blocks only while the
        host array this step reuses (the one of step index - arrays_per_stream * depth) is still being delivered."""
        return self.submit_many([seed])[0]

    def submit_many(self, seeds: Sequence[int]) -> range:
        """Enqueue len(seeds) steps with ONE library call (a Python call per step costs the launching thread most of a
        2^20-board Connect4 step); returns their step indices."""
        arr = np.ascontiguousarray(np.asarray([int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds], dtype=np.uint64))
        first = self._next
        if arr.size:
            self._exe.enqueue_seeds(arr)
            self._next += int(arr.size)
        return range(first, self._next)

    def result(self, step: int) -> np.ndarray:
        """The rewards int8[n, 2] of `step` (waits for their delivery).  The array is reused by step + arrays_per_stream * depth."""
        if step < 0 or step >= self._next or step + self.slots < self._next:
            raise KeyError(f"step {step} is not in flight any more (its host array was reused)")
        self._exe.wait_handover(step)
        return self.host[step % self.slots]

    # ---- the loop ----------------------------------------------------------------------------------
    def run(self, seeds: Iterable[int]) -> Iterator[Tuple[int, np.ndarray]]:
        """Yield (step, rewards) for every seed, in order.  The seeds are FED to the native loop (`bgs_pipeline_feed`): a
        thread of the library's own enqueues a step as soon as the host array it lands in is free, up to
        arrays_per_stream * depth steps ahead of the consumer, so this generator only waits for a step, yields its array
        and releases it when the consumer comes back for the next one -- the launches are not made from Python at all
        (round 4 made a Python call per step, round 5's first version a call per burst; both were bound by the interpreter
        on a busy host).  The array of a yielded step stays untouched until the consumer asks for the next step."""
        it = iter(seeds)
        exhausted = False
        nxt = self._next          # the next step to yield
        try:
            while True:
                # keep a few hundred steps' worth of seeds with the feeder (it enqueues them as arrays come free)
                if not exhausted and self._next - nxt < 2 * self.slots:
                    chunk = []
                    for seed in it:
                        chunk.append(int(seed) & 0xFFFFFFFFFFFFFFFF)
                        if len(chunk) >= 256:
                            break
                    else:
                        exhausted = True
                    if chunk:
                        self._exe.feed(np.asarray(chunk, dtype=np.uint64))
                        self._next += len(chunk)
                if nxt >= self._next:
                    if exhausted:
                        return
                    continue
                self._exe.wait_handover(nxt)
                yield nxt, self.host[nxt % self.slots]
                self._exe.release(nxt)   # (the consumer is back: it is done with that array)
                nxt += 1
        finally:
            # a consumer that stops early: whatever was fed is still played and delivered (its arrays are nobody's any more)
            # (a generator left suspended across close() finds no executor any more: close() has drained and dropped it)
            if nxt < self._next and self._exe is not None:
                self._exe.drain()

    @property
    def env_steps(self) -> int:
        """Transitions applied on the device so far (summed over the batches)."""
        return sum(b.steps for b in self.batches)

    def close(self) -> None:
        if getattr(self, "_exe", None) is not None:
            self._exe.close()   # (drains: every enqueued step is delivered before the sink goes)
            self._exe = None
        if getattr(self, "sink", None) is not None:
            self.sink.close()
            self.sink = None
        for b in getattr(self, "batches", []):
            b.close()
        self.batches = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
