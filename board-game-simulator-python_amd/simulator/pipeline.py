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

from typing import Iterable, Iterator, Optional, Tuple

import numpy as np

from .batch import RewardSink


class RolloutPipeline:
    def __init__(self, batch_cls, config_args: tuple, n: int, depth: int = 3, host_threads: int = 6, device: int = 0,
                 first_game: int = 0, max_plies: int = 2**31 - 1, host_arrays=None, arrays_per_stream: int = 3):
        """`batch_cls(*config_args, n, device=..., use_torch=True)` is built `depth` times, each bound to its own stream.
        `host_arrays`: optional list of arrays_per_stream * depth C-contiguous int8[n, 2] destinations (e.g. rows of a shared array,
        `SharedRewardRing.mine(slot)`); by default the pipeline allocates (and pre-faults) its own."""
        import torch

        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.n, self.depth, self.max_plies = int(n), int(depth), int(max_plies)
        self._torch = torch
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.depth)]
        self.batches = []
        for s in self.streams:
            with torch.cuda.stream(s):  # the batch binds to the stream that is current when it is created
                b = batch_cls(*config_args, self.n, device=device, use_torch=True)
                b.set_first_game(first_game)
                self.batches.append(b)
        self.slots = max(1, int(arrays_per_stream)) * self.depth
        if host_arrays is None:
            host_arrays = [np.full((self.n, 2), 0, dtype=np.int8) for _ in range(self.slots)]  # (np.full: pages mapped now)
        if len(host_arrays) != self.slots:
            raise ValueError(f"need {self.slots} host arrays (arrays_per_stream x depth)")
        self.host = list(host_arrays)
        self.sink = RewardSink(self.n, slots=self.slots, threads=max(1, host_threads), device=device)
        self._tickets = [None] * self.slots
        self._steps = [None] * self.slots
        self._next = 0

    # ---- one step at a time -----------------------------------------------------------------------
    def submit(self, seed: int) -> int:
        """Enqueue one step (all n boards, initial state to terminal) and return its step index.  Blocks only if the host
        array this step reuses (the one of step index - arrays_per_stream * depth) has not been collected with `result()` yet AND is
        still being delivered."""
        i = self._next
        h = i % self.slots
        if self._tickets[h] is not None:
            self.sink.wait(self._tickets[h])  # the array is about to be overwritten: its previous delivery must be over
        self._tickets[h] = self.sink.rollout(self.batches[i % self.depth], self.host[h], int(seed), self.max_plies, from_initial=True)
        self._steps[h] = i
        self._next = i + 1
        return i

    def result(self, step: int) -> np.ndarray:
        """The rewards int8[n, 2] of `step` (waits for their delivery).  The array is reused by step + arrays_per_stream * depth."""
        h = step % self.slots
        if self._steps[h] != step:
            raise KeyError(f"step {step} is not in flight any more (its host array was reused)")
        if self._tickets[h] is not None:
            self.sink.wait(self._tickets[h])
            self._tickets[h] = None
        return self.host[h]

    # ---- the loop ----------------------------------------------------------------------------------
    def run(self, seeds: Iterable[int]) -> Iterator[Tuple[int, np.ndarray]]:
        """Yield (step, rewards) for every seed, in order, keeping `depth` steps ahead of the consumer."""
        pending = []
        for seed in seeds:
            pending.append(self.submit(seed))
            if len(pending) > self.depth:
                step = pending.pop(0)
                yield step, self.result(step)
        for step in pending:
            yield step, self.result(step)

    @property
    def env_steps(self) -> int:
        """Transitions applied on the device so far (summed over the batches)."""
        return sum(b.steps for b in self.batches)

    def close(self) -> None:
        if getattr(self, "sink", None) is not None:
            for h, t in enumerate(self._tickets):
                if t is not None:
                    self.sink.wait(t)
                    self._tickets[h] = None
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
