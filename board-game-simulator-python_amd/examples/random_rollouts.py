#!/usr/bin/env python3
"""The reference's random-play loop (README.md:45-72; textual/examples/agent.py RandomAgent), three ways:
one object at a time through the drop-in classes, a policy-driven batch (the caller picks the columns on the device),
and the fused batched rollout.  Needs one MI355X.

    python board-game-simulator-python_amd/examples/random_rollouts.py
"""

import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from simulator.batch import ConnectBatch
from simulator.game.connect import Config


def one_game_with_objects() -> None:
    state = Config(6, 7, 4).sample_initial_state()
    while not state.has_ended:
        state = random.choice(state.actions).sample_next_state()
    print("object API   : reward", state.reward.tolist(), "after", int((state.grid >= 0).sum()), "plies")


def policy_driven_batch(n: int = 4096) -> None:
    """An external policy chooses the moves: here 'leftmost legal column', computed from the legal mask."""
    batch = ConnectBatch(6, 7, 4, n)
    while not batch.has_ended.all():
        legal = batch.legal                      # uint8[n, 7]; use legal_tensor() to stay on the device
        columns = np.where(legal.any(axis=1), legal.argmax(axis=1), -1).astype(np.int32)
        batch.step_actions(columns)              # -1 skips boards that have ended
    wins = (batch.reward[:, 0] == 1).mean()
    print(f"policy batch : {n} games, player 0 wins {wins:.0%} with the leftmost-column policy")


def fused_rollouts(n: int = 1 << 20) -> None:
    batch = ConnectBatch(6, 7, 4, n)
    batch.rollout(seed=1, from_initial=True)     # warm-up launch
    batch.reset_steps()
    t0 = time.perf_counter()
    batch.rollout(seed=2, from_initial=True)
    batch.synchronize()
    dt = time.perf_counter() - t0
    reward = batch.reward
    print(f"fused rollout: {n} games, {batch.steps} env-steps in {dt * 1e3:.2f} ms; "
          f"P(first player wins) = {(reward[:, 0] == 1).mean():.3f}, draws = {(reward[:, 0] == 0).mean():.4f}")


if __name__ == "__main__":
    one_game_with_objects()
    policy_driven_batch()
    fused_rollouts()
