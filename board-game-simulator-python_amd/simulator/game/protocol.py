"""Structural types every game module satisfies (name contract of the reference's
src/simulator/game/protocol.py:8-29: num_players, sample_initial_state, config, has_ended, player, reward,
actions, state, sample_next_state)."""

from __future__ import annotations

from typing import ClassVar, List, Protocol, runtime_checkable

import numpy as np


@runtime_checkable
class ActionLike(Protocol):
    state: "StateLike"

    def sample_next_state(self) -> "StateLike": ...


@runtime_checkable
class StateLike(Protocol):
    Action: ClassVar[type]
    config: "ConfigLike"
    has_ended: bool
    player: int
    reward: np.ndarray
    actions: List[ActionLike]


@runtime_checkable
class ConfigLike(Protocol):
    State: ClassVar[type]
    num_players: int

    def sample_initial_state(self) -> StateLike: ...
