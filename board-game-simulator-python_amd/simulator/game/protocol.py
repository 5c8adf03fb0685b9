"""Structural types every game module satisfies.

This is the NAME CONTRACT of the drop-in (reference src/simulator/game/protocol.py:8-29): a game exposes a `Config`
that samples initial states, immutable `State` objects that list their legal `Action`s, and actions that sample the
next state.  `simulator.game.connect` and `simulator.game.bounce` implement it on top of libbgs.so; the batched
classes in `simulator.batch` are the N-boards-per-call form of the same loop.
"""

from __future__ import annotations

from typing import ClassVar, List, Protocol, runtime_checkable

import numpy as np


@runtime_checkable
class ActionLike(Protocol):
    """One legal move of `state`; applying it never mutates `state` (objects are immutable values)."""

    state: "StateLike"

    def sample_next_state(self) -> "StateLike":
        """The position after this move (the name says `sample` because transitions may be stochastic in general)."""
        ...


@runtime_checkable
class StateLike(Protocol):
    """A position: whose turn it is, whether the game is over, the per-player reward once it is, the legal moves."""

    Action: ClassVar[type]
    config: "ConfigLike"
    has_ended: bool
    player: int
    reward: np.ndarray
    actions: List[ActionLike]


@runtime_checkable
class ConfigLike(Protocol):
    """Game-wide parameters; the factory of initial states."""

    State: ClassVar[type]
    num_players: int

    def sample_initial_state(self) -> StateLike:
        ...
