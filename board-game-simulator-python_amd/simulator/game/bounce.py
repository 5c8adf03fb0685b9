"""``simulator.game.bounce`` -- drop-in for the reference's nanobind module (src/simulator/game/bounce.cpp:19-61).

Same names and semantics as the reference's ``Config`` / ``State`` / ``Action`` (bounce.pyi:5-43).  The move search
(active row, exact-count walks with bouncing), the transition and the goal / blocked / draw test run in the HIP
kernels behind libbgs.so on a one-board device batch; nothing here re-implements the rules.  For many boards at a
time use ``simulator.batch.BounceBatch``.

Coordinates are ``(x, y)`` arrays with ``y = 0`` the bottom row (tests/test_bounce.py:27-35); ``grid[y, x]`` is 0 for
an empty cell, otherwise the piece's step count.
"""

from __future__ import annotations

import os
import threading
from typing import Any, ClassVar, Dict, List, Tuple

import numpy as np

from . import _abi
from ._engine import EngineBase, EngineCache
from ._value import ValueObject
from ..batch import BounceBatch

_DEVICE = int(os.environ.get("BGS_DEVICE", "0"))

Cell = Tuple[int, int]


def _as_cell(value, what: str) -> Cell:
    a = np.asarray(value)
    if a.shape != (2,) or not np.issubdtype(a.dtype, np.integer):
        raise TypeError(f"{what} must be an integer array of shape (2,)")
    return int(a[0]), int(a[1])


class _Engine(EngineBase):
    """One-board device batch per start grid AND per calling thread, each on a HIP stream of its own (see connect.py)."""

    _cache = EngineCache()

    def __init__(self, grid: np.ndarray):
        self.batch = BounceBatch(grid, 1, device=_DEVICE, use_torch=False)
        self.lock = threading.Lock()  # (uncontended: the engine belongs to one thread)
        self._own_stream(_DEVICE)
        self.call = self.batch.one_board_call()
        self.held = None  # (grid bytes, player, winner, plies) of the board the device batch holds, when known

    @classmethod
    def get(cls, grid: np.ndarray) -> "_Engine":
        return cls._cache.get((grid.shape, grid.tobytes()), lambda: _Engine(grid))

    def _round_trip(self, grid=None, player=0, winner=-1, plies=0, move=None):
        """One fused call (bgs_transition): optional load, optional move, then the observations a State needs.  The
        load is left out when the device batch already holds exactly that board (see connect.py)."""
        call = self.call
        held = self.held
        if grid is not None and held is not None and grid is held[0] and (player, winner, plies) == held[1:]:
            grid = None  # the very array the last round trip handed out (read-only, owned by its State)
        self.held = None
        status = call(grid, player, winner, plies, move)
        if status == _abi.BGS_ERR_ILLEGAL:
            raise RuntimeError(f"illegal action: {tuple(move[:2])} -> {tuple(move[2:])}")
        if status != 0:
            raise RuntimeError("malformed Bounce state")
        grid_out, reward_out = call.grid_out.copy(), call.reward_out.copy()
        grid_out.flags.writeable = False
        reward_out.flags.writeable = False
        player_out, winner_out, plies_out = int(call.player_out[0]), int(call.winner_out[0]), int(call.plies_out[0])
        self.held = (grid_out, player_out, winner_out, plies_out)
        moves = self.batch.decode_moves(call.legal_out[0], winner_out)
        return grid_out, player_out, winner_out, plies_out, moves, reward_out

    def initial(self):
        with self.lock:
            self.held = None
            self.batch.reset()
            return self._round_trip()

    def load(self, grid, player, winner, plies):
        with self.lock:
            return self._round_trip(grid, player, winner, plies)

    def after(self, grid, player, winner, plies, source: Cell, target: Cell):
        with self.lock:
            return self._round_trip(grid, player, winner, plies, (source[0], source[1], target[0], target[1]))


class Config(ValueObject):
    """``Config(grid)`` -- any integer 2-D array convertible to int8 (bounce.cpp:26, tensor.hpp:41-66)."""

    __slots__ = ("_grid",)
    num_players: ClassVar[int] = 2
    State: ClassVar[type]

    def __init__(self, grid) -> None:
        a = np.asarray(grid)
        if a.ndim != 2 or not np.issubdtype(a.dtype, np.integer):
            raise TypeError("Config(grid: 2-D integer ndarray)")
        if a.size and (a.min() < -128 or a.max() > 127):
            raise TypeError("Config grid does not fit int8")
        g = np.ascontiguousarray(a, dtype=np.int8)
        g.setflags(write=False)
        object.__setattr__(self, "_grid", g)
        h, w = g.shape
        if h < 3 or w < 1 or (g < 0).any() or g[0].any() or g[-1].any():
            raise RuntimeError("invalid Bounce grid: need height >= 3, non-negative values and empty goal rows")

    def __setattr__(self, name, value):
        raise AttributeError("Config is immutable")

    def _key(self):
        return (self._grid.shape, self._grid.tobytes())

    def __repr__(self):
        return f"Config({self._grid.tolist()})"

    @property
    def grid(self) -> np.ndarray:
        return self._grid.copy()

    def _engine(self) -> _Engine:
        return _Engine.get(self._grid)

    def sample_initial_state(self) -> "State":
        return State._fresh(self, *self._engine().initial())

    def to_json(self) -> Dict[str, Any]:
        return {"grid": self._grid.tolist()}

    @staticmethod
    def from_json(value: Dict[str, Any]) -> "Config":
        try:
            return Config(np.array(value["grid"], dtype=np.int64))
        except (KeyError, TypeError, ValueError) as exc:
            raise RuntimeError(f"invalid Bounce config JSON: {exc}") from None


class State(ValueObject):
    __slots__ = ("config", "_grid", "_player", "_winner", "_plies", "_moves", "_reward")
    Action: ClassVar[type]

    def __init__(self, config: Config, grid, player: int, winner: int, plies: int, moves, reward):
        object.__setattr__(self, "config", config)
        g = np.array(grid, dtype=np.int8)
        g.setflags(write=False)
        object.__setattr__(self, "_grid", g)
        object.__setattr__(self, "_player", int(player))
        object.__setattr__(self, "_winner", int(winner))
        object.__setattr__(self, "_plies", int(plies))
        object.__setattr__(self, "_moves", tuple(moves))
        r = np.array(reward, dtype=np.int8)  # the pair the device computed (State::get_reward, bounce.cpp:38)
        r.setflags(write=False)
        object.__setattr__(self, "_reward", r)

    @classmethod
    def _fresh(cls, config, grid, player, winner, plies, moves, reward) -> "State":
        """A State of what an engine's round trip returned: read-only arrays nobody else holds, ints, a tuple of moves."""
        s = _new(cls)
        _state_config(s, config)
        _state_grid(s, grid)
        _state_player(s, player)
        _state_winner(s, winner)
        _state_plies(s, plies)
        _state_moves(s, moves)
        _state_reward(s, reward)
        return s

    def __setattr__(self, name, value):
        raise AttributeError("State is immutable")

    def _key(self):
        return (self.config._key(), self._grid.tobytes(), self._player, self._winner)

    def __repr__(self):
        return f"State(player={self._player}, winner={self._winner}, grid={self._grid.tolist()})"

    @property
    def has_ended(self) -> bool:
        return self._winner != -1

    @property
    def player(self) -> int:
        return self._player

    @property
    def grid(self) -> np.ndarray:
        return self._grid.copy()

    @property
    def reward(self) -> np.ndarray:
        return self._reward.copy()

    @property
    def actions(self) -> List["Action"]:
        of = Action._of
        return [of(self, s, t) for s, t in self._moves]

    def actions_at(self, source) -> List["Action"]:
        src = _as_cell(source, "source")
        h, w = self._grid.shape
        if not (0 <= src[0] < w and 0 <= src[1] < h):
            raise RuntimeError(f"source {src} is outside the board")
        return [Action._of(self, s, t) for s, t in self._moves if s == src]

    def action_at(self, source, target) -> "Action":
        src, dst = _as_cell(source, "source"), _as_cell(target, "target")
        if (src, dst) not in self._moves:
            raise RuntimeError(f"illegal action: {src} -> {dst}")
        return Action._of(self, src, dst)

    def to_json(self) -> Dict[str, Any]:
        return {"grid": self._grid.tolist(), "player": self._player, "winner": self._winner}

    @staticmethod
    def from_json(value: Dict[str, Any], config: Config) -> "State":
        try:
            grid = np.array(value["grid"], dtype=np.int8)
            player, winner = int(value["player"]), int(value["winner"])
        except (KeyError, TypeError, ValueError) as exc:
            raise RuntimeError(f"invalid Bounce state JSON: {exc}") from None
        if grid.shape != config._grid.shape:
            raise RuntimeError("state grid does not match the config")
        return State._fresh(config, *config._engine().load(grid, player, winner, player))


class Action(ValueObject):
    __slots__ = ("state", "_source", "_target")

    def __init__(self, state: State, source: Cell, target: Cell):
        object.__setattr__(self, "state", state)
        object.__setattr__(self, "_source", (int(source[0]), int(source[1])))
        object.__setattr__(self, "_target", (int(target[0]), int(target[1])))

    @classmethod
    def _of(cls, state: State, source: Cell, target: Cell) -> "Action":
        """An Action of one of the state's own decoded moves (tuples of ints already): `state.actions` builds a dozen
        of these per ply, a third of the Python side of a transition."""
        a = _new(cls)
        _action_state(a, state)
        _action_source(a, source)
        _action_target(a, target)
        return a

    def __setattr__(self, name, value):
        raise AttributeError("Action is immutable")

    def _key(self):
        return (self.state._key(), self._source, self._target)

    def __repr__(self):
        return f"Action(source={self._source}, target={self._target})"

    @property
    def source(self) -> np.ndarray:
        return np.array(self._source, dtype=np.int64)

    @property
    def target(self) -> np.ndarray:
        return np.array(self._target, dtype=np.int64)

    def sample_next_state(self) -> State:
        s = self.state
        return State._fresh(s.config, *s.config._engine().after(s._grid, s._player, s._winner, s._plies, self._source, self._target))

    def to_json(self) -> Dict[str, Any]:
        return {"source": list(self._source), "target": list(self._target)}

    @staticmethod
    def from_json(value: Dict[str, Any], state: State) -> "Action":
        try:
            return state.action_at(np.array(value["source"]), np.array(value["target"]))
        except (KeyError, TypeError, ValueError) as exc:
            raise RuntimeError(f"invalid Bounce action JSON: {exc}") from None


Config.State = State
State.Action = Action

# The slots' own setters (see connect.py): the classes refuse attribute assignment, these are how the module fills them.
_new = object.__new__
_state_config, _state_grid, _state_player = State.config.__set__, State._grid.__set__, State._player.__set__
_state_winner, _state_plies, _state_moves = State._winner.__set__, State._plies.__set__, State._moves.__set__
_state_reward = State._reward.__set__
_action_state, _action_source, _action_target = Action.state.__set__, Action._source.__set__, Action._target.__set__
