"""``simulator.game.connect`` -- drop-in for the reference's nanobind module (src/simulator/game/connect.cpp:19-62).

Same names and semantics as the reference's ``Config`` / ``State`` / ``Action`` (connect.pyi:5-43); every rule
(legal columns, the drop, the k-in-a-row / draw test, the reward) is evaluated by the HIP kernels behind libbgs.so
on a one-board device batch -- there is no Python or CPU re-implementation of the game here.  For many boards at a
time use ``simulator.batch.ConnectBatch``.

Cell codes (tests/test_connect.py:24-25): -1 empty, 0 / 1 the players; ``grid[0]`` is the bottom row.
"""

from __future__ import annotations

import ctypes
import os
import threading
from typing import Any, ClassVar, Dict, List, Tuple

import numpy as np

from . import _abi
from ._engine import EngineBase, EngineCache
from ._value import ValueObject
from ..batch import ConnectBatch

_DEVICE = int(os.environ.get("BGS_DEVICE", "0"))


class _Engine(EngineBase):
    """One-board device batch that evaluates transitions for the object API: one per geometry AND per calling thread,
    each on a HIP stream of its own, so the thread pools the reference's callers use (8 boards at once in
    textual/examples/arena.py:53, agent.py:61,71) run their round trips side by side instead of queueing on one lock."""

    _cache = EngineCache()

    def __init__(self, height: int, width: int, count: int):
        self.batch = ConnectBatch(height, width, count, 1, device=_DEVICE, use_torch=False)
        self.lock = threading.Lock()  # (uncontended: the engine belongs to one thread; States may cross threads, engines do not)
        self._own_stream(_DEVICE)
        self.call = self.batch.one_board_call()
        self.held = None  # (grid bytes, player, winner) of the board the device batch holds, when known

    @classmethod
    def get(cls, height: int, width: int, count: int) -> "_Engine":
        return cls._cache.get((height, width, count), lambda: _Engine(height, width, count))

    def _round_trip(self, grid=None, player=0, winner=-1, column=None):
        """One fused call (bgs_transition): optional load, optional move, then the observations a State needs.  The
        load is left out when the device batch already holds exactly that board (the usual case: a game played move by
        move), which saves the pack kernel and the grid's trip to the device."""
        call = self.call
        held = self.held
        if grid is not None and held is not None and grid is held[0] and player == held[1] and winner == held[2]:
            grid = None  # the very array the last round trip handed out (read-only, owned by its State)
        self.held = None
        status = call(grid, player, winner, None, column)
        if status == _abi.BGS_ERR_ILLEGAL:
            raise RuntimeError(f"illegal action: column {column}")
        if status != 0:
            raise RuntimeError("malformed Connect state")
        legal = tuple([c for c, open_ in enumerate(call.legal_out[0].tolist()) if open_])
        grid_out, reward_out = call.grid_out.copy(), call.reward_out.copy()
        grid_out.flags.writeable = False
        reward_out.flags.writeable = False
        player_out, winner_out = int(call.player_out[0]), int(call.winner_out[0])
        self.held = (grid_out, player_out, winner_out)
        return grid_out, player_out, winner_out, legal, reward_out

    def initial(self):
        with self.lock:
            self.held = None
            self.batch.reset()
            return self._round_trip()

    def load(self, grid: np.ndarray, player: int, winner: int):
        with self.lock:
            return self._round_trip(grid, player, winner)

    def after(self, grid: np.ndarray, player: int, winner: int, column: int):
        with self.lock:
            return self._round_trip(grid, player, winner, column)


class Config(ValueObject):
    """``Config(height, width, count)`` -- positional ints, as bound at connect.cpp:26."""

    __slots__ = ("height", "width", "count")
    num_players: ClassVar[int] = 2
    State: ClassVar[type]

    def __init__(self, height: int, width: int, count: int) -> None:
        for v in (height, width, count):
            if isinstance(v, bool) or not isinstance(v, (int, np.integer)):
                raise TypeError("Config(height: int, width: int, count: int)")
        object.__setattr__(self, "height", int(height))
        object.__setattr__(self, "width", int(width))
        object.__setattr__(self, "count", int(count))
        # geometry limits are the packed representation's (host-side check, no GPU needed); ValueError if outside
        nbytes = ctypes.c_size_t()
        _abi.check(_abi.lib().bgs_connect_arena_bytes(self.height, self.width, self.count, 1, ctypes.byref(nbytes)))

    def __setattr__(self, name, value):
        raise AttributeError("Config is immutable")

    def _key(self):
        return (self.height, self.width, self.count)

    def __repr__(self):
        return f"Config({self.height}, {self.width}, {self.count})"

    def _engine(self) -> _Engine:
        return _Engine.get(self.height, self.width, self.count)

    def sample_initial_state(self) -> "State":
        return State._fresh(self, *self._engine().initial())

    def to_json(self) -> Dict[str, Any]:
        return {"height": self.height, "width": self.width, "count": self.count}

    @staticmethod
    def from_json(value: Dict[str, Any]) -> "Config":
        try:
            return Config(value["height"], value["width"], value["count"])
        except (KeyError, TypeError) as exc:
            raise RuntimeError(f"invalid Connect config JSON: {exc}") from None


class State(ValueObject):
    __slots__ = ("config", "_grid", "_player", "_winner", "_legal", "_reward")
    Action: ClassVar[type]

    def __init__(self, config: Config, grid: np.ndarray, player: int, winner: int, legal: Tuple[int, ...], reward):
        object.__setattr__(self, "config", config)
        g = np.array(grid, dtype=np.int8)
        g.setflags(write=False)
        object.__setattr__(self, "_grid", g)
        object.__setattr__(self, "_player", int(player))
        object.__setattr__(self, "_winner", int(winner))
        object.__setattr__(self, "_legal", tuple(legal))
        r = np.array(reward, dtype=np.int8)  # the pair the device computed (State::get_reward, connect.cpp:41)
        r.setflags(write=False)
        object.__setattr__(self, "_reward", r)

    @classmethod
    def _fresh(cls, config, grid, player, winner, legal, reward) -> "State":
        """A State of what an engine's round trip returned: read-only arrays nobody else holds, ints, a tuple."""
        s = _new(cls)
        _state_config(s, config)
        _state_grid(s, grid)
        _state_player(s, player)
        _state_winner(s, winner)
        _state_legal(s, legal)
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
        return self._grid.copy()  # the reference hands out a fresh copy per access (tensor.hpp:80-84)

    @property
    def reward(self) -> np.ndarray:
        return self._reward.copy()

    @property
    def actions(self) -> List["Action"]:
        of = Action._of
        return [of(self, c) for c in self._legal]

    def action_at(self, column: int) -> "Action":
        if isinstance(column, bool) or not isinstance(column, (int, np.integer)):
            raise TypeError("action_at(column: int)")
        if int(column) not in self._legal:
            raise RuntimeError(f"illegal action: column {int(column)}")
        return Action(self, int(column))

    def to_json(self) -> Dict[str, Any]:
        return {"grid": self._grid.tolist(), "player": self._player, "winner": self._winner}

    @staticmethod
    def from_json(value: Dict[str, Any], config: Config) -> "State":
        try:
            grid = np.array(value["grid"], dtype=np.int8)
            player, winner = int(value["player"]), int(value["winner"])
        except (KeyError, TypeError, ValueError) as exc:
            raise RuntimeError(f"invalid Connect state JSON: {exc}") from None
        if grid.shape != (config.height, config.width):
            raise RuntimeError("state grid does not match the config")
        return State._fresh(config, *config._engine().load(grid, player, winner))


class Action(ValueObject):
    __slots__ = ("state", "column")

    def __init__(self, state: State, column: int):
        object.__setattr__(self, "state", state)
        object.__setattr__(self, "column", int(column))

    @classmethod
    def _of(cls, state: State, column: int) -> "Action":
        """An Action of one of the state's own legal columns (ints already)."""
        a = _new(cls)
        _action_state(a, state)
        _action_column(a, column)
        return a

    def __setattr__(self, name, value):
        raise AttributeError("Action is immutable")

    def _key(self):
        return (self.state._key(), self.column)

    def __repr__(self):
        return f"Action(column={self.column})"

    def sample_next_state(self) -> State:
        s = self.state
        return State._fresh(s.config, *s.config._engine().after(s._grid, s._player, s._winner, self.column))

    def to_json(self) -> Dict[str, Any]:
        return {"column": self.column}

    @staticmethod
    def from_json(value: Dict[str, Any], state: State) -> "Action":
        try:
            return state.action_at(int(value["column"]))
        except (KeyError, TypeError, ValueError) as exc:
            raise RuntimeError(f"invalid Connect action JSON: {exc}") from None


Config.State = State
State.Action = Action

# The slots' own setters: the classes refuse attribute assignment (immutable values, README.md:67), and going through
# `object.__setattr__(obj, "name", value)` for each of a ply's dozen objects was a third of the Python side of a transition.
_new = object.__new__
_state_config, _state_grid, _state_player = State.config.__set__, State._grid.__set__, State._player.__set__
_state_winner, _state_legal, _state_reward = State._winner.__set__, State._legal.__set__, State._reward.__set__
_action_state, _action_column = Action.state.__set__, Action.column.__set__
