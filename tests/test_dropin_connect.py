"""The reference's Connect tests (tests/test_connect.py:68-145) replayed through the drop-in
``simulator.game.connect`` module, i.e. through libbgs.so and the HIP kernels.  Positions come from
tests/golden/reference_connect.json (transcribed data)."""

import json
import os
import random

import numpy as np
import pytest

from tests.knobs import experiment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    with open(os.path.join(golden_dir, "reference_connect.json")) as fh:
        return json.load(fh)


def assert_state(pos, state):
    np.testing.assert_array_equal(np.array(pos["grid"]), state.grid)
    if state.has_ended:
        assert pos["column"] is None
    else:
        assert pos["player"] == state.player
    return state.action_at(pos["column"]) if pos["column"] is not None else None


def test_small(fx):
    from simulator.game.connect import Config

    game = fx["games"][0]
    config = Config(*game["config"])
    state = config.sample_initial_state()
    for pos in game["positions"]:
        action = assert_state(pos, state)
        if action is not None:
            assert action.state is state
            state = action.sample_next_state()
    assert state.has_ended
    np.testing.assert_array_equal(state.reward, game["reward"])
    assert state.actions == []
    with pytest.raises(RuntimeError):
        state.action_at(0)


def test_json(fx):
    from simulator.game.connect import Action, Config, State

    j = fx["json"]
    config = Config(*j["config_args"])
    assert config.to_json() == j["config"]
    assert Config.from_json(config.to_json()) == config

    state = config.sample_initial_state()
    for col in j["state_after_columns"]:
        state = state.action_at(col).sample_next_state()
    assert state.to_json() == j["state"]
    assert State.from_json(state.to_json(), config) == state

    action = state.action_at(j["action_column"])
    assert action.to_json() == j["action"]
    assert Action.from_json(action.to_json(), state) == action


def test_surface_and_value_semantics():
    from simulator.game import ConfigLike, StateLike, ActionLike
    from simulator.game.connect import Action, Config, State

    config = Config(6, 7, 4)
    assert Config.num_players == 2 and config.num_players == 2
    assert (config.height, config.width, config.count) == (6, 7, 4)
    assert Config.State is State and State.Action is Action
    state = config.sample_initial_state()
    assert state.config is config and state.player == 0 and not state.has_ended
    assert state.grid.shape == (6, 7) and (state.grid == -1).all()
    assert state.reward.tolist() == [0, 0]
    assert [a.column for a in state.actions] == list(range(7))
    assert isinstance(config, ConfigLike) and isinstance(state, StateLike) and isinstance(state.actions[0], ActionLike)
    # immutability: the old state is untouched by a transition (README.md:67)
    nxt = state.action_at(3).sample_next_state()
    assert (state.grid == -1).all() and nxt.grid[0, 3] == 0 and nxt.player == 1
    # equality, ordering and hashing are by value (helper.hpp:10-25)
    again = config.sample_initial_state().action_at(3).sample_next_state()
    assert again == nxt and hash(again) == hash(nxt) and again is not nxt
    assert nxt != state and (nxt < state) != (state < nxt)
    assert len({nxt, again, state}) == 2
    assert state.action_at(2) == state.action_at(2) and state.action_at(2) != state.action_at(3)
    assert Config(6, 7, 4) == config and Config(6, 7, 5) != config and Config(6, 7, 4) <= config
    # illegal picks raise RuntimeError, as the UI expects (textual/connect.py:115-118)
    with pytest.raises(RuntimeError):
        state.action_at(7)
    with pytest.raises(RuntimeError):
        state.action_at(-1)
    full = state
    for _ in range(6):
        full = full.action_at(0).sample_next_state()
    assert [a.column for a in full.actions] == [1, 2, 3, 4, 5, 6]
    with pytest.raises(RuntimeError):
        full.action_at(0)


def test_readme_loop_matches_oracle():
    """README.md:45-72 verbatim loop with random.choice, every state checked against the CPU oracle."""
    from oracle import oracle
    from simulator.game.connect import Config

    rnd = random.Random(7)
    for _ in range(3):
        config = Config(6, 7, 4)
        state = config.sample_initial_state()
        orc = oracle.ConnectOracle(6, 7, 4, 1)
        while not state.has_ended:
            assert state.player == orc.player[0]
            np.testing.assert_array_equal(state.grid, orc.grid[0])
            actions = state.actions
            assert [a.column for a in actions] == np.flatnonzero(orc.legal()[0]).tolist()
            action = rnd.choice(actions)
            state = action.sample_next_state()
            orc.step_actions([action.column])
        assert orc.ended[0]
        np.testing.assert_array_equal(state.grid, orc.grid[0])
        np.testing.assert_array_equal(state.reward, orc.reward[0])


def test_json_of_terminal_and_illegal_inputs():
    from simulator.game.connect import Action, Config, State

    config = Config(2, 2, 3)  # can only be drawn
    state = config.sample_initial_state()
    for col in (0, 0, 1, 1):
        state = state.action_at(col).sample_next_state()
    assert state.has_ended and state.reward.tolist() == [0, 0] and state.actions == []
    j = state.to_json()
    assert j["winner"] == 2 and j["player"] == 0
    again = State.from_json(j, config)
    assert again == state and again.has_ended and again.actions == []
    # a finished game: winner survives the round trip, further moves are refused
    config = Config(6, 7, 4)
    state = config.sample_initial_state()
    for col in (0, 1, 0, 1, 0, 1, 0):
        state = state.action_at(col).sample_next_state()
    assert state.has_ended and state.reward.tolist() == [1, -1]
    back = State.from_json(state.to_json(), config)
    assert back == state and back.reward.tolist() == [1, -1]
    with pytest.raises(RuntimeError):
        Action.from_json({"column": 3}, back)
    # malformed states are refused by the device-side validation
    bad = state.to_json()
    bad["grid"][5][6] = 0  # floating stone
    with pytest.raises(RuntimeError):
        State.from_json(bad, config)
    with pytest.raises(RuntimeError):
        State.from_json({"grid": [[0]], "player": 0, "winner": -1}, config)
    with pytest.raises(RuntimeError):
        Action.from_json({"col": 1}, config.sample_initial_state())


def test_object_api_from_a_thread_pool():
    """The reference's callers play boards from worker threads (textual/examples/arena.py:53: eight at once).  Each thread
    gets a one-board engine and a HIP stream of its own; every game played that way must be the game the oracle plays
    when it is given the same columns, and states created on one thread keep working on another."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle
    from simulator.game.connect import Config

    config = Config(6, 7, 4)

    def play(k):
        rng = random.Random(k)
        s = config.sample_initial_state()
        columns = []
        while not s.has_ended:
            a = rng.choice(s.actions)
            columns.append(a.column)
            s = a.sample_next_state()
        return columns, s

    with ThreadPoolExecutor(8) as pool:
        games = list(pool.map(play, range(24)))
        for columns, final in games:
            orc = oracle.ConnectOracle(6, 7, 4, 1)
            for c in columns:
                assert orc.step_actions(np.array([c], dtype=np.int32))[0] == 0
            np.testing.assert_array_equal(final.grid, orc.grid[0])
            np.testing.assert_array_equal(final.reward, orc.reward[0])
            assert bool(orc.ended[0]) and final.has_ended
        # a state made on a pool thread, continued on this one
        start = pool.submit(config.sample_initial_state).result()
    nxt = start.action_at(3).sample_next_state()
    assert nxt.player == 1 and nxt.grid[0, 3] == 0


def test_branching_from_one_state_reloads_the_board():
    """The engine skips the load when its device board already is the state being stepped; a second action taken from
    an OLDER state must put that state back first.  Every child of every state of a short game, visited in both orders,
    against the oracle."""
    from oracle import oracle
    from simulator.game.connect import Config

    config = Config(5, 4, 3)
    rnd = random.Random(11)
    state, history = config.sample_initial_state(), []
    while not state.has_ended:
        children = {}
        for order in (state.actions, state.actions[::-1]):
            for action in order:
                orc = oracle.ConnectOracle(5, 4, 3, 1)
                for column in history + [action.column]:
                    orc.step_actions([column])
                child = action.sample_next_state()
                np.testing.assert_array_equal(child.grid, orc.grid[0])
                assert child.has_ended == bool(orc.ended[0])
                np.testing.assert_array_equal(child.reward, orc.reward[0])
                assert [a.column for a in child.actions] == (np.flatnonzero(orc.legal()[0]).tolist() if not orc.ended[0] else [])
                assert children.setdefault(action.column, child) == child
        column = rnd.choice(sorted(children))
        history.append(column)
        state = children[column]


_PLAYTHROUGH = r"""
import hashlib, random, sys
sys.path[:0] = [{root!r}, {pkg!r}]
import numpy as np
from simulator.game.connect import Config as Connect
from simulator.game.bounce import Config as Bounce
grid = np.zeros((9, 6), dtype=np.int64); grid[1] = grid[7] = [1, 2, 3, 3, 2, 1]
h = hashlib.sha256()
for config, cap in ((Connect(6, 7, 4), 100), (Connect(12, 13, 5), 60), (Bounce(grid), 80)):
    rnd = random.Random(17)
    for _ in range(3):
        s = config.sample_initial_state()
        first, n = s, 0
        while not s.has_ended and n < cap:
            a = rnd.choice(s.actions)
            s = a.sample_next_state()
            h.update(s.grid.tobytes()); h.update(bytes([s.player, s.has_ended])); h.update(s.reward.tobytes())
            h.update(repr(len(s.actions)).encode())
            n += 1
        h.update(first.actions[0].sample_next_state().grid.tobytes())  # an older state again: the board is reloaded
print(h.hexdigest())
"""


def test_the_three_forms_of_the_round_trip_agree():
    """bgs_transition on a one-board batch: staged copies, blocks the device addresses in host memory, the same launches
    replayed from a HIP graph, and the default -- in place with move + observation fused into one kernel
    (BGS_EXPERIMENT: transition=...), the host taking the records when it sees the kernel's ticket or, transition_spin=0, when the
    stream has completed -- one digest over three playthroughs (12x13x5 and Bounce included)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _PLAYTHROUGH.format(root=root, pkg=os.path.join(root, "board-game-simulator-python_amd"))
    digests = {}
    for form in ("staged", "mapped", "graph", "fused", "fused-stream", "fused-thread"):
        env = dict(os.environ, BGS_EXPERIMENT=experiment(transition=form.split("-")[0], transition_spin="0" if form.endswith("stream") else "1",
                                                        transition_wave="0" if form.endswith("thread") else "1"))   # Bounce: a thread per board instead of a piece per lane
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[form] = out.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests


def test_engine_cache_is_bounded_and_evicted_configs_still_play():
    """A thread keeps at most 32 one-board engines per game (device batch + stream + two page-locked blocks each); a
    Config whose engine was evicted gets a new one on its next use, and its old States still step correctly."""
    from oracle import oracle
    from simulator.game import _engine
    from simulator.game.connect import Config, _Engine

    first = Config(4, 5, 3)
    s0 = first.sample_initial_state()
    child = s0.action_at(2).sample_next_state()
    for width in range(2, 2 + _engine.MAX_ENGINES_PER_THREAD + 6):
        Config(3, width, 3).sample_initial_state()
    engines = _Engine._cache._local.engines
    assert len(engines) <= _engine.MAX_ENGINES_PER_THREAD and (4, 5, 3) not in engines
    again = child.action_at(2).sample_next_state()  # an old State of the evicted Config: new engine, board reloaded
    orc = oracle.ConnectOracle(4, 5, 3, 1)
    orc.step_actions([2]); orc.step_actions([2])
    np.testing.assert_array_equal(again.grid, orc.grid[0])
    assert (4, 5, 3) in engines
