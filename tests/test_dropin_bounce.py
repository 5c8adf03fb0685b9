"""The reference's Bounce tests (tests/test_bounce.py:92-410) replayed through the drop-in
``simulator.game.bounce`` module, i.e. through libbgs.so and the HIP kernels.  Positions come from
tests/golden/reference_bounce.json (transcribed data): 16 positions with exhaustive target sets."""

import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    with open(os.path.join(golden_dir, "reference_bounce.json")) as fh:
        return json.load(fh)


def assert_state(pos, state=None):
    from simulator.game.bounce import Config

    grid = np.array(pos["grid"], dtype=np.int8)
    if state is None:
        assert pos["player"] == 0
        state = Config(grid).sample_initial_state()
    np.testing.assert_array_equal(grid, state.grid)
    assert pos["player"] == state.player
    selected = None
    if pos["source"] is not None:
        actions = {tuple(a.target): a for a in state.actions_at(np.array(pos["source"]))}
        assert {tuple(t) for t in pos["targets"]} == set(actions)  # exhaustive
        if pos["chosen"] is not None:
            selected = actions[tuple(pos["chosen"])]
    return state, selected


def test_reference_games(fx):
    names = []
    for test in fx["tests"]:
        state = None
        for pos in test["positions"]:
            state, action = assert_state(pos, state)
            assert tuple(action.source) == tuple(pos["source"])
            state = action.sample_next_state()
        assert state.has_ended == test["final"]["has_ended"], test["name"]
        assert len(state.actions) == test["final"]["n_actions"]
        assert state.reward.tolist() == test["final"]["reward"], test["name"]
        names.append(test["name"])
    assert names == ["test_small", "test_normal", "test_last_row", "test_larger_values", "test_block_victory", "test_draw"]


def test_json(fx):
    from simulator.game.bounce import Action, Config, State

    j = fx["json"]
    state, action = assert_state(j["position"])
    config = state.config
    assert config.to_json() == j["config"]
    assert Config.from_json(config.to_json()) == config
    assert state.to_json() == j["state"]
    assert State.from_json(state.to_json(), config) == state
    assert action.to_json() == j["action"]
    assert Action.from_json(action.to_json(), state) == action


def test_surface(fx):
    from simulator.game.bounce import Action, Config, State

    grid = np.array(fx["tests"][1]["positions"][0]["grid"])  # int64, as textual/bounce.py:66-79 passes it
    config = Config(grid)
    assert Config.num_players == 2 and Config.State is State and State.Action is Action
    assert config.grid.dtype == np.int8 and config.grid.shape == (9, 6)
    state = config.sample_initial_state()
    assert state.config is config and state.grid.dtype == np.int8
    assert state.reward.tolist() == [0, 0]
    a = state.action_at(np.array([0, 1]), np.array([0, 2]))
    assert a.source.tolist() == [0, 1] and a.target.tolist() == [0, 2] and a.state is state
    assert state.actions_at(np.array([0, 7])) == []      # not the active row
    assert state.actions_at(np.array([3, 4])) == []      # empty cell
    with pytest.raises(RuntimeError):
        state.action_at(np.array([0, 1]), np.array([0, 3]))
    with pytest.raises(RuntimeError):
        state.actions_at(np.array([6, 1]))               # outside the board
    with pytest.raises(TypeError):
        Config(np.zeros(5, dtype=np.int8))               # wrong rank (tensor.hpp:45-59)
    # canonical order of state.actions: sources by x, targets by (y, x); no duplicates
    listed = [(tuple(x.source), tuple(x.target)) for x in state.actions]
    assert listed == sorted(set(listed), key=lambda st: (st[0][0], st[1][1], st[1][0]))
    nxt = a.sample_next_state()
    assert state.grid[1, 0] == 1 and nxt.grid[1, 0] == 0 and nxt.grid[2, 0] == 1 and nxt.player == 1
    assert nxt == state.action_at(np.array([0, 1]), np.array([0, 2])).sample_next_state()
    assert hash(nxt) == hash(a.sample_next_state()) and nxt != state


def _crowded_grids():
    """Start grids either side of the one-wave transition kernel's reach (a piece per lane, at most 16): 16 pieces on 7x8
    and 18 on 6x8 (served by the thread-per-board code inside the same kernel), values up to 7."""
    sixteen = np.zeros((7, 8), dtype=np.int8)
    sixteen[1] = [1, 2, 3, 1, 2, 3, 1, 2]
    sixteen[5] = [2, 1, 3, 2, 1, 7, 2, 1]
    eighteen = np.zeros((6, 8), dtype=np.int8)
    eighteen[1] = [1, 2, 3, 1, 2, 3, 1, 2]
    eighteen[2, 3] = eighteen[3, 4] = 2
    eighteen[4] = [2, 1, 3, 2, 1, 3, 2, 1]
    return [sixteen, eighteen]


def test_random_playthrough_matches_oracle(fx):
    """Random games through the object API against the oracle, ply by ply: the action LIST (order included), the grid, the
    player, the end and the reward -- on the reference's default start and on two crowded ones."""
    from oracle import oracle
    from simulator.game.bounce import Config

    grid = np.array(fx["tests"][1]["positions"][0]["grid"], dtype=np.int8)
    rnd = random.Random(3)
    for start, games in ((grid, 2), *((g, 3) for g in _crowded_grids())):
        for _ in range(games):
            state = Config(start).sample_initial_state()
            orc = oracle.BounceOracle(start, 1)
            plies = 0
            while not state.has_ended and plies < 300:
                assert state.player == orc.player[0]
                np.testing.assert_array_equal(state.grid, orc.grid[0])
                listed = [(tuple(a.source), tuple(a.target)) for a in state.actions]
                assert listed == orc.actions(0)
                action = rnd.choice(state.actions)
                state = action.sample_next_state()
                orc.step_actions([[*action.source, *action.target]])
                plies += 1
            assert bool(orc.ended[0]) == state.has_ended
            np.testing.assert_array_equal(state.reward, orc.reward[0])


def test_json_of_terminal_states(fx):
    from simulator.game.bounce import Action, Config, State

    test = fx["tests"][0]  # test_small: player 1 reaches the bottom row
    state = None
    for pos in test["positions"]:
        state, action = assert_state(pos, state)
        state = action.sample_next_state()
    assert state.has_ended and state.reward.tolist() == [-1, 1]
    j = state.to_json()
    assert j["winner"] == 1
    back = State.from_json(j, state.config)
    assert back == state and back.has_ended and back.actions == [] and back.reward.tolist() == [-1, 1]
    with pytest.raises(RuntimeError):
        Action.from_json({"source": [0, 1], "target": [0, 2]}, back)
    with pytest.raises(RuntimeError):
        State.from_json({"grid": [[0, 0, 0]], "player": 0, "winner": -1}, state.config)


def test_branching_from_one_state_reloads_the_board(fx):
    """As in test_dropin_connect: actions taken from an older state, in both orders, against the oracle."""
    from oracle import oracle
    from simulator.game.bounce import Config

    grid = np.array(fx["tests"][1]["positions"][0]["grid"], dtype=np.int8)
    rnd = random.Random(5)
    state, history = Config(grid).sample_initial_state(), []
    for _ in range(6):
        if state.has_ended:
            break
        actions = state.actions
        sample = actions if len(actions) <= 6 else rnd.sample(actions, 6)
        children = {}
        for order in (sample, sample[::-1]):
            for action in order:
                move = [*action.source, *action.target]
                orc = oracle.BounceOracle(grid, 1)
                for past in history + [move]:
                    orc.step_actions([past])
                child = action.sample_next_state()
                np.testing.assert_array_equal(child.grid, orc.grid[0])
                assert child.has_ended == bool(orc.ended[0])
                assert child.player == orc.player[0] or child.has_ended
                np.testing.assert_array_equal(child.reward, orc.reward[0])
                if not child.has_ended:
                    assert [(tuple(a.source), tuple(a.target)) for a in child.actions] == orc.actions(0)
                assert children.setdefault(tuple(move), child) == child
        move = rnd.choice(sorted(children))
        history.append(list(move))
        state = children[move]
