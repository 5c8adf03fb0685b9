"""N3: the terminal widgets on the HIP-backed game objects, driven headless (textual's pilot): key presses become
`action_at` / `actions_at` calls, illegal picks (RuntimeError in the game objects) are ignored, chosen actions are
applied with `sample_next_state` -- the conventions of the reference UI (textual/connect.py:111-119,
textual/bounce.py:118-162, examples/arena.py:61-69)."""

import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

textual = pytest.importorskip("textual")


def test_widgets_play_through_the_drop_in_objects():
    from simulator.game import connect
    from simulator.textual import BounceBoard, ConnectBoard, DemoApp

    async def scenario():
        app = DemoApp(connect_config=connect.Config(2, 3, 2))
        async with app.run_test(size=(80, 24)) as pilot:
            board = app.query_one(ConnectBoard)
            board.focus()
            await pilot.pause()
            # the reference's own small game (tests/test_connect.py:68-115): columns 1, 1, 2 -> player 0 wins
            await pilot.press("right", "enter")
            await pilot.pause()
            assert board.state.grid[0, 1] == 0 and board.state.player == 1
            await pilot.press("enter")           # column 1 again
            await pilot.pause()
            assert board.state.grid[1, 1] == 1
            await pilot.press("enter")           # column 1 is full now: RuntimeError inside, nothing happens
            await pilot.pause()
            assert board.state.player == 0 and not board.state.has_ended
            await pilot.press("right", "enter")  # column 2
            await pilot.pause()
            assert board.state.has_ended and list(board.state.reward) == [1, -1]
            await pilot.press("enter")           # finished game: ignored
            await pilot.press("r")               # new game
            await pilot.pause()
            assert not board.state.has_ended and (board.state.grid == -1).all()
            assert "player 0" in board.render().plain

            bb = app.query_one(BounceBoard)
            bb.focus()
            await pilot.pause()
            await pilot.press("enter")           # (0, 0) is an empty goal cell: nothing to pick up
            assert bb.source is None
            await pilot.press("up", "enter")     # the piece "1" at (0, 1)
            await pilot.pause()
            assert bb.source == (0, 1)
            legal = {tuple(int(v) for v in a.target) for a in bb.state.actions_at(np.array([0, 1]))}
            from oracle import oracle
            from simulator.textual.boards import DEFAULT_BOUNCE_GRID

            assert legal == oracle.BounceOracle(DEFAULT_BOUNCE_GRID.astype(np.int8), 1).targets(0, 0, 1) and (0, 2) in legal
            await pilot.press("up", "enter")     # move it there
            await pilot.pause()
            assert bb.state.grid[2, 0] == 1 and bb.state.grid[1, 0] == 0 and bb.state.player == 1 and bb.source is None
            assert "player 1" in bb.render().plain

    asyncio.run(scenario())
