"""Terminal widgets over the drop-in `simulator.game` objects.

A small, independent implementation of what the reference's `simulator.textual` package offers its users: a board
widget per game whose cursor turns into `state.action_at(...)` calls -- illegal picks raise `RuntimeError` in the game
objects and are simply ignored here, as the reference UI does (textual/connect.py:111-119, textual/bounce.py:118-162)
-- and which reports the chosen `Action` to the application, which applies it with `action.sample_next_state()`.
Every rule is evaluated by the HIP kernels behind the game objects; the widgets only draw and forward."""

from .boards import BounceBoard, ConnectBoard, DemoApp  # noqa: F401
