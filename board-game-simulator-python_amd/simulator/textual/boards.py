"""`ConnectBoard`, `BounceBoard` and a two-board demo application (python -m simulator.textual.boards)."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
from rich.text import Text
from textual.app import App, ComposeResult
from textual.message import Message
from textual.reactive import reactive
from textual.widget import Widget

from ..game import bounce, connect


class _Board(Widget, can_focus=True):
    """What both boards share: a `state`, a cursor, `Reset` / `Selected` messages."""

    DEFAULT_CSS = "_Board { width: auto; height: auto; padding: 1; }"
    BINDINGS = [
        ("r,backspace", "reset", "New game"),
        ("enter,space", "select", "Select"),
        ("left", "move(-1, 0)", "Left"),
        ("right", "move(1, 0)", "Right"),
        ("up", "move(0, 1)", "Up"),
        ("down", "move(0, -1)", "Down"),
    ]

    state = reactive(None, layout=True)
    cursor: reactive[Tuple[int, int]] = reactive((0, 0))  # (x, y), y = 0 at the bottom row

    class Reset(Message):
        def __init__(self, board: "_Board") -> None:
            self.board = board
            super().__init__()

    class Selected(Message):
        def __init__(self, board: "_Board", action) -> None:
            self.board = board
            self.action = action
            super().__init__()

    def __init__(self, state=None, **kwargs) -> None:
        super().__init__(**kwargs)
        self.state = state

    def _shape(self) -> Tuple[int, int]:
        h, w = self.state.grid.shape
        return int(h), int(w)

    def action_reset(self) -> None:
        self.post_message(self.Reset(self))

    def action_move(self, dx: int, dy: int) -> None:
        if self.state is not None:
            h, w = self._shape()
            self.cursor = ((self.cursor[0] + dx) % w, (self.cursor[1] + dy) % h)

    def action_select(self) -> None:  # pragma: no cover - overridden
        raise NotImplementedError


class ConnectBoard(_Board):
    """A Connect-k position; the cursor is a column, `Select` drops a stone there."""

    MARKS = {-1: ("·", "grey50"), 0: ("O", "bold blue"), 1: ("X", "bold red")}

    def action_select(self) -> None:
        if self.state is None:
            return
        try:
            action = self.state.action_at(self.cursor[0])
        except RuntimeError:  # full column or finished game: nothing happens
            return
        self.post_message(self.Selected(self, action))

    def render(self) -> Text:
        text = Text()
        if self.state is None:
            return text
        grid = self.state.grid
        h, w = grid.shape
        for y in range(h - 1, -1, -1):
            for x in range(w):
                mark, style = self.MARKS[int(grid[y, x])]
                text.append(f" {mark}", style=f"{style} reverse" if x == self.cursor[0] and self.has_focus else style)
            text.append("\n")
        text.append("ended: reward %s" % list(self.state.reward) if self.state.has_ended else f"player {self.state.player} to move")
        return text


class BounceBoard(_Board):
    """A Bounce position; `Select` on a movable piece picks it up (its legal targets are highlighted), `Select` on a
    highlighted cell moves it there, anything else puts it down again."""

    source: reactive[Optional[Tuple[int, int]]] = reactive(None)

    def watch_state(self, old, new) -> None:
        self.source = None

    def _targets(self):
        if self.state is None or self.source is None:
            return {}
        return {tuple(int(v) for v in a.target): a for a in self.state.actions_at(np.array(self.source))}

    def action_select(self) -> None:
        if self.state is None:
            return
        if self.source is not None:
            action = self._targets().get(self.cursor)
            self.source = None
            if action is not None:
                self.post_message(self.Selected(self, action))
            return
        try:
            if self.state.actions_at(np.array(self.cursor)):
                self.source = self.cursor
        except RuntimeError:
            pass

    def render(self) -> Text:
        text = Text()
        if self.state is None:
            return text
        grid = self.state.grid
        h, w = grid.shape
        targets = self._targets()
        for y in range(h - 1, -1, -1):
            for x in range(w):
                v = int(grid[y, x])
                mark = "·" if v == 0 else (str(v) if v < 10 else "#")
                style = "grey50" if v == 0 else "bold"
                if (x, y) in targets:
                    style += " on green"
                if (x, y) == self.source:
                    style += " underline"
                if (x, y) == self.cursor and self.has_focus:
                    style += " reverse"
                text.append(f" {mark}", style=style)
            text.append("\n")
        text.append("ended: reward %s" % list(self.state.reward) if self.state.has_ended else f"player {self.state.player} to move")
        return text


DEFAULT_BOUNCE_GRID = np.zeros((9, 6), dtype=np.int64)
DEFAULT_BOUNCE_GRID[1] = DEFAULT_BOUNCE_GRID[7] = [1, 2, 3, 3, 2, 1]


class DemoApp(App):
    """Both boards side by side; tab switches focus."""

    CSS = "Screen { layout: horizontal; }"

    def __init__(self, connect_config=None, bounce_config=None) -> None:
        super().__init__()
        self.connect_config = connect_config or connect.Config(6, 7, 4)
        self.bounce_config = bounce_config or bounce.Config(DEFAULT_BOUNCE_GRID)

    def compose(self) -> ComposeResult:
        yield ConnectBoard(self.connect_config.sample_initial_state(), id="connect")
        yield BounceBoard(self.bounce_config.sample_initial_state(), id="bounce")

    def on__board_reset(self, event: _Board.Reset) -> None:
        config = self.connect_config if isinstance(event.board, ConnectBoard) else self.bounce_config
        event.board.state = config.sample_initial_state()

    def on__board_selected(self, event: _Board.Selected) -> None:
        event.board.state = event.action.sample_next_state()


if __name__ == "__main__":
    DemoApp().run()
