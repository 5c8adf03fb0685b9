#!/usr/bin/env python3
"""Transcribe the reference's test positions into data fixtures (tests/golden/reference_*.json).

The reference package cannot be imported here (its game core is an un-vendored CMake FetchContent dependency and
nanobind is absent), so its own tests are the only executable statement of the rules.  This script READS the two
test files as text (never imports or executes them), walks their syntax trees, pulls out every ASCII board passed
to ``assert_state`` plus the literal rewards / JSON dictionaries they are compared with, and writes plain data:
grids in the reference layout (row 0 = bottom), the side to move, the selected column / piece, the exhaustive
target set and the chosen target.  No reference source text is stored in the fixtures.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_reference_fixtures.py
"""

from __future__ import annotations

import ast
import json
import os
import sys

REFERENCE = os.environ.get("BGS_REFERENCE", "/root/reference")
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ board notations -> data


def connect_board(text: str):
    """'.' empty (-1), 'O' player 0, 'X' player 1; printed top row first; optional 'v' line marks a column."""
    lines = [ln for ln in text.split("\n") if ln.strip()]
    indent = min(len(ln) - len(ln.lstrip(" ")) for ln in lines)
    column = None
    if "v" in lines[0]:
        column = (lines[0].index("v") - indent) // 2
        lines = lines[1:]
    code = {".": -1, "O": 0, "X": 1}
    grid = [[code[ch] for ch in ln.split()] for ln in reversed(lines)]
    stones = [sum(row.count(p) for row in grid) for p in (0, 1)]
    player = 0 if stones[0] == stones[1] else 1
    return {"grid": grid, "player": player, "column": column}


def bounce_board(text: str):
    """Two characters per cell: piece ('.', digit, or '*' = empty legal target) + marker (']' = selected)."""
    rows = []
    for ln in text.split("\n"):
        ln = ln.strip()
        if not ln:
            continue
        if ln.startswith("["):
            ln = ln[1:]
        ln += " "
        rows.append([(ln[i], ln[i + 1]) for i in range(0, len(ln) - 1, 2)])
    rows.reverse()
    height, width = len(rows), len(rows[0])
    grid = [[0] * width for _ in range(height)]
    targets, selected = [], []
    for y, row in enumerate(rows):
        assert len(row) == width
        for x, (piece, marker) in enumerate(row):
            if piece.isdigit():
                grid[y][x] = int(piece)
            if piece == "*":
                targets.append([x, y])
            if marker == "]":
                selected.append([x, y])
    source = next((s for s in selected if s not in targets), None)
    chosen = next((s for s in selected if s in targets), None)
    player = 0
    if source is not None and any(v > 0 for row in grid[: source[1]] for v in row):
        player = 1
    return {"grid": grid, "player": player, "source": source, "targets": sorted(targets, key=lambda t: (t[1], t[0])), "chosen": chosen}


# ------------------------------------------------------------------ syntax-tree helpers


def functions(path: str):
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read())
    return {node.name: node for node in tree.body if isinstance(node, ast.FunctionDef) and node.name.startswith("test_")}


def boards_in(fn: ast.FunctionDef):
    out = []
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == "assert_state":
            out.append((node.lineno, node.args[0].value))
    return [text for _, text in sorted(out)]


def literals_compared(fn: ast.FunctionDef):
    """Every literal list / dict that an `assert a == <literal>` or `assert_array_equal(a, <literal>)` compares against."""
    found = []
    for node in ast.walk(fn):
        lit = None
        if isinstance(node, ast.Compare) and isinstance(node.comparators[0], (ast.Dict, ast.List)):
            lit = node.comparators[0]
        if isinstance(node, ast.Call) and getattr(node.func, "attr", None) == "assert_array_equal":
            if isinstance(node.args[1], (ast.Dict, ast.List)):
                lit = node.args[1]
        if lit is not None:
            found.append((lit.lineno, ast.literal_eval(lit)))
    return [v for _, v in sorted(found, key=lambda t: t[0])]


def config_args(fn: ast.FunctionDef):
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == "Config":
            return [ast.literal_eval(a) for a in node.args]
    return None


# ------------------------------------------------------------------ main


def main() -> int:
    connect_src = os.path.join(REFERENCE, "tests", "test_connect.py")
    bounce_src = os.path.join(REFERENCE, "tests", "test_bounce.py")
    if not (os.path.exists(connect_src) and os.path.exists(bounce_src)):
        print("reference tests not found under", REFERENCE, file=sys.stderr)
        return 1

    # ---- Connect
    fns = functions(connect_src)
    small = fns["test_small"]
    positions = [connect_board(t) for t in boards_in(small)]
    [reward] = [v for v in literals_compared(small) if isinstance(v, list)]
    cfg_json, state_json, action_json = [v for v in literals_compared(fns["test_json"]) if isinstance(v, dict)]
    connect = {
        "origin": "reference tests/test_connect.py (test_small :68-115, test_json :118-145), transcribed as data",
        "games": [
            {
                "name": "test_small",
                "config": config_args(small),
                "positions": positions,
                "reward": reward,
            }
        ],
        "json": {
            "config_args": config_args(fns["test_json"]),
            "config": cfg_json,
            "state_after_columns": [0],
            "state": state_json,
            "action_column": 1,
            "action": action_json,
        },
    }
    with open(os.path.join(OUT_DIR, "reference_connect.json"), "w") as fh:
        json.dump(connect, fh, indent=1)
        fh.write("\n")

    # ---- Bounce
    fns = functions(bounce_src)
    tests = []
    for name, fn in fns.items():
        if name == "test_json":
            continue
        boards = [bounce_board(t) for t in boards_in(fn)]
        [reward] = [v for v in literals_compared(fn) if isinstance(v, list)]
        tests.append({"name": name, "positions": boards, "final": {"has_ended": True, "n_actions": 0, "reward": reward}})
    jfn = fns["test_json"]
    [jboard] = [bounce_board(t) for t in boards_in(jfn)]
    cfg_json, state_json, action_json = [v for v in literals_compared(jfn) if isinstance(v, dict)]
    bounce = {
        "origin": "reference tests/test_bounce.py (:92-362 scripted games, :365-410 test_json), transcribed as data",
        "tests": tests,
        "json": {"position": jboard, "config": cfg_json, "state": state_json, "action": action_json},
    }
    with open(os.path.join(OUT_DIR, "reference_bounce.json"), "w") as fh:
        json.dump(bounce, fh, indent=1)
        fh.write("\n")

    n_pos = sum(len(t["positions"]) for t in tests)
    print(f"connect: {len(positions)} positions; bounce: {len(tests)} tests, {n_pos} positions")
    return 0


if __name__ == "__main__":
    sys.exit(main())
