"""Value semantics shared by Config / State / Action objects.

The reference binds ==, !=, <, <=, >, >= and __hash__ by VALUE for all three classes
(src/simulator/game/helper.hpp:10-25); objects are immutable (README.md:67)."""

from __future__ import annotations

import functools


@functools.total_ordering
class ValueObject:
    __slots__ = ()

    def _key(self):  # pragma: no cover - overridden
        raise NotImplementedError

    def __eq__(self, other):
        if type(other) is not type(self):
            return NotImplemented
        return self._key() == other._key()

    def __lt__(self, other):
        if type(other) is not type(self):
            return NotImplemented
        return self._key() < other._key()

    def __hash__(self):
        return hash(self._key())
