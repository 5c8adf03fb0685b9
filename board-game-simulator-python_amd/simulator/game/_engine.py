"""What the one-board engines of `connect.py` and `bounce.py` share: a HIP stream of the engine's own, released with it, and
a bounded per-thread cache (a caller that walks through thousands of different Bounce start grids must not keep a device
batch, a stream and two page-locked blocks alive for each of them)."""

from __future__ import annotations

import ctypes
import threading
from collections import OrderedDict

from . import _abi

MAX_ENGINES_PER_THREAD = 32


class EngineBase:
    """A one-board batch on a stream of its own.  Subclasses set `self.batch` before calling `_own_stream()`."""

    batch = None
    _stream = None
    _device = 0

    def _own_stream(self, device: int) -> None:
        stream = ctypes.c_void_p()
        _abi.check(_abi.lib().bgs_stream_create(device, ctypes.byref(stream)))
        self._stream, self._device = stream.value, device
        self.batch.set_stream(stream.value)

    def close(self) -> None:
        batch, self.batch = self.batch, None
        if batch is not None:
            batch.close()  # (synchronises the stream before the batch's buffers go)
        stream, self._stream = self._stream, None
        if stream:
            _abi.lib().bgs_stream_destroy(self._device, ctypes.c_void_p(stream))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EngineCache:
    """Per calling thread: key -> engine, least recently used out beyond MAX_ENGINES_PER_THREAD."""

    def __init__(self):
        self._local = threading.local()

    def get(self, key, make):
        engines = self._local.__dict__.setdefault("engines", OrderedDict())
        eng = engines.get(key)
        if eng is None:
            eng = engines[key] = make()
            while len(engines) > MAX_ENGINES_PER_THREAD:
                _, old = engines.popitem(last=False)
                old.close()
        else:
            engines.move_to_end(key)
        return eng
