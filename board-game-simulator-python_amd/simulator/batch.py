"""Batched boards on one MI355X: the host-side mirror of the reference's Config/State/Action loop, N boards a call.

The reference loop (README.md:45-72)::

    state = config.sample_initial_state()
    while not state.has_ended:
        action = random.choice(state.actions)
        state = action.sample_next_state()
    reward = state.reward

becomes ``batch = ConnectBatch(6, 7, 4, n); batch.rollout(seed); batch.reward`` -- one HIP launch.  Every method is
a thin call into libbgs.so (include/bgs.h); arrays come back in the reference layout (``grid`` int8[n, H, W], row 0 =
bottom row).  PyTorch, when present with a GPU, only provides the device arena and the stream (plumbing).
"""

from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from .game import _abi

DEFAULT_SEED = 0x0123456789ABCDEF


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def _torch_cuda():
    try:
        import torch
    except ImportError:  # torch is optional: the library then owns its device memory
        return None
    return torch if torch.cuda.is_available() else None


class PinnedArray:
    """A page-locked host array (bgs_host_alloc) viewed as a numpy array: the destination of asynchronous device ->
    host copies.  Keep the object alive while copies into it may be in flight."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 1)
        p = ctypes.c_void_p()
        _abi.check(_abi.lib().bgs_host_alloc(nbytes, ctypes.byref(p)))
        self._ptr = p
        buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    @property
    def ptr(self) -> int:
        return self._ptr.value

    def close(self) -> None:
        if self._ptr:
            self.array = None
            _abi.lib().bgs_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostEvent:
    """A device event the asynchronous hand-over records behind its copy (bgs_event)."""

    def __init__(self, device: int = 0):
        self._handle = _abi.c_handle()
        _abi.check(_abi.lib().bgs_event_create(int(device), ctypes.byref(self._handle)))

    def synchronize(self) -> None:
        _abi.check(_abi.lib().bgs_event_synchronize(self._handle))

    def done(self) -> bool:
        flag = ctypes.c_int(0)
        _abi.check(_abi.lib().bgs_event_query(self._handle, ctypes.byref(flag)))
        return bool(flag.value)

    def close(self) -> None:
        if self._handle:
            _abi.lib().bgs_event_destroy(self._handle)
            self._handle = _abi.c_handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _host_ptr(dst) -> int:
    """Address of a host destination: PinnedArray, numpy array or (pinned) CPU torch tensor."""
    if isinstance(dst, PinnedArray):
        return dst.ptr
    if isinstance(dst, np.ndarray):
        if not dst.flags.c_contiguous:
            raise TypeError("host destination must be C-contiguous")
        return dst.ctypes.data
    if hasattr(dst, "data_ptr") and not getattr(dst, "is_cuda", False):
        return dst.data_ptr()
    raise TypeError("host destination must be a PinnedArray, a numpy array or a CPU torch tensor")


def expand_outcomes_host(packed, n: int, out=None, first: int = 0, count: Optional[int] = None) -> np.ndarray:
    """Packed 2-bit outcome codes on the HOST (uint8[(n + 3) // 4]) -> reward int8[n, 2] (bgs_expand_outcomes_host)."""
    if out is None:
        out = np.empty((n, 2), dtype=np.int8)
    cnt = n - first if count is None else count
    _abi.check(_abi.lib().bgs_expand_outcomes_host(ctypes.c_void_p(_host_ptr(packed)), first, cnt, ctypes.c_void_p(_host_ptr(out))))
    return out


def _reward_destination(dst, n_games: int, per_game: int = 2) -> int:
    """Address of a host array the library's worker threads will fill with int8[n_games, per_game] LATER (rewards: 2
    bytes per game; grids: height * width): it must be large enough, of 1-byte items and writeable -- the C side cannot
    check any of that from a bare pointer."""
    if per_game != 2:
        return _reward_destination(dst, (n_games * per_game + 1) // 2)
    if isinstance(dst, PinnedArray):
        arr = dst.array
    elif isinstance(dst, np.ndarray):
        arr = dst
    elif hasattr(dst, "data_ptr") and not getattr(dst, "is_cuda", False):
        if dst.element_size() != 1 or dst.numel() < 2 * n_games or not dst.is_contiguous():
            raise TypeError(f"host reward destination must be a contiguous 1-byte-item tensor of at least {2 * n_games} elements")
        return dst.data_ptr()
    else:
        raise TypeError("host reward destination must be a PinnedArray, a numpy array or a CPU torch tensor")
    if arr is None:
        raise ValueError("host reward destination has been closed")
    if arr.dtype.itemsize != 1:
        raise TypeError(f"host reward destination must have 1-byte items (int8), got {arr.dtype}")
    if not arr.flags.c_contiguous:
        raise TypeError("host reward destination must be C-contiguous")
    if not arr.flags.writeable:
        raise ValueError("host reward destination is read-only")
    if arr.nbytes < 2 * n_games:
        raise ValueError(f"host reward destination holds {arr.nbytes} bytes, int8[{n_games}, 2] needs {2 * n_games}")
    return arr.ctypes.data


class RewardSink:
    """Delivers the rewards of successive batch steps into host arrays int8[n, 2] while the GPU goes on playing
    (bgs_sink_*): outcome codes cross PCIe into pinned slots, worker threads expand them on arrival.

    The destination of a submission is written by the library's worker threads after the call has returned: the sink
    checks its size / item size / writeability up front and keeps a reference to it until the submission has been
    waited for (or the sink is closed), so dropping the array early cannot turn into a write to freed memory."""

    def __init__(self, max_games: int, slots: int = 4, threads: int = 4, device: int = 0):
        self._handle = _abi.c_handle()
        self._alive = {}  # ticket -> destination (and anything else that must outlive the delivery)
        self.max_games = int(max_games)
        _abi.check(_abi.lib().bgs_sink_create(int(device), int(max_games), int(slots), int(threads), ctypes.byref(self._handle)))

    def _destination(self, host_reward, n_games: int) -> int:
        return _reward_destination(host_reward, n_games)

    def _keep(self, ticket: int, *objects) -> int:
        if len(self._alive) >= 64:  # (callers that never wait: forget what has been delivered, now and then)
            done = self.completed
            for t in [t for t in self._alive if t < done]:
                del self._alive[t]
        self._alive[ticket] = objects
        return ticket

    def submit(self, batch: "_Batch", host_reward) -> int:
        ticket = ctypes.c_int64(-1)
        ptr = self._destination(host_reward, batch.n)
        try:
            _abi.check(_abi.lib().bgs_sink_submit(self._handle, batch._handle, ctypes.c_void_p(ptr), ctypes.byref(ticket)))
        finally:
            if ticket.value >= 0:  # the ticket exists even when the enqueue failed
                self._keep(ticket.value, host_reward, batch)
        return ticket.value

    def rollout(self, batch: "_Batch", host_reward, seed: int = DEFAULT_SEED, max_plies: int = 2**31 - 1,
                from_initial: bool = False) -> int:
        """`batch.rollout(...)` + `submit(batch, host_reward)` in one library call (bgs_sink_rollout)."""
        flags = _abi.ROLLOUT_FROM_INITIAL if from_initial else 0
        ticket = ctypes.c_int64(-1)
        ptr = self._destination(host_reward, batch.n)
        try:
            _abi.check(
                _abi.lib().bgs_sink_rollout(
                    self._handle, batch._handle, ctypes.c_uint64(seed), ctypes.c_int32(max_plies), ctypes.c_uint32(flags),
                    ctypes.c_void_p(ptr), ctypes.byref(ticket),
                )
            )
        finally:
            if ticket.value >= 0:
                self._keep(ticket.value, host_reward, batch)
        return ticket.value

    def submit_packed(self, device_packed, n_games: int, host_reward, stream: int = 0) -> int:
        """`device_packed`: CUDA uint8 tensor (or device address) of the codes of n_games games, e.g. the RCCL-gathered
        codes of all ranks; copied on HIP stream `stream`."""
        ptr = device_packed.data_ptr() if hasattr(device_packed, "data_ptr") else int(device_packed)
        if hasattr(device_packed, "numel") and device_packed.numel() * device_packed.element_size() < (int(n_games) + 3) // 4:
            raise ValueError("device_packed is smaller than the codes of n_games games")
        ticket = ctypes.c_int64(-1)
        dst = _reward_destination(host_reward, int(n_games))
        try:
            _abi.check(
                _abi.lib().bgs_sink_submit_packed(
                    self._handle, ctypes.c_void_p(stream), ctypes.c_void_p(ptr), int(n_games), ctypes.c_void_p(dst), ctypes.byref(ticket)
                )
            )
        finally:
            if ticket.value >= 0:
                self._keep(ticket.value, host_reward, device_packed)
        return ticket.value

    def wait(self, ticket: int) -> None:
        _abi.check(_abi.lib().bgs_sink_wait(self._handle, int(ticket)))
        self._alive.pop(int(ticket), None)

    @property
    def completed(self) -> int:
        """Number of submissions whose rewards are in their host arrays (they complete in ticket order)."""
        if not self._handle:
            return 0
        v = ctypes.c_int64(0)
        _abi.check(_abi.lib().bgs_sink_completed(self._handle, ctypes.byref(v)))
        return v.value

    def set_progress(self, word_address: Optional[int], keepalive=None) -> None:
        """Announce `completed` in the int64 at `word_address` (host memory, e.g. a SharedRewardRing progress word) after
        every delivery; sleepers use `bgs_progress_wait`.  None stops it."""
        _abi.check(_abi.lib().bgs_sink_set_progress(self._handle, ctypes.c_void_p(word_address) if word_address else None))
        self._progress_keepalive = keepalive

    def close(self) -> None:
        if self._handle:
            _abi.lib().bgs_sink_destroy(self._handle)  # waits for every claimed submission before the workers stop
            self._handle = _abi.c_handle()
        self._alive.clear()
        self._progress_keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GridSink(RewardSink):
    """Delivers the BOARDS of successive batch steps -- `state.grid` of every game, int8[n, H, W] in the reference layout --
    into host arrays while the GPU goes on playing (bgs_grid_sink_create): the boards cross PCIe bit-packed (16 B per
    6x7 Connect board instead of 42), worker threads expand them.  `submit` / `rollout` / `wait` / `completed` as for a
    RewardSink, with int8[n, H, W] destinations; a `RolloutExecutor` takes one as its sink."""

    def __init__(self, like: "_Batch", slots: int = 4, threads: int = 4):
        self._handle = _abi.c_handle()
        self._alive = {}
        self.max_games = like.n
        self._cells = like.height * like.width
        _abi.check(_abi.lib().bgs_grid_sink_create(like._handle, int(slots), int(threads), ctypes.byref(self._handle)))

    def _destination(self, host_grid, n_games: int) -> int:
        return _reward_destination(host_grid, n_games, self._cells)

    def submit_packed(self, *args, **kwargs):
        raise TypeError("a grid sink takes boards, not outcome codes")


class DeviceView:
    """A batch buffer as a `__cuda_array_interface__` object (version 2; ROCm frameworks use the same protocol): a
    zero-copy hand-over to torch (`torch.as_tensor(view, device="cuda")`), CuPy or Numba without importing any of
    them here.  From a torch tensor, DLPack is one call away (`tensor.__dlpack__()`)."""

    def __init__(self, owner, ptr: int, shape, typestr: str):
        self._owner = owner  # keeps the batch (and its arena) alive
        self.__cuda_array_interface__ = {
            "shape": tuple(int(x) for x in shape),
            "typestr": typestr,
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }


class _Batch:
    """Common part of ConnectBatch / BounceBatch: lifetime, stepping, observation."""

    game = 0

    def __init__(self, n: int, height: int, width: int, device: int, use_torch: Optional[bool]):
        self.n = int(n)
        self.height = int(height)
        self.width = int(width)
        self.device = int(device)
        self._handle = _abi.c_handle()
        self._arena = None
        self._torch = None
        if use_torch is None or use_torch:
            self._torch = _torch_cuda()
            if use_torch and self._torch is None:
                raise RuntimeError("use_torch=True but torch with a visible GPU is not available")

    # ---- lifetime -------------------------------------------------------------------------------
    def _make_arena(self, nbytes: int):
        if self._torch is None:
            return None, 0
        self._arena = self._torch.empty(nbytes, dtype=self._torch.uint8, device=f"cuda:{self.device}")
        return ctypes.c_void_p(self._arena.data_ptr()), nbytes

    def _after_create(self):
        nbytes, generic = ctypes.c_size_t(), ctypes.c_int()
        _abi.check(_abi.lib().bgs_legal_bytes(self._handle, ctypes.byref(nbytes), ctypes.byref(generic)))
        self._legal_bytes = nbytes.value   # bytes per board of bgs_transition's legal-move record
        self.generic = bool(generic.value)  # served by the generic kernels (geometry beyond the bit-packed limits)
        if self._torch is not None:
            self.set_stream(self._torch.cuda.current_stream(self.device).cuda_stream)

    def close(self) -> None:
        if self._handle:
            _abi.lib().bgs_destroy(self._handle)
            self._handle = _abi.c_handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int) -> None:
        _abi.check(_abi.lib().bgs_set_stream(self._handle, ctypes.c_void_p(hip_stream)))

    def set_first_game(self, first_game: int) -> None:
        """Global id of board 0: RNG streams are keyed by global game id, so shards reproduce the unsharded run."""
        _abi.check(_abi.lib().bgs_set_first_game(self._handle, ctypes.c_uint64(first_game)))

    def set_rng_contract(self, contract: str) -> None:
        """Connect: "per-block" (default: a philox word per block of four plies) or "per-ply" (strict: a word per ply, as
        Bounce draws -- one independent uniform choice per ply, what `random.choice` gives a caller of the reference,
        README.md:62).  Applies to every later random step / rollout of this batch (bgs_set_rng_contract)."""
        codes = {"per-block": _abi.RNG_PER_BLOCK, "per-ply": _abi.RNG_PER_PLY}
        if contract not in codes:
            raise ValueError(f"unknown RNG contract {contract!r}: 'per-block' or 'per-ply'")
        _abi.check(_abi.lib().bgs_set_rng_contract(self._handle, codes[contract]))

    def set_launches_in_flight(self, launches: int) -> None:
        """A hint: how many rollout launches the caller keeps in flight on the device (1, the default: one at a time).
        Results never depend on it; the Bounce rollout shapes its launch by it (bgs_set_launches_in_flight).  The
        rollout executor and `RolloutPipeline` pass their depth."""
        _abi.check(_abi.lib().bgs_set_launches_in_flight(self._handle, int(launches)))

    def synchronize(self) -> None:
        _abi.check(_abi.lib().bgs_synchronize(self._handle))

    # ---- the hot path ---------------------------------------------------------------------------
    def reset(self) -> None:
        _abi.check(_abi.lib().bgs_reset(self._handle))

    def step_random(self, seed: int = DEFAULT_SEED, plies: int = 1) -> None:
        """`plies` uniformly sampled plies on every running board (one launch where the kernel can keep the boards
        in registers in between)."""
        _abi.check(_abi.lib().bgs_step_random_n(self._handle, ctypes.c_uint64(seed), ctypes.c_int32(plies)))

    def rollout(self, seed: int = DEFAULT_SEED, max_plies: int = 2**31 - 1, from_initial: bool = False) -> None:
        flags = _abi.ROLLOUT_FROM_INITIAL if from_initial else 0
        _abi.check(_abi.lib().bgs_rollout(self._handle, ctypes.c_uint64(seed), ctypes.c_int32(max_plies), ctypes.c_uint32(flags)))

    def _step_actions(self, actions, per_board: int, want_status: bool):
        status = np.zeros(self.n, dtype=np.int32) if want_status else None
        sp = _ptr(status, ctypes.c_int32) if want_status else None
        if self._torch is not None and isinstance(actions, self._torch.Tensor):
            t = actions
            if not (t.is_cuda and t.dtype == self._torch.int32 and t.is_contiguous() and t.numel() == self.n * per_board):
                raise TypeError("device actions must be a contiguous int32 CUDA tensor of the batch's shape")
            _abi.check(_abi.lib().bgs_step_actions(self._handle, ctypes.c_void_p(t.data_ptr()), 1, sp))
        else:
            a = np.ascontiguousarray(actions, dtype=np.int32)
            if a.size != self.n * per_board:
                raise TypeError(f"expected {self.n * per_board} action entries, got {a.size}")
            _abi.check(_abi.lib().bgs_step_actions(self._handle, ctypes.c_void_p(a.ctypes.data), 0, sp))
        return status

    @property
    def steps(self) -> int:
        """env-steps applied since the last reset (transitions on running boards)."""
        v = ctypes.c_uint64(0)
        _abi.check(_abi.lib().bgs_steps(self._handle, ctypes.byref(v)))
        return v.value

    def reset_steps(self) -> None:
        _abi.check(_abi.lib().bgs_reset_steps(self._handle))

    # ---- observation (reference layout, host copies) ------------------------------------------------
    @property
    def grid(self) -> np.ndarray:
        out = np.empty((self.n, self.height, self.width), dtype=np.int8)
        _abi.check(_abi.lib().bgs_read_grid(self._handle, _ptr(out, ctypes.c_int8)))
        return out

    @property
    def player(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int8)
        _abi.check(_abi.lib().bgs_read_player(self._handle, _ptr(out, ctypes.c_int8)))
        return out

    @property
    def has_ended(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint8)
        _abi.check(_abi.lib().bgs_read_ended(self._handle, _ptr(out, ctypes.c_uint8)))
        return out.astype(bool)

    @property
    def winner(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int8)
        _abi.check(_abi.lib().bgs_read_winner(self._handle, _ptr(out, ctypes.c_int8)))
        return out

    @property
    def reward(self) -> np.ndarray:
        out = np.empty((self.n, 2), dtype=np.int8)
        _abi.check(_abi.lib().bgs_read_reward(self._handle, _ptr(out, ctypes.c_int8)))
        return out

    # ---- asynchronous hand-over to host memory (the batch form of State.reward, connect.cpp:41) ----------------
    def read_reward_async(self, host_dst, event: Optional[HostEvent] = None) -> None:
        """Enqueue reward int8[n, 2] -> `host_dst` (page-locked) on the batch's stream; `event` fires behind the copy."""
        _abi.check(_abi.lib().bgs_read_reward_async(self._handle, ctypes.c_void_p(_host_ptr(host_dst)), event._handle if event else None))

    def read_outcomes_async(self, host_dst, event: Optional[HostEvent] = None) -> None:
        """Enqueue the 2-bit outcome codes uint8[(n + 3) // 4] -> `host_dst` (page-locked); expand with
        `expand_outcomes_host`."""
        _abi.check(_abi.lib().bgs_read_outcomes_async(self._handle, ctypes.c_void_p(_host_ptr(host_dst)), event._handle if event else None))

    def rollout_to_host(self, host_dst, seed: int = DEFAULT_SEED, max_plies: int = 2**31 - 1, from_initial: bool = False,
                        codes: bool = False, event: Optional[HostEvent] = None) -> None:
        """`rollout` + `read_reward_async` (or `read_outcomes_async` with codes=True) in one library call."""
        flags = _abi.ROLLOUT_FROM_INITIAL if from_initial else 0
        _abi.check(
            _abi.lib().bgs_rollout_to_host(
                self._handle, ctypes.c_uint64(seed), ctypes.c_int32(max_plies), ctypes.c_uint32(flags),
                ctypes.c_void_p(_host_ptr(host_dst)), 1 if codes else 0, event._handle if event else None,
            )
        )

    @property
    def plies(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int32)
        _abi.check(_abi.lib().bgs_read_plies(self._handle, _ptr(out, ctypes.c_int32)))
        return out

    @property
    def action_count(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int32)
        _abi.check(_abi.lib().bgs_read_action_count(self._handle, _ptr(out, ctypes.c_int32)))
        return out

    def write_state(self, grid, player=None, winner=None, plies=None) -> np.ndarray:
        """Load boards in the reference layout; returns per-board status (0 ok, -1 malformed and left untouched)."""
        g = np.ascontiguousarray(grid, dtype=np.int8)
        if g.shape != (self.n, self.height, self.width):
            raise TypeError(f"grid must have shape {(self.n, self.height, self.width)}, got {g.shape}")
        p = None if player is None else np.ascontiguousarray(player, dtype=np.int8)
        w = None if winner is None else np.ascontiguousarray(winner, dtype=np.int8)
        l = None if plies is None else np.ascontiguousarray(plies, dtype=np.int32)
        status = np.zeros(self.n, dtype=np.int32)
        _abi.check(
            _abi.lib().bgs_write_state(
                self._handle,
                _ptr(g, ctypes.c_int8),
                None if p is None else _ptr(p, ctypes.c_int8),
                None if w is None else _ptr(w, ctypes.c_int8),
                None if l is None else _ptr(l, ctypes.c_int32),
                _ptr(status, ctypes.c_int32),
            )
        )
        return status

    def transition(self, grid=None, player=None, winner=None, plies=None, actions=None):
        """One round trip for the object API: optionally load boards, optionally apply one chosen move per board, then
        observe.  Returns (status int32[n], grid, player, winner, plies, legal, reward) with `legal` the Connect mask
        uint8[n, W] or the Bounce target masks uint64[n, W + 1] and `reward` int8[n, 2] as the device holds it."""
        n = self.n
        i8 = ctypes.c_int8
        g = p = w = l = a = None
        if grid is not None:
            g = np.ascontiguousarray(grid, dtype=np.int8)
            if g.shape != (n, self.height, self.width):
                raise TypeError(f"grid must have shape {(n, self.height, self.width)}, got {g.shape}")
            p = np.ascontiguousarray(player, dtype=np.int8)
            w = np.ascontiguousarray(winner, dtype=np.int8)
            l = None if plies is None else np.ascontiguousarray(plies, dtype=np.int32)
        if actions is not None:
            a = np.ascontiguousarray(actions, dtype=np.int32)
            if a.size != n * self._action_width:
                raise TypeError(f"expected {n * self._action_width} action entries, got {a.size}")
        status = np.zeros(n, dtype=np.int32)
        grid_out = np.empty((n, self.height, self.width), dtype=np.int8)
        player_out = np.empty(n, dtype=np.int8)
        winner_out = np.empty(n, dtype=np.int8)
        plies_out = np.empty(n, dtype=np.int32)
        legal_out = self._empty_legal()
        reward_out = np.empty((n, 2), dtype=np.int8)
        _abi.check(
            _abi.lib().bgs_transition(
                self._handle,
                None if g is None else _ptr(g, i8),
                None if p is None else _ptr(p, i8),
                None if w is None else _ptr(w, i8),
                None if l is None else _ptr(l, ctypes.c_int32),
                None if a is None else _ptr(a, ctypes.c_int32),
                _ptr(status, ctypes.c_int32),
                _ptr(grid_out, i8),
                _ptr(player_out, i8),
                _ptr(winner_out, i8),
                _ptr(plies_out, ctypes.c_int32),
                ctypes.c_void_p(legal_out.ctypes.data),
                _ptr(reward_out, i8),
            )
        )
        return status, grid_out, player_out, winner_out, plies_out, legal_out, reward_out

    def one_board_call(self) -> "OneBoardCall":
        return OneBoardCall(self)

    # ---- device-side hand-over (torch / RCCL plumbing) -----------------------------------------------
    def buffer(self, buffer_id: int):
        """(device pointer, bytes) of one of the batch's buffers (see bgs_buffer_id in include/bgs.h)."""
        p = ctypes.c_void_p()
        sz = ctypes.c_size_t()
        _abi.check(_abi.lib().bgs_buffer(self._handle, buffer_id, ctypes.byref(p), ctypes.byref(sz)))
        return p.value, sz.value

    def device_view(self, what: str = "reward") -> DeviceView:
        """`__cuda_array_interface__` view of a batch buffer without torch: "reward" int8[n, 2], "status" uint8[n],
        "planes" uint64[planes, n]."""
        if what == "reward":
            ptr, _ = self.buffer(_abi.BUF_REWARD)
            return DeviceView(self, ptr, (self.n, 2), "|i1")
        if what == "status":
            ptr, _ = self.buffer(_abi.BUF_STATUS)
            return DeviceView(self, ptr, (self.n,), "|u1")
        if what == "planes":
            ptr, nbytes = self.buffer(_abi.BUF_PLANES)
            return DeviceView(self, ptr, (nbytes // (8 * self.n), self.n), "<u8")
        raise ValueError(f"unknown device view {what!r}")

    def _arena_view(self, buffer_id: int):
        if self._arena is None:
            raise RuntimeError("device views need the torch-owned arena (construct the batch with torch + GPU available)")
        ptr, nbytes = self.buffer(buffer_id)
        off = ptr - self._arena.data_ptr()
        return self._arena[off : off + nbytes]

    def reward_tensor(self):
        """Zero-copy torch view int8[n, 2] of the device reward buffer (what the RCCL gather ships)."""
        return self._arena_view(_abi.BUF_REWARD).view(self._torch.int8).view(self.n, 2)

    def status_tensor(self):
        return self._arena_view(_abi.BUF_STATUS)

    def steps_tensor(self):
        """Device view of the sharded env-step counter (int64 words; their SUM is the count)."""
        return self._arena_view(_abi.BUF_STEPS).view(self._torch.int64)

    def outcomes_tensor(self, out=None):
        """uint8[(n + 3) // 4] on the device: 2-bit outcome codes (0 running, 1 / 2 winner, 3 draw), 4 boards per
        byte -- the compact form of the rewards that ranks exchange (see simulator.sharding)."""
        t = self._torch
        if t is None:
            raise RuntimeError("outcomes_tensor needs torch with a GPU")
        if out is None:
            out = t.empty((self.n + 3) // 4, dtype=t.uint8, device=f"cuda:{self.device}")
        _abi.check(_abi.lib().bgs_pack_outcomes(self._handle, ctypes.c_void_p(out.data_ptr())))
        return out

    def rollout_outcomes_tensor(self, out, seed: int = DEFAULT_SEED, max_plies: int = 2**31 - 1, from_initial: bool = False):
        """`rollout` + `outcomes_tensor(out)` in one library call (bgs_rollout_pack): the rollout kernel writes the codes
        itself where it can.  `out`: CUDA uint8 tensor of ((n + 63) // 64) * 16 bytes."""
        if out.numel() < (self.n + 63) // 64 * 16 or not out.is_cuda or not out.is_contiguous():
            raise TypeError("outcome buffer must be a contiguous CUDA uint8 tensor of ((n + 63) // 64) * 16 bytes")
        flags = _abi.ROLLOUT_FROM_INITIAL if from_initial else 0
        _abi.check(_abi.lib().bgs_rollout_pack(self._handle, ctypes.c_uint64(seed), ctypes.c_int32(max_plies),
                                               ctypes.c_uint32(flags), ctypes.c_void_p(out.data_ptr())))
        return out

    def grid_tensor(self, out=None):
        """Observation tensor int8[n, H, W] on the device (no host round trip), e.g. as policy-network input."""
        t = self._torch
        if t is None:
            raise RuntimeError("grid_tensor needs torch with a GPU")
        if out is None:
            out = t.empty((self.n, self.height, self.width), dtype=t.int8, device=f"cuda:{self.device}")
        _abi.check(_abi.lib().bgs_export_device(self._handle, ord("g"), ctypes.c_void_p(out.data_ptr())))
        return out


    def _export(self, what: str, out):
        _abi.check(_abi.lib().bgs_export_device(self._handle, ord(what), ctypes.c_void_p(out.data_ptr())))
        return out

    def action_count_tensor(self, out=None):
        """int32[n] on the device: len(state.actions) per board (0 once ended)."""
        t = self._need_torch("action_count_tensor")
        if out is None:
            out = t.empty(self.n, dtype=t.int32, device=f"cuda:{self.device}")
        return self._export("c", out)

    def reward_copy_tensor(self, out=None):
        """int8[n, 2] on the device: a copy of the reward buffer (bgs_export_device 'r')."""
        t = self._need_torch("reward_copy_tensor")
        if out is None:
            out = t.empty((self.n, 2), dtype=t.int8, device=f"cuda:{self.device}")
        return self._export("r", out)

    def env_step(self, actions, observation=None, ended=None, reward=None, status=None, auto_reset: bool = True):
        """The step of a vector environment (bgs_env_step): `step_actions_observe` plus `reward` int8[n, 2] -- the finished
        game's pair where `ended` is set, 0 / 0 while a game runs -- and, with auto_reset, boards that have ended put back to
        the initial state in the same call, so that the returned observation is already the new game's.  All device tensors,
        no synchronisation.  Returns the observation tensor."""
        return self._observe(actions, observation, ended, reward, status, _abi.ENV_AUTO_RESET if auto_reset else 0)

    def step_actions_observe(self, actions, observation=None, ended=None, status=None):
        """ONE library call per policy ply (bgs_step_actions_observe): apply `actions` -- a DEVICE tensor, int32[n] columns for
        Connect, int32[n, 4] moves for Bounce; negative = skip the board -- and write the observation of the boards after the
        move for the policy's next choice: Connect the legal mask uint8[n, width], Bounce the target masks int64[n, width + 1]
        (`observation`, allocated when None), plus `ended` uint8[n] and `status` int32[n] (per-board result) when given.
        Everything stays on the device and on the batch's stream: no synchronisation, capturable in a HIP graph.  Returns the
        observation tensor.  The loop: obs = batch.legal_tensor(); while ...: obs = batch.step_actions_observe(policy(obs), obs)."""
        return self._observe(actions, observation, ended, None, status, 0)

    def _observe(self, actions, observation, ended, reward, status, flags: int):
        t = self._need_torch("step_actions_observe")
        width = 1 if self.game == _abi.GAME_CONNECT else 4
        want = (self.n,) if width == 1 else (self.n, 4)
        if not (hasattr(actions, "data_ptr") and actions.is_cuda and actions.dtype == t.int32 and tuple(actions.shape) == want
                and actions.is_contiguous()):
            raise TypeError(f"actions must be a contiguous int32 device tensor of shape {want}")
        if observation is None:
            observation = (t.empty((self.n, self.width), dtype=t.uint8, device=actions.device) if width == 1
                           else t.empty((self.n, self.width + 1), dtype=t.int64, device=actions.device))
        shape = (self.n, self.width) if width == 1 else (self.n, self.width + 1)
        dtypes = (t.uint8,) if width == 1 else (t.int64, getattr(t, "uint64", t.int64))   # (64-bit masks: either signedness)
        if not (observation.is_cuda and observation.dtype in dtypes and tuple(observation.shape) == shape and observation.is_contiguous()):
            raise TypeError(f"observation must be a contiguous {dtypes[0]} device tensor of shape {shape}")
        for name, buf, dt in (("ended", ended, t.uint8), ("status", status, t.int32)):
            if buf is not None and not (buf.is_cuda and buf.dtype == dt and tuple(buf.shape) == (self.n,) and buf.is_contiguous()):
                raise TypeError(f"{name} must be a contiguous {dt} device tensor of shape ({self.n},)")
        if reward is not None and not (reward.is_cuda and reward.dtype == t.int8 and tuple(reward.shape) == (self.n, 2) and reward.is_contiguous()):
            raise TypeError(f"reward must be a contiguous int8 device tensor of shape ({self.n}, 2)")
        _abi.check(_abi.lib().bgs_env_step(
            self._handle, ctypes.c_void_p(actions.data_ptr()), ctypes.c_void_p(observation.data_ptr()),
            ctypes.c_void_p(ended.data_ptr()) if ended is not None else None,
            ctypes.c_void_p(reward.data_ptr()) if reward is not None else None,
            ctypes.c_void_p(status.data_ptr()) if status is not None else None, ctypes.c_uint32(flags)))
        return observation

    def _need_torch(self, who: str):
        if self._torch is None:
            raise RuntimeError(f"{who} needs torch with a GPU")
        return self._torch

    # ---- snapshot / restore in the reference's wire format (State.to_json / State.from_json) -------------------
    def to_json_states(self) -> list:
        """The batch as a list of reference-shaped State JSON dicts {"grid", "player", "winner"}
        (tests/test_connect.py:131-138, tests/test_bounce.py:392-403)."""
        grid, player, winner = self.grid, self.player, self.winner
        return [{"grid": grid[i].tolist(), "player": int(player[i]), "winner": int(winner[i])} for i in range(self.n)]

    def from_json_states(self, states) -> np.ndarray:
        """Load a list of n State JSON dicts; returns per-board status (0 ok, -1 malformed and left untouched)."""
        if len(states) != self.n:
            raise TypeError(f"expected {self.n} states, got {len(states)}")
        try:
            grid = np.array([s["grid"] for s in states], dtype=np.int8)
            player = np.array([s["player"] for s in states], dtype=np.int8)
            winner = np.array([s["winner"] for s in states], dtype=np.int8)
        except (KeyError, TypeError, ValueError) as exc:
            raise RuntimeError(f"invalid state JSON: {exc}") from None
        return self.write_state(grid, player, winner)


def expand_outcomes(packed, n: int, out=None):
    """Packed 2-bit outcome codes (CUDA uint8 tensor) of n boards -> reward int8[n, 2] on the same device."""
    import torch

    if not packed.is_cuda or packed.dtype != torch.uint8 or not packed.is_contiguous() or packed.numel() < (n + 3) // 4:
        raise TypeError("packed outcomes must be a contiguous CUDA uint8 tensor of (n + 3) // 4 bytes")
    if out is None:
        out = torch.empty((n, 2), dtype=torch.int8, device=packed.device)
    stream = torch.cuda.current_stream(packed.device).cuda_stream
    _abi.check(
        _abi.lib().bgs_expand_outcomes(
            packed.device.index, ctypes.c_void_p(stream), ctypes.c_void_p(packed.data_ptr()), n, ctypes.c_void_p(out.data_ptr())
        )
    )
    return out


class OneBoardCall:
    """`bgs_transition` on a one-board batch with everything that does not change between calls made once: the input
    and output arrays, their ctypes pointers and the bound function.  The object API makes one such call per State;
    building seven arrays and thirteen pointer objects per call cost as much as the device round trip."""

    def __init__(self, batch: "GameBatch"):
        if batch.n != 1:
            raise ValueError("OneBoardCall serves one-board batches")
        self.batch = batch
        h, w = batch.height, batch.width
        self.grid_in = np.empty((h, w), dtype=np.int8)
        self.player_in = np.zeros(1, dtype=np.int8)
        self.winner_in = np.zeros(1, dtype=np.int8)
        self.plies_in = np.zeros(1, dtype=np.int32)
        self.action_in = np.zeros(batch._action_width, dtype=np.int32)
        self.status = np.zeros(1, dtype=np.int32)
        self.grid_out = np.empty((h, w), dtype=np.int8)
        self.player_out = np.empty(1, dtype=np.int8)
        self.winner_out = np.empty(1, dtype=np.int8)
        self.plies_out = np.empty(1, dtype=np.int32)
        self.legal_out = batch._empty_legal()
        self.reward_out = np.empty(2, dtype=np.int8)
        i8, i32 = ctypes.c_int8, ctypes.c_int32
        self._load = (_ptr(self.grid_in, i8), _ptr(self.player_in, i8), _ptr(self.winner_in, i8))
        self._plies = _ptr(self.plies_in, i32)
        self._action = _ptr(self.action_in, i32)
        self._out = (
            _ptr(self.status, i32), _ptr(self.grid_out, i8), _ptr(self.player_out, i8), _ptr(self.winner_out, i8),
            _ptr(self.plies_out, i32), ctypes.c_void_p(self.legal_out.ctypes.data), _ptr(self.reward_out, i8),
        )
        self._fn = _abi.lib().bgs_transition
        self._none3 = (None, None, None)

    def __call__(self, grid=None, player: int = 0, winner: int = -1, plies=None, action=None) -> int:
        """Optional load (grid int8[h, w] + player + winner [+ plies]), optional move (an int or a sequence of
        `_action_width` ints), then observe into the `*_out` arrays.  Returns the board's status (0, or a BGS_ERR_*)."""
        load = self._none3
        if grid is not None:
            self.grid_in[...] = grid
            self.player_in[0] = player
            self.winner_in[0] = winner
            load = self._load
            if plies is not None:
                self.plies_in[0] = plies
        if action is not None:
            self.action_in[...] = action
        rc = self._fn(
            self.batch._handle, *load, None if grid is None or plies is None else self._plies,
            None if action is None else self._action, *self._out,
        )
        if rc:
            _abi.check(rc)
        return int(self.status[0])


class ConnectBatch(_Batch):
    """N Connect-k boards: ``Config(height, width, count)`` (reference connect.cpp:26) times n."""

    game = _abi.GAME_CONNECT
    _action_width = 1

    def _empty_legal(self):
        return np.empty((self.n, self.width), dtype=np.uint8)

    def __init__(self, height: int, width: int, count: int, n: int, device: int = 0, use_torch: Optional[bool] = None):
        super().__init__(n, height, width, device, use_torch)
        self.count = int(count)
        nbytes = ctypes.c_size_t()
        _abi.check(_abi.lib().bgs_connect_arena_bytes(self.height, self.width, self.count, self.n, ctypes.byref(nbytes)))
        arena, arena_bytes = self._make_arena(nbytes.value)
        _abi.check(
            _abi.lib().bgs_connect_create(
                self.height, self.width, self.count, self.n, self.device, arena, arena_bytes, ctypes.byref(self._handle)
            )
        )
        self._after_create()

    def step_actions(self, columns, want_status: bool = True):
        """columns int32[n]; a negative entry skips the board.  Returns per-board status (0 / -2 illegal)."""
        return self._step_actions(columns, 1, want_status)

    @property
    def legal(self) -> np.ndarray:
        out = np.empty((self.n, self.width), dtype=np.uint8)
        _abi.check(_abi.lib().bgs_read_legal(self._handle, _ptr(out, ctypes.c_uint8)))
        return out

    def legal_tensor(self, out=None):
        t = self._torch
        if t is None:
            raise RuntimeError("legal_tensor needs torch with a GPU")
        if out is None:
            out = t.empty((self.n, self.width), dtype=t.uint8, device=f"cuda:{self.device}")
        _abi.check(_abi.lib().bgs_export_device(self._handle, ord("l"), ctypes.c_void_p(out.data_ptr())))
        return out


class BounceBatch(_Batch):
    """N Bounce boards sharing one start grid: ``Config(grid)`` (reference bounce.cpp:26) times n."""

    game = _abi.GAME_BOUNCE
    _action_width = 4

    def _empty_legal(self):
        if self.generic:  # wide record: int32 active row, then uint8 flags[W][H * W] (include/bgs.h, bgs_legal_bytes)
            return np.empty((self.n, self._legal_bytes), dtype=np.uint8)
        return np.empty((self.n, self.width + 1), dtype=np.uint64)

    def decode_moves(self, legal_row, winner: int):
        """One board's legal-move record (either format) -> tuple of ((sx, sy), (tx, ty)) in canonical order."""
        width, height = self.width, self.height
        moves = []
        if winner != -1:
            return ()
        if self.generic:
            row = int(legal_row[:4].view(np.int32)[0])
            if row < 0:
                return ()
            flags = legal_row[4 : 4 + width * height * width].reshape(width, height * width)
            for x in range(width):
                for c in np.flatnonzero(flags[x]):
                    moves.append(((x, row), (int(c) % width, int(c) // width)))
            return tuple(moves)
        masks = legal_row.tolist()
        row = masks[width]
        if row >= height:  # all ones: nothing can move
            return ()
        cells = self._cells  # cell index -> (x, y), built once
        for x in range(width):
            m = masks[x]
            source = (x, row)
            while m:
                low = m & -m
                moves.append((source, cells[low.bit_length() - 1]))
                m ^= low
        return tuple(moves)

    def __init__(self, grid, n: int, device: int = 0, use_torch: Optional[bool] = None):
        cfg = np.ascontiguousarray(grid)
        if cfg.ndim != 2:
            raise TypeError("Bounce config grid must be 2-dimensional")
        if not np.issubdtype(cfg.dtype, np.integer):
            raise TypeError("Bounce config grid must be an integer array")
        if cfg.size and (cfg.min() < -128 or cfg.max() > 127):
            raise TypeError("Bounce config grid does not fit int8")
        cfg = cfg.astype(np.int8)
        super().__init__(n, cfg.shape[0], cfg.shape[1], device, use_torch)
        self.config_grid = cfg
        nbytes = ctypes.c_size_t()
        _abi.check(_abi.lib().bgs_bounce_arena_bytes(self.height, self.width, self.n, ctypes.byref(nbytes)))
        arena, arena_bytes = self._make_arena(nbytes.value)
        _abi.check(
            _abi.lib().bgs_bounce_create(
                _ptr(cfg, ctypes.c_int8), self.height, self.width, self.n, self.device, arena, arena_bytes, ctypes.byref(self._handle)
            )
        )
        self._cells = tuple((c % self.width, c // self.width) for c in range(self.height * self.width))
        self._after_create()

    def step_actions(self, moves, want_status: bool = True):
        """moves int32[n, 4] = source x, y, target x, y; a negative first entry skips the board."""
        return self._step_actions(moves, 4, want_status)

    def targets_tensor(self, out=None):
        """int64[n, W + 1] on the device (bgs_export_device 't'; the bits of `targets`, as torch's signed 64-bit type)."""
        t = self._need_torch("targets_tensor")
        if self.generic:
            raise ValueError("this board does not fit 64-bit target masks: use transition() / decode_moves()")
        if out is None:
            out = t.empty((self.n, self.width + 1), dtype=t.int64, device=f"cuda:{self.device}")
        return self._export("t", out)

    @property
    def targets(self) -> np.ndarray:
        """uint64[n, W + 1]: entry i < W has bit (y * W + x) set for each legal target of the piece in column i of the
        active row; entry W is the active row's y (all ones when nothing can move)."""
        if self.generic:
            raise ValueError("this board does not fit 64-bit target masks: use transition() / decode_moves()")
        out = np.empty((self.n, self.width + 1), dtype=np.uint64)
        _abi.check(_abi.lib().bgs_bounce_read_targets(self._handle, _ptr(out, ctypes.c_uint64)))
        return out
