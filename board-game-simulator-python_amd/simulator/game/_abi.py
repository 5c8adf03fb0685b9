"""ctypes binding of libbgs.so (C ABI: include/bgs.h) -- the only door from Python into the HIP kernels.

There is no CPU fallback: if the library is missing or no GPU is visible, the compute entry points raise.
The nanobind modules of the reference (src/simulator/game/connect.cpp, bounce.cpp) are replaced by this file plus
the thin Python classes in connect.py / bounce.py / ../batch.py.
"""

from __future__ import annotations

import ctypes
import importlib.util
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.environ.get("BGS_LIBRARY", os.path.join(_PKG_DIR, "libbgs.so"))

BGS_OK = 0
BGS_ERR_ARG = -1
BGS_ERR_ILLEGAL = -2
BGS_ERR_RUNTIME = -3
BGS_ERR_NO_DEVICE = -4

BUF_PLANES, BUF_STATUS, BUF_PLIES, BUF_REWARD, BUF_STEPS, BUF_STAGING = range(6)
ROLLOUT_FROM_INITIAL = 1
ROLLOUT_DRAW_PER_PLY = 4   # Connect: this rollout call draws under the strict contract
RNG_PER_BLOCK, RNG_PER_PLY = 0, 1   # bgs_set_rng_contract
ENV_AUTO_RESET = 1

GAME_CONNECT = 1
GAME_BOUNCE = 2

c_handle = ctypes.c_void_p
_i8p = ctypes.POINTER(ctypes.c_int8)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)

# name -> (restype, argtypes): every symbol include/bgs.h declares
SIGNATURES = {
    "bgs_version": (ctypes.c_int, []),
    "bgs_last_error": (ctypes.c_char_p, []),
    "bgs_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "bgs_build_id": (ctypes.c_char_p, []),
    "bgs_kernel_unit_id": (ctypes.c_char_p, [ctypes.c_int]),
    "bgs_connect_arena_bytes": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]),
    "bgs_connect_create": (
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(c_handle)],
    ),
    "bgs_bounce_arena_bytes": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]),
    "bgs_bounce_create": (
        ctypes.c_int,
        [_i8p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(c_handle)],
    ),
    "bgs_destroy": (ctypes.c_int, [c_handle]),
    "bgs_set_stream": (ctypes.c_int, [c_handle, ctypes.c_void_p]),
    "bgs_stream_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "bgs_stream_destroy": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "bgs_set_first_game": (ctypes.c_int, [c_handle, ctypes.c_uint64]),
    "bgs_set_rng_contract": (ctypes.c_int, [c_handle, ctypes.c_int]),
    "bgs_set_launches_in_flight": (ctypes.c_int, [c_handle, ctypes.c_int32]),
    "bgs_synchronize": (ctypes.c_int, [c_handle]),
    "bgs_info": (
        ctypes.c_int,
        [c_handle] + [ctypes.POINTER(ctypes.c_int)] * 4 + [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)],
    ),
    "bgs_legal_bytes": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]),
    "bgs_buffer": (ctypes.c_int, [c_handle, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]),
    "bgs_reset": (ctypes.c_int, [c_handle]),
    "bgs_step_random": (ctypes.c_int, [c_handle, ctypes.c_uint64]),
    "bgs_step_random_n": (ctypes.c_int, [c_handle, ctypes.c_uint64, ctypes.c_int32]),
    "bgs_step_actions": (ctypes.c_int, [c_handle, ctypes.c_void_p, ctypes.c_int, _i32p]),
    "bgs_rollout": (ctypes.c_int, [c_handle, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32]),
    "bgs_steps": (ctypes.c_int, [c_handle, _u64p]),
    "bgs_reset_steps": (ctypes.c_int, [c_handle]),
    "bgs_read_grid": (ctypes.c_int, [c_handle, _i8p]),
    "bgs_read_player": (ctypes.c_int, [c_handle, _i8p]),
    "bgs_read_ended": (ctypes.c_int, [c_handle, _u8p]),
    "bgs_read_winner": (ctypes.c_int, [c_handle, _i8p]),
    "bgs_read_reward": (ctypes.c_int, [c_handle, _i8p]),
    "bgs_read_plies": (ctypes.c_int, [c_handle, _i32p]),
    "bgs_read_legal": (ctypes.c_int, [c_handle, _u8p]),
    "bgs_read_action_count": (ctypes.c_int, [c_handle, _i32p]),
    "bgs_bounce_read_targets": (ctypes.c_int, [c_handle, _u64p]),
    "bgs_export_device": (ctypes.c_int, [c_handle, ctypes.c_int, ctypes.c_void_p]),
    "bgs_step_actions_observe": (ctypes.c_int, [c_handle, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "bgs_env_step": (ctypes.c_int, [c_handle, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "bgs_pack_outcomes": (ctypes.c_int, [c_handle, ctypes.c_void_p]),
    "bgs_rollout_pack": (ctypes.c_int, [c_handle, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p]),
    "bgs_expand_outcomes": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "bgs_transition": (
        ctypes.c_int,
        [c_handle, _i8p, _i8p, _i8p, _i32p, _i32p, _i32p, _i8p, _i8p, _i8p, _i32p, ctypes.c_void_p, _i8p],
    ),
    # asynchronous hand-over to host memory
    "bgs_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "bgs_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "bgs_event_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_handle)]),
    "bgs_event_destroy": (ctypes.c_int, [c_handle]),
    "bgs_event_synchronize": (ctypes.c_int, [c_handle]),
    "bgs_event_query": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_int)]),
    "bgs_read_reward_async": (ctypes.c_int, [c_handle, ctypes.c_void_p, c_handle]),
    "bgs_read_outcomes_async": (ctypes.c_int, [c_handle, ctypes.c_void_p, c_handle]),
    "bgs_expand_outcomes_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "bgs_rollout_to_host": (
        ctypes.c_int,
        [c_handle, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, c_handle],
    ),
    "bgs_sink_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_handle)]),
    "bgs_sink_destroy": (ctypes.c_int, [c_handle]),
    "bgs_grid_sink_create": (ctypes.c_int, [c_handle, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_handle)]),
    "bgs_expand_grid_host": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
         ctypes.c_int],
    ),
    "bgs_sink_submit": (ctypes.c_int, [c_handle, c_handle, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "bgs_sink_rollout": (
        ctypes.c_int,
        [c_handle, c_handle, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
    ),
    "bgs_sink_submit_packed": (
        ctypes.c_int,
        [c_handle, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
    ),
    "bgs_sink_wait": (ctypes.c_int, [c_handle, ctypes.c_int64]),
    "bgs_sink_completed": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_int64)]),
    "bgs_sink_set_progress": (ctypes.c_int, [c_handle, ctypes.c_void_p]),
    "bgs_progress_store": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "bgs_progress_barrier": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "bgs_progress_wait": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)],
    ),
    "bgs_bind_host_thread": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "bgs_gather_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "bgs_gather_create": (
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_handle)],
    ),
    "bgs_gather_rollout": (
        ctypes.c_int,
        [c_handle, c_handle, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
    ),
    "bgs_gather_wait": (ctypes.c_int, [c_handle, ctypes.c_int64]),
    "bgs_gather_info": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bgs_gather_comm": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bgs_gather_transport": (ctypes.c_char_p, []),
    "bgs_gather_destroy": (ctypes.c_int, [c_handle]),
    "bgs_pipeline_create": (
        ctypes.c_int,
        [ctypes.POINTER(c_handle), ctypes.c_int, c_handle, c_handle, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_uint64,
         ctypes.c_int32, ctypes.c_uint32, ctypes.POINTER(c_handle)],
    ),
    "bgs_pipeline_enqueue": (ctypes.c_int, [c_handle, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "bgs_pipeline_enqueue_seeds": (ctypes.c_int, [c_handle, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    "bgs_pipeline_wait": (ctypes.c_int, [c_handle, ctypes.c_int64]),
    "bgs_pipeline_feed": (ctypes.c_int, [c_handle, ctypes.c_void_p, ctypes.c_int64]),
    "bgs_pipeline_release": (ctypes.c_int, [c_handle, ctypes.c_int64]),
    "bgs_pipeline_drain": (ctypes.c_int, [c_handle]),
    "bgs_pipeline_progress": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "bgs_pipeline_kernel_ms": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "bgs_pipeline_timeline": (ctypes.c_int, [c_handle, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "bgs_pipeline_set_ring": (
        ctypes.c_int,
        [c_handle, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64],
    ),
    "bgs_pipeline_destroy": (ctypes.c_int, [c_handle]),
    "bgs_multi_connect_rollout": (
        ctypes.c_int,
        [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64,
         ctypes.c_void_p, _u64p],
    ),
    "bgs_multi_create": (
        ctypes.c_int,
        [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(c_handle)],
    ),
    "bgs_multi_rollout": (ctypes.c_int, [c_handle, ctypes.c_uint64, ctypes.c_void_p, _u64p]),
    "bgs_multi_destroy": (ctypes.c_int, [c_handle]),
    "bgs_write_state": (ctypes.c_int, [c_handle, _i8p, _i8p, _i8p, _i32p, _i32p]),
}

_lib = None


class BgsError(RuntimeError):
    """A libbgs call failed (the reference surfaces C++ exceptions as RuntimeError too)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libbgs error {code}: {message}")
        self.code = code


def _one_hip_runtime_per_process() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (SONAME libamdhip64.so.7, the name
    libbgs.so links against).  Two HSA runtimes in one process cannot both open the GPU, so whichever of torch and
    libbgs loads first must decide for both: if torch is installed but not imported yet, map ITS runtime now (no
    torch import needed); libbgs.so and a later `import torch` then bind to that same copy."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)


HW_QUEUES_WANTED = 24   # (16 / 20 / 24 / 28 / 32 Bounce batches in flight: 10.1 / 11.0 / 10.1 / 7.9 / 5.9 x 10^9 -- beyond 24 the queues thrash)


def request_hardware_queues(wanted: int = HW_QUEUES_WANTED) -> int:
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default -- and reads
    the variable once, when it initialises: beyond 4 batches in flight nothing overlaps until it is raised (Bounce
    rollouts want 20 batches in flight: 9x one launch at a time; Connect's 3 are unaffected either way, measured).
    OPT-IN: called by `RolloutPipeline` / `RolloutExecutor` users and bench.py when they need more than 4, never by
    importing the package -- the setting is process-wide and other HIP users of the process (torch) see it too.
    Returns the number of hardware queues this process has, or will have: `wanted` if the variable could still be set (or
    was already set by the user to at least that), else what the runtime came up with."""
    if "GPU_MAX_HW_QUEUES" not in os.environ and not _gpu_runtime_is_up():
        os.environ["GPU_MAX_HW_QUEUES"] = str(int(wanted))
    return hardware_queues()


def _gpu_runtime_is_up() -> bool:
    """Has anything in this process initialised the HIP / HSA runtime yet?  The runtime opens /dev/kfd when it comes up
    (`torch.cuda.is_available()` is enough; torch's own `is_initialized()` does not tell)."""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink(f"/proc/self/fd/{fd}") == "/dev/kfd":
                    return True
            except OSError:
                continue
    except OSError:
        pass
    torch = sys.modules.get("torch")
    try:
        return torch is not None and torch.cuda.is_initialized()
    except Exception:  # noqa: BLE001
        return False


_queues_at_start = None   # what the runtime came up with, once known


def hardware_queues() -> int:
    """Hardware queues this process's streams can spread over: what the HIP runtime was initialised with (4 unless
    GPU_MAX_HW_QUEUES said otherwise at that moment), or will be initialised with."""
    global _queues_at_start
    if _queues_at_start is not None:
        return _queues_at_start
    try:
        now = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        now = 4
    if _gpu_runtime_is_up():
        # the runtime has read the variable; a later change of the environment does not count.  (If somebody changed
        # it between the runtime's start and this first look, the figure is optimistic: nothing here can tell.)
        _queues_at_start = now
    return now


def lib() -> ctypes.CDLL:
    """Load libbgs.so; fail loudly when it is absent (build it with `python __graft_entry__.py`)."""
    global _lib
    if _lib is None:
        hardware_queues()   # (notes what the runtime has, if it is up already; changes nothing)
        _one_hip_runtime_per_process()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `make -C board-game-simulator-python_amd/csrc` or `python __graft_entry__.py`."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def last_error() -> str:
    msg = lib().bgs_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    if rc == BGS_OK:
        return
    msg = last_error()
    if rc == BGS_ERR_ARG:
        raise ValueError(f"libbgs: {msg}")
    raise BgsError(rc, msg)


def build_id() -> str:
    return lib().bgs_build_id().decode("ascii")


UNITS = ("connect", "bounce", "generic")


def unit_ids() -> dict:
    """{"connect": id, "bounce": id, "generic": id}: the id each kernel unit was compiled with (its source, bgs_common.h,
    its own unit header and the compile flags).  Counter files under profiles/ name the unit id of the kernel they
    describe; an edit to one unit leaves the others' counters quotable."""
    return {name: (lib().bgs_kernel_unit_id(k) or b"unknown").decode("ascii") for k, name in enumerate(UNITS)}


def device_count() -> int:
    n = ctypes.c_int(0)
    check(lib().bgs_device_count(ctypes.byref(n)))
    return n.value
