"""The in-library reward gather (bgs_gather_*, csrc/bgs_multi.hip) with PEERS on the one-GPU box.  RCCL cannot put two
ranks on one device, so libbgs.so is pointed at a test-only transport with the same nine nccl* entry points
(BGS_RCCL_LIB=tests/c/libfake_rccl.so: shared-memory mailboxes + stream-ordered copies) and 2-3 processes sharing the GPU
run the library's real world > 1 branch: the communication thread, groups of ncclSend / ncclRecv of 1 and of slots / 2
steps, partial groups when somebody waits for the newest step, a dawdling rank (the ranks' groups then differ in size),
receives into device memory + the copy kernel and straight into the sink's device-mapped page-locked slot, the
create-time transport check.  Rank 0 compares EVERY rank's rows of EVERY delivered step with the oracle
(tests/gather_peer.py).  What this does not show is RCCL itself over xGMI: that needs two GPUs."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "c", "libfake_rccl.so")
PEER = os.path.join(ROOT, "tests", "gather_peer.py")


def build_fake_rccl():
    src = os.path.join(ROOT, "tests", "c", "fake_rccl.hip")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "--offload-arch=gfx950", "-Wall",
                               src, "-o", FAKE, "-lpthread", "-lrt"])
    return FAKE


def run_ranks(tmp_path, world, mode, extra_env=None, timeout=240, per_process=1):
    """`world` ranks as world / per_process processes (per_process ranks on threads of one process: the box allows at most
    6 processes on the card); returns one stdout / stderr per PROCESS."""
    env = dict(os.environ, BGS_RCCL_LIB=build_fake_rccl())
    env.update(extra_env or {})
    groups = [",".join(str(r) for r in range(first, min(world, first + per_process))) for first in range(0, world, per_process)]
    assert len(groups) <= 5   # (the box allows 6 processes on the card, and the test process is one of them)
    procs = [subprocess.Popen(["timeout", "-k", "10", str(timeout), sys.executable, PEER, str(tmp_path), g, str(world), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for g in groups]
    outs = [p.communicate() for p in procs]
    for g, p, (out, err) in zip(groups, procs, outs):
        assert p.returncode == 0, f"rank(s) {g} exit {p.returncode}\n{out[-2000:]}\n{err[-3000:]}"
    return [out for out, _ in outs], [err for _, err in outs]


@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("batch", ["1", "4"])
def test_gather_with_three_ranks_sharing_the_gpu(tmp_path, direct, batch):
    """3 ranks x {copy kernel, direct receives} x {a group per step, groups of slots / 2 steps}; rank 2 dawdles, so its
    groups are cut where the others' are not.  Single calls with waits that force partial groups, then the native loop on
    the same gather."""
    outs, _ = run_ranks(tmp_path, 3, "steps", {"BGS_GATHER_DIRECT": direct, "BGS_GATHER_BATCH": batch, "PEER_SLOW_RANK": "2"})
    for r, out in enumerate(outs):
        assert f"PEER_OK rank {r} verified 43" in out, out
    assert f"'direct': {direct == '1'}" in outs[0] and f"'batch': {batch}" in outs[0] and "'transport_check': 'passed'" in outs[0]
    assert "libfake_rccl.so" in outs[0]


def test_gather_defaults_with_two_ranks(tmp_path):
    """The defaults a real N > 1 run gets: receives into device memory + one copy kernel per group, groups of slots / 2."""
    outs, _ = run_ranks(tmp_path, 2, "steps", {"PEER_GAMES": "20012"})
    assert "'direct': False" in outs[0] and "'batch': 4" in outs[0]


def test_a_step_that_cannot_be_enqueued_does_not_stall_the_gather(tmp_path):
    """Round-3 advisor: a failure after rank 0 claimed the step's sink ticket left the ticket unpublished and
    bgs_gather_destroy waiting for it.  The failed step now travels through the communication thread with its flag
    down: the call reports the failure, waits return, close() returns -- in the middle of a group of 4."""
    outs, _ = run_ranks(tmp_path, 2, "inject", {"BGS_EXPERIMENT": "gather_inject_failure=5", "BGS_GATHER_BATCH": "4"}, timeout=120)
    for r, out in enumerate(outs):
        assert f"INJECT_OK rank {r}" in out, out


def test_one_process_two_logical_devices(tmp_path):
    """bgs_multi_create / rollout / destroy and bgs_multi_connect_rollout (one process, every device of the node) with
    the one GPU listed two and three times: the sends and receives of bgs_multi_rollout's group, device r playing global
    games [r n, (r + 1) n)."""
    outs, _ = run_ranks(tmp_path, 1, "multi")
    assert "MULTI_OK" in outs[0]


@pytest.mark.parametrize("direct", ["0", "1"])
def test_gather_with_eight_ranks(tmp_path, direct):
    """The world size BASELINE config 5 names.  Eight ranks as 4 processes x 2 ranks (the box's limit is 6 processes on
    the card): rank 0 receives 7 messages per step at dst + r * code_bytes, its sink expands 8 x n games per step; {copy
    kernel, direct receives} x the default group (slots / 2); every rank's rows of every delivered step == oracle."""
    outs, _ = run_ranks(tmp_path, 8, "steps", {"BGS_GATHER_DIRECT": direct, "PEER_SLOW_RANK": "5", "PEER_GAMES": "4096",
                                               "BGS_FAKE_RCCL_MSG_BYTES": "4096"}, per_process=2, timeout=400)
    text = "\n".join(outs)
    for r in range(8):
        assert f"PEER_OK rank {r} verified 43" in text, text
    assert f"'direct': {direct == '1'}" in outs[0] and "'batch': 4" in outs[0] and "'ranks': 8" in outs[0]


@pytest.mark.parametrize("world,per_process", [(2, 1), (4, 1), (8, 2)])
def test_config_5_shapes_through_the_gather(tmp_path, world, per_process):
    """BASELINE config 5's arithmetic at the world sizes the metric names: 2^18 games per rank here (2^20 with
    PEER_GAMES=1048576, tools/gather_full_size.sh), 12 host arrays, groups of 6, 3 batches in flight, the native loop;
    the communicator itself reports `world` ranks; rank 0 compares every rank's rows of all 14 steps with the oracle."""
    n = 1 << 18
    outs, _ = run_ranks(tmp_path, world, "full", {"PEER_GAMES": str(n), "PEER_STEPS": "14", "BGS_FAKE_RCCL_MSG_BYTES": str(n // 4)},
                        per_process=per_process, timeout=600)
    text = "\n".join(outs)
    for r in range(world):
        assert f"FULL_OK rank {r} of {world} verified 14 steps of {world} x {n} games" in text, text
    assert f"'ranks': {world}" in outs[0]


def test_a_rank_that_submits_one_step_and_then_blocks_elsewhere(tmp_path):
    """Round-4 advisor (medium): ranks other than 0 submit ONE step of a group of 4 and sit at a barrier that only opens
    when rank 0 has that step's rewards; nobody but rank 0 waits for a ticket.  The partial group leaves by itself."""
    outs, _ = run_ranks(tmp_path, 3, "lone_step", {"BGS_GATHER_BATCH": "4"}, timeout=120)
    for r, out in enumerate(outs):
        assert f"LONE_OK rank {r}" in out, out


@pytest.mark.parametrize("bad_rank", [1, 0])
def test_one_rank_fails_alone_and_nobody_is_left_waiting(tmp_path, bad_rank):
    """Round-4 advisor (low): a step that cannot be enqueued on ONE rank.  That rank reports it; its message still goes
    out (zeros), the peers' groups complete, rank 0 delivers every step."""
    outs, _ = run_ranks(tmp_path, 3, "inject_one", {"BGS_EXPERIMENT": f"gather_inject_failure=5;gather_inject_rank={bad_rank}", "PEER_BAD_RANK": str(bad_rank),
                                                    "PEER_BAD_STEP": "5", "BGS_GATHER_BATCH": "4"}, timeout=120)
    for r, out in enumerate(outs):
        assert f"INJECT_ONE_OK rank {r}" in out, out


# ---- the stand-in refuses what RCCL refuses (or answers with a hang / silent corruption): one test per refusal ----
class _Fake:
    """The stand-in driven directly (ctypes): two communicators of ONE process on the one GPU (ncclCommInitAll), device
    buffers from torch.  Run in a child process per case so that BGS_FAKE_RCCL_TIMEOUT_MS is read afresh."""

    CODE = r'''
import ctypes, os, sys
import torch
lib = ctypes.CDLL(sys.argv[1])
lib.ncclGetErrorString.restype = ctypes.c_char_p
for f in (lib.ncclSend, lib.ncclRecv):
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
comms = (ctypes.c_void_p * 2)()
assert lib.ncclCommInitAll(comms, 2, (ctypes.c_int * 2)(0, 0)) == 0
a = torch.arange(64, dtype=torch.uint8, device="cuda")
b = torch.zeros(64, dtype=torch.uint8, device="cuda")
U8, I32 = 1, 2
def err(rc):
    return lib.ncclGetErrorString(rc).decode() if rc else "ok"
case = sys.argv[2]
if case == "good":
    assert lib.ncclGroupStart() == 0
    assert lib.ncclSend(a.data_ptr(), 64, U8, 1, comms[0], None) == 0
    assert lib.ncclRecv(b.data_ptr(), 64, U8, 0, comms[1], None) == 0
    rc = lib.ncclGroupEnd()
    torch.cuda.synchronize()
    print("RESULT", err(rc), bool((a == b).all()))
elif case in ("count", "dtype"):
    lib.ncclGroupStart()
    lib.ncclSend(a.data_ptr(), 64, U8, 1, comms[0], None)
    lib.ncclRecv(b.data_ptr(), 32 if case == "count" else 16, U8 if case == "count" else I32, 0, comms[1], None)
    print("RESULT", err(lib.ncclGroupEnd()))
elif case == "ungrouped_send":
    print("RESULT", err(lib.ncclSend(a.data_ptr(), 64, U8, 1, comms[0], None)))
elif case == "ungrouped_recv":
    print("RESULT", err(lib.ncclRecv(b.data_ptr(), 64, U8, 0, comms[1], None)))
elif case == "end_without_start":
    print("RESULT", err(lib.ncclGroupEnd()))
elif case == "peer_out_of_range":
    lib.ncclGroupStart()
    print("RESULT", err(lib.ncclSend(a.data_ptr(), 64, U8, 2, comms[0], None)))
elif case == "null_buffer":
    lib.ncclGroupStart()
    print("RESULT", err(lib.ncclRecv(None, 64, U8, 0, comms[1], None)))
elif case == "bad_type":
    lib.ncclGroupStart()
    print("RESULT", err(lib.ncclSend(a.data_ptr(), 64, 99, 1, comms[0], None)))
elif case == "destroy_in_group":
    lib.ncclGroupStart()
    print("RESULT", err(lib.ncclCommDestroy(comms[0])))
elif case == "destroyed_comm":
    assert lib.ncclCommDestroy(comms[1]) == 0
    n = ctypes.c_int(0)
    r1 = lib.ncclCommCount(comms[1], ctypes.byref(n))
    lib.ncclGroupStart()
    print("RESULT", err(r1), "|", err(lib.ncclRecv(b.data_ptr(), 64, U8, 0, comms[1], None)))
elif case == "silent_peer":
    # the receive's peer never posts the matching send: the wait must end in an error, not in a hang
    lib.ncclGroupStart()
    lib.ncclRecv(b.data_ptr(), 64, U8, 0, comms[1], None)
    print("RESULT", err(lib.ncclGroupEnd()))
sys.stdout.flush()
os._exit(0)
'''

    @staticmethod
    def run(case, timeout_ms="1500"):
        env = dict(os.environ, BGS_FAKE_RCCL_TIMEOUT_MS=timeout_ms)
        proc = subprocess.run([sys.executable, "-c", _Fake.CODE, build_fake_rccl(), case], env=env, capture_output=True, text=True, timeout=180)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT")]
        assert proc.returncode == 0 and lines, proc.stderr[-2000:]
        return lines[-1][len("RESULT "):]


def test_stand_in_delivers_a_matched_message():
    assert _Fake.run("good") == "ok True"


@pytest.mark.parametrize("case,what", [
    ("count", "count / data type mismatch"),
    ("dtype", "count / data type mismatch"),
    ("ungrouped_send", "ncclSend outside ncclGroupStart / ncclGroupEnd"),
    ("ungrouped_recv", "ncclRecv outside ncclGroupStart / ncclGroupEnd"),
    ("end_without_start", "ncclGroupEnd without ncclGroupStart"),
    ("peer_out_of_range", "peer 2 is outside the communicator"),
    ("null_buffer", "NULL buffer with a non-zero count"),
    ("bad_type", "unknown data type 99"),
    ("destroy_in_group", "inside an open group"),
    ("silent_peer", "the peer never sent the message (timeout)"),
])
def test_stand_in_refuses(case, what):
    """What RCCL answers with an error, a hang or silent corruption is an error code here (tests/c/fake_rccl.hip, header)."""
    assert what in _Fake.run(case)


def test_stand_in_refuses_a_destroyed_communicator():
    out = _Fake.run("destroyed_comm")
    assert "bad arguments to ncclCommCount" in out and "not a live communicator" in out
