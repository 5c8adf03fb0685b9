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


def run_ranks(tmp_path, world, mode, extra_env=None, timeout=240):
    env = dict(os.environ, BGS_RCCL_LIB=build_fake_rccl())
    env.update(extra_env or {})
    procs = [subprocess.Popen(["timeout", "-k", "10", str(timeout), sys.executable, PEER, str(tmp_path), str(r), str(world), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate() for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exit {p.returncode}\n{out[-2000:]}\n{err[-3000:]}"
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
    outs, _ = run_ranks(tmp_path, 2, "inject", {"BGS_GATHER_INJECT_FAILURE": "5", "BGS_GATHER_BATCH": "4"}, timeout=120)
    for r, out in enumerate(outs):
        assert f"INJECT_OK rank {r}" in out, out


def test_one_process_two_logical_devices(tmp_path):
    """bgs_multi_create / rollout / destroy and bgs_multi_connect_rollout (one process, every device of the node) with
    the one GPU listed two and three times: the sends and receives of bgs_multi_rollout's group, device r playing global
    games [r n, (r + 1) n)."""
    outs, _ = run_ranks(tmp_path, 1, "multi")
    assert "MULTI_OK" in outs[0]
