"""CPU-only: the host half of the reward hand-over (bgs_expand_outcomes_host) against the oracle's reward rule.
The function is a table look-up in libbgs.so's host code and needs no GPU."""

import ctypes

import numpy as np

from oracle import oracle


def pack_codes(status: np.ndarray) -> np.ndarray:
    """2 bits per game, 4 games per byte, game 4i in the low bits (include/bgs.h, bgs_pack_outcomes)."""
    n = status.shape[0]
    padded = np.zeros((n + 3) // 4 * 4, dtype=np.uint8)
    padded[:n] = status
    q = padded.reshape(-1, 4)
    return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)


def winner_of(status: np.ndarray) -> np.ndarray:
    return np.where(status == 0, -1, np.where(status == 3, 2, status.astype(np.int8) - 1)).astype(np.int8)


def test_expand_matches_the_oracle_reward_rule():
    from simulator.batch import expand_outcomes_host

    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4, 5, 63, 64, 1000, 4099, 1 << 16):
        status = rng.integers(0, 4, size=n).astype(np.uint8)
        got = expand_outcomes_host(pack_codes(status), n)
        np.testing.assert_array_equal(got, oracle.reward(winner_of(status)))


def test_expand_in_shares_like_the_sink_workers():
    from simulator.batch import expand_outcomes_host

    rng = np.random.default_rng(12)
    n, threads = 10007, 7
    status = rng.integers(0, 4, size=n).astype(np.uint8)
    packed = pack_codes(status)
    out = np.full((n, 2), 99, dtype=np.int8)
    nbytes = (n + 3) // 4
    for t in range(threads):
        first = nbytes * t // threads * 4
        last = min(nbytes * (t + 1) // threads * 4, n)
        expand_outcomes_host(packed, n, out, first=first, count=last - first)
    np.testing.assert_array_equal(out, oracle.reward(winner_of(status)))


def test_expand_rejects_bad_arguments():
    from simulator.game import _abi

    buf = np.zeros(8, dtype=np.uint8)
    out = np.zeros((32, 2), dtype=np.int8)
    lib = _abi.lib()
    assert lib.bgs_expand_outcomes_host(ctypes.c_void_p(buf.ctypes.data), 2, 4, ctypes.c_void_p(out.ctypes.data)) == _abi.BGS_ERR_ARG
    assert lib.bgs_expand_outcomes_host(None, 0, 4, ctypes.c_void_p(out.ctypes.data)) == _abi.BGS_ERR_ARG
    assert lib.bgs_expand_outcomes_host(ctypes.c_void_p(buf.ctypes.data), 0, 0, ctypes.c_void_p(out.ctypes.data)) == 0
