"""CPU-only: the host half of the reward hand-over (bgs_expand_outcomes_host) against the oracle's reward rule.
The function is a table look-up in libbgs.so's host code and needs no GPU."""

import ctypes

import numpy as np
import pytest

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


@pytest.mark.parametrize("portable", [0, 1])
@pytest.mark.parametrize("cells,sets,offset,weights", [(42, 2, -1, (1, 1)), (54, 4, 0, (1, 2, 4, 8)), (156, 2, -1, (1, 1)),
                                                       (64, 4, 0, (1, 2, 4, 8)), (7, 2, -1, (1, 1)), (129, 1, 3, (5,))])
def test_grid_expansion_on_the_host(cells, sets, offset, weights, portable):
    """bgs_expand_grid_host (the host half of the grid hand-over): bit sets over the cells -> int8 per cell, against a
    numpy restatement; the AVX-512 path (where the CPU has it) and the table path; sub-ranges leave the rest alone."""
    import ctypes

    from simulator.game import _abi

    rng = np.random.default_rng(cells * 10 + sets)
    n, nwc = 300, (cells + 63) // 64   # (whole 64-game blocks go out with non-temporal stores when the array is 64-byte aligned)
    bits = rng.integers(0, 2, size=(sets, n, cells), dtype=np.uint8)
    wire = np.zeros((sets * nwc, n), dtype=np.uint64)
    for p in range(sets):
        for c in range(cells):
            wire[p * nwc + c // 64] |= bits[p, :, c].astype(np.uint64) << np.uint64(c % 64)
    want = (offset + sum(int(w) * bits[p].astype(np.int32) for p, w in enumerate(weights))).astype(np.int8)
    raw = np.full(n * cells + 64, 99, dtype=np.int8)
    shift = (-raw.ctypes.data) % 64
    got = raw[shift : shift + n * cells].reshape(n, cells)
    assert got.ctypes.data % 64 == 0
    wts = (ctypes.c_int32 * sets)(*weights)
    _abi.check(_abi.lib().bgs_expand_grid_host(ctypes.c_void_p(wire.ctypes.data), n, cells, sets, offset, wts, 5, 270,
                                               ctypes.c_void_p(got.ctypes.data), portable))
    np.testing.assert_array_equal(got[5:275], want[5:275])
    assert (got[:5] == 99).all() and (got[275:] == 99).all() and (raw[:shift] == 99).all() and (raw[shift + n * cells :] == 99).all()
    _abi.check(_abi.lib().bgs_expand_grid_host(ctypes.c_void_p(wire.ctypes.data), n, cells, sets, offset, wts, 0, n,
                                               ctypes.c_void_p(got.ctypes.data), portable))
    np.testing.assert_array_equal(got, want)
    assert _abi.lib().bgs_expand_grid_host(ctypes.c_void_p(wire.ctypes.data), n, cells, sets, offset, wts, 290, 20,
                                           ctypes.c_void_p(got.ctypes.data), portable) == _abi.BGS_ERR_ARG
