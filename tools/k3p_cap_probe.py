"""Probe: short-cap rollouts from the start position (K3p, or K3f with bounce_pieces=0) with parking on: every board must be stored."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from tests.knobs import knobs  # BGS_EXPERIMENT ("name=value;...") as a mapping
knobs.update({"bounce_group": "1", "bounce_park": os.environ.get("PARK", "32"), "bounce_chunk": "32"})
import numpy as np
from oracle import oracle
from simulator.batch import BounceBatch
g = np.zeros((9, 6), dtype=np.int8); g[1] = g[7] = [1, 2, 3, 3, 2, 1]
SEED = 0x0123456789ABCDEF
bad = 0
for n in (20011, 4099, 65536 + 77):
    for cap in (5, 2, 9):
        orc = oracle.BounceOracle(g, n); want = orc.rollout(SEED ^ n, first_game=n, max_plies=cap)
        for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
            dev = BounceBatch(g, n)
            dev.set_first_game(n)
            dev.rollout(SEED ^ n, max_plies=cap, from_initial=True)
            plies = dev.plies
            if dev.steps != want or not np.array_equal(plies, orc.plies):
                bad += 1
                wrong = np.flatnonzero(plies != orc.plies)
                print(f"n={n} cap={cap} rep {rep}: steps {dev.steps} vs {want}; wrong plies on {len(wrong)} boards, first {wrong[:4].tolist()}")
            dev.close()
print("mismatches:", bad)
