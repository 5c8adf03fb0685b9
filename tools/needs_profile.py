#!/usr/bin/env python3
"""Does a counters file have to be re-taken?  exit 0 = yes (its kernel unit moved, or there is no file), 1 = no.

    python3 tools/needs_profile.py stem rollout_counters      # profiles/r*_rollout_counters.json
    python3 tools/needs_profile.py case k3p_8x                # a case of profiles/r*_valu_busy.json

The files name the id of the kernel UNIT they were taken on (bgs_kernel_unit_id: connect / bounce / generic); a pass is
repeated only when that unit's id differs from the running library's -- a Bounce-only edit leaves the Connect counters
alone (BGS_PROFILE_FORCE=1 re-takes everything).  Reads the library's ids without touching the GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]


def main():
    import bench

    kind, name = sys.argv[1], sys.argv[2]
    if os.environ.get("BGS_PROFILE_FORCE") == "1":
        return 0
    ids = bench.running_ids()
    if kind == "stem":
        counters, _ = bench.committed_counters(ids, name)
        have = counters is not None and bool(counters.get("unit_id"))
    else:
        have = bench.busy_block(ids, name) is not None
    print(f"{kind} {name}: {'up to date for this unit' if have else 'to be taken'}", file=sys.stderr)
    return 1 if have else 0


if __name__ == "__main__":
    sys.exit(main())
