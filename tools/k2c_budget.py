#!/usr/bin/env python3
"""ISA-level budget of K2c's ply block (round-5 review, item 5): where the ~520 VALU of a four-ply block of
k_connect_rollout_lds<Geo<3, 12, 13, 5>, uncapped, codes, opened entry> go, by opcode class.
usage: python tools/k2c_budget.py [connect_kernels.s]   (without an argument: compiles csrc/connect_kernels.hip -S itself)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_blocks import blocks

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/k2c_budget.s"
if len(sys.argv) <= 1:
    csrc = os.path.join(ROOT, "board-game-simulator-python_amd", "csrc")
    flags = open(os.path.join(csrc, "flags.stamp")).read().split()[1:] if os.path.exists(os.path.join(csrc, "flags.stamp")) else []
    flags = [f for f in flags if f not in ("-fPIC",)]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-DBGS_TU_ID=\"budget\"", "--cuda-device-only", "-S",
                           os.path.join(csrc, "connect_kernels.hip"), "-o", path], stderr=subprocess.DEVNULL)
syms = [l.split(":")[0] for l in open(path) if re.match(r"^_Z\w*k_connect_rollout_lds\w*:", l)]
names = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
want = [s for s, n in zip(syms, names) if "Geo<3, 12, 13, 5>, false, true, 1, false>" in n]
assert len(want) == 1, (len(want), names[:4])
bl = blocks(path, want[0])
name, body = max(bl, key=lambda nb: sum(1 for x in nb[1] if x.startswith("v_")))
ops = collections.Counter(x.split()[0] for x in body)
GROUPS = [
    ("philox multiplies (v_mul_hi_u32 / v_mul_lo_u32 / v_mad_u64_u32)", r"v_mul_hi_u32|v_mul_lo_u32|v_mad_u64_u32"),
    ("three-input bit ops (v_bitop3: philox xor3, masked stone, and-or)", r"v_bitop3|v_and_or|v_or3|v_xad|v_xor3"),
    ("window alignment (v_alignbit)", r"v_alignbit"),
    ("shifts (run test doubling, nibble fields; 32- and 64-bit)", r"v_lshl|v_lshr|v_ashr"),
    ("and / or / xor / not (run test, masks)", r"v_and_b32|v_or_b32|v_xor_b32|v_not_b32"),
    ("popcount / ffs (open columns, column select)", r"v_bcnt|v_ffb|v_mbcnt"),
    ("add / sub / mad (LDS addresses, heights, counters)", r"v_add|v_sub|v_mad_u32|v_mul_u32_u24|v_mad_i32|v_lshl_add|v_add_lshl|v_lshl_or"),
    ("compares + selects", r"v_cmp|v_cndmask"),
    ("moves / readlane / bfe / other", r"v_"),
]
seen = set()
valu = sum(n for k, n in ops.items() if k.startswith("v_"))
print(f"kernel {want[0][:60]}...  block {name}: {len(body)} instructions, {valu} VALU, "
      f"{sum(n for k, n in ops.items() if k.startswith('ds_'))} LDS, {sum(n for k, n in ops.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop')))} SALU, "
      f"{sum(n for k, n in ops.items() if k.startswith(('s_waitcnt', 's_nop')))} waits / nops")
for title, pat in GROUPS:
    rows = {k: n for k, n in ops.items() if k.startswith("v_") and k not in seen and re.match(pat, k)}
    seen |= set(rows)
    total = sum(rows.values())
    print(f"  {total:4d}  {100 * total / valu:5.1f} %  {title}: " + ", ".join(f"{k} {n}" for k, n in sorted(rows.items(), key=lambda kv: -kv[1])[:8]))
print("  LDS: " + ", ".join(f"{k} {n}" for k, n in sorted(ops.items()) if k.startswith("ds_")))
