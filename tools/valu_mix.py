#!/usr/bin/env python3
"""Issue cost of a kernel's instruction mix: the VALU instructions of its two hot basic blocks (the opening stage and the
rollout loop body of K2o; tools/collect_profiles.py weights them by how often each runs) weighted by the per-instruction issue costs measured on gfx950 with tools/ubench.hip (4 waves per SIMD; committed as
profiles/r01_ubench_valu_issue.txt).  mix_cycles_per_instruction feeds bench.py's `valu_issue.mix_ceiling`: the rate
the VALU could sustain on THIS mix, as opposed to the guide's 2-cycle SIMD-32 peak that only plain VOP2 streams reach.

    python tools/valu_mix.py            (compiles connect_kernels.hip to ISA with hipcc, prints JSON)"""
import json, os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_blocks import blocks  # noqa: E402

# cycles per wave64 instruction per SIMD at 4 waves per SIMD (profiles/r01_ubench_valu_issue.txt); classes not
# measured individually take the cost of their class's measured member
COST = {
    "v_add_u32": 2.85, "v_sub_u32": 2.85, "v_subrev_u32": 2.85, "v_and_b32": 2.79, "v_or_b32": 2.79, "v_xor_b32": 2.79,
    "v_not_b32": 2.79, "v_mov_b32": 2.79, "v_lshrrev_b32": 2.58, "v_lshlrev_b32": 2.58, "v_ashrrev_i32": 2.58,
    "v_alignbit_b32": 4.46, "v_bcnt_u32_b32": 4.45, "v_mul_lo_u32": 4.73, "v_mul_hi_u32": 4.42, "v_bfe_u32": 4.25,
    "v_ffbl_b32": 4.23, "v_ffbh_u32": 4.23, "v_bitop3_b32": 4.06, "v_and_or_b32": 4.49, "v_or3_b32": 4.49,
    "v_lshl_add_u32": 4.55, "v_lshl_or_b32": 4.55, "v_add3_u32": 4.55, "v_add_lshl_u32": 4.55, "v_mad_u32_u24": 4.90,
    "v_mul_u32_u24": 4.90, "v_perm_b32": 4.75, "v_xad_u32": 4.73, "v_lshrrev_b64": 4.60, "v_lshlrev_b64": 4.73,
    "v_mad_u64_u32": 5.25, "v_lshl_add_u64": 4.85, "v_mov_b64": 4.39, "v_cndmask_b32": 3.5, "v_readfirstlane_b32": 4.0,
    "v_mbcnt_lo_u32_b32": 4.45, "v_mbcnt_hi_u32_b32": 4.45, "v_add_co_u32": 4.0, "v_addc_co_u32": 4.0,
}
CMP_COST = 3.5  # v_cmp_* (+ its v_cndmask partner: 7.09 per pair measured)
DEFAULT = 4.4   # unlisted VOP3


def block_mix(body):
    ops = Counter()
    for line in body:
        op = line.split()[0]
        if op.startswith("v_"):
            ops[re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)] += 1
    total = sum(ops.values())
    cycles = sum(n * (CMP_COST if op.startswith("v_cmp") else COST.get(op, DEFAULT)) for op, n in ops.items())
    return {"valu_instructions": total, "mix_cycles_per_instruction": cycles / total, "top_opcodes": dict(ops.most_common(12))}


def mix(path, sym):
    """The two big basic blocks of K2o: the opening stage (run once per 64 games) and the 4-ply loop body."""
    big = sorted(blocks(path, sym), key=lambda nb: -len(nb[1]))[:2]
    opening, loop = (block_mix(big[0][1]), block_mix(big[1][1]))
    if opening["valu_instructions"] < loop["valu_instructions"]:
        opening, loop = loop, opening
    return {"opening_block": opening, "loop_body": loop,
            # (kept for readers of round-2 files: the loop body's figures under the old names)
            "valu_instructions_in_loop_body": loop["valu_instructions"],
            "mix_cycles_per_instruction": loop["mix_cycles_per_instruction"]}


if __name__ == "__main__":
    csrc = os.path.join(ROOT, "board-game-simulator-python_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "connect.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                               os.path.join(csrc, "connect_kernels.hip"), "-o", out, "-Wno-unused-function", "-mllvm", "-enable-post-misched=0"],
                              stderr=subprocess.DEVNULL)
        # the bench kernel: Connect4(6,7,4), uncapped, from the initial state, 3 opening blocks, outcome codes fused
        # (template arguments: geometry, opening blocks, codes, RNG contract [, drain merge]: a prefix of the mangled name)
        sym = "_ZN3bgs12_GLOBAL__N_124k_connect_rollout_openedINS0_3GeoILi1ELi6ELi7ELi4EEELi3ELb1ELb0E"
        print(json.dumps(mix(out, sym), indent=1))
