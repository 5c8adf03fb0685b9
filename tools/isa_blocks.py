#!/usr/bin/env python3
"""Static instruction counts per basic block of one kernel in a hipcc -S listing.
usage: isa_blocks.py file.s kernel_symbol_prefix"""
import re, sys
from collections import Counter

def blocks(path, sym):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(("):", ":")) or (l.startswith(sym) and ":" in l))
    out, cur, name = [], [], "entry"
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            cur.append(t); break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            out.append((name, cur)); cur, name = [], m.group(1); continue
        if not l.startswith("\t") or t.startswith((".", ";")) or not t:
            continue
        cur.append(t)
    out.append((name, cur))
    return out

if __name__ == "__main__":
    for name, body in blocks(sys.argv[1], sys.argv[2]):
        c = Counter(x.split()[0] for x in body)
        valu = sum(n for k, n in c.items() if k.startswith("v_"))
        salu = sum(n for k, n in c.items() if k.startswith("s_") and not k.startswith(("s_nop", "s_waitcnt")))
        print(f"{name:12s} instr {len(body):4d}  valu {valu:4d}  salu {salu:4d}  mem {sum(n for k, n in c.items() if k.startswith(('global_', 'ds_', 'buffer_', 'flat_', 's_load'))):3d}")
