#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing (loop bodies = blocks that branch backwards).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o ew.s -Iinclude fluidnet_amd/csrc/fnet_elementwise.hip
    python tools/isa_blocks.py ew.s k_inact_bwd_reduceItLi2E [min_instructions]
"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") or (key in l and ": ;" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    cur.append(t.split(";")[0].strip())
blocks.append((name, cur))
order = {n: k for k, (n, _) in enumerate(blocks)}
TR = ("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")
for k, (n, ins) in enumerate(blocks):
    if len(ins) < min_n:
        continue
    back = [t for i in ins if i.startswith(("s_cbranch", "s_branch")) for t in [i.split()[-1]] if t in order and order[t] <= k]
    c = {"valu": 0, "pk": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0, "mfma": 0, "wait": 0, "other": 0}
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith(TR):
            c["trans"] += 1
        elif op.startswith("v_pk_"):
            c["pk"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_barrier"):
            c["wait"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        else:
            c["other"] += 1
    print(f"{n:>12} n={len(ins):5d} {'LOOP' if back else '    '} " + " ".join(f"{a}={b}" for a, b in c.items() if b))
