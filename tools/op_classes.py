"""Per-class table of a `bench.py --dump-ops` CSV: convolutions / weight gradients by (pixels, Wp, Cin, Cout, mode), the
bandwidth ops by (op, H, C).     usage: python tools/op_classes.py ops.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
cls = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    iv = [int(x) for x in r["i"].split()]
    op = r["op"]
    if op in ("CONV3X3", "WGRAD"):
        key = (op, f"npix={iv[0]} Wp={iv[1]} {iv[2]}->{iv[5]} mode={iv[9] if op == 'CONV3X3' else '-'}")
    else:
        iv += [0] * 4
        key = (op, f"B={iv[0]} H={iv[1]} W={iv[2]} C={iv[3]}")
    c = cls[key]
    c[0] += 1
    c[1] += float(r["ms"])
    c[2] += float(r["gflop"])
tot = defaultdict(float)
for (op, k), c in cls.items():
    tot[op] += c[1]
for (op, k), c in sorted(cls.items(), key=lambda kv: (-tot[kv[0][0]], -kv[1][1])):
    if tot[op] < 0.25:
        continue
    tf = c[2] / c[1] if c[1] > 0 else 0.0
    print(f"{op:18s} {k:44s} n={c[0]:4d}  {c[1]:7.3f} ms  {1e3 * c[1] / c[0]:7.1f} us/launch  {tf:7.1f} TFLOP/s")
