#!/usr/bin/env python3
"""Copy a rocprofv3 *_kernel_stats.csv from gpurun_out/ into profiles/ with shortened kernel names.
usage: tools/save_profile.py <src kernel_stats.csv> <dst name under profiles/> "<comment line>" """
import csv
import re
import sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]


src, dst, comment = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.reader(open(src)))
with open(f"profiles/{dst}", "w") as f:
    f.write(f"# {comment}\n")
    w = csv.writer(f)
    for r in rows:
        if r and r[0] != "Name":
            r[0] = short(r[0])
        w.writerow(r)
print("wrote", dst)
