#!/usr/bin/env python3
"""How the two lanes share the chip: from a rocprofv3 kernel trace (t_kernel_trace.csv) of bench.py, for every kernel kind the time its
launches ran alone and the time they ran while a kernel of the other queue was running, and the slowdown of overlapped launches against
launches of the same kernel and grid that ran alone.
usage: tools/lane_overlap.py gpurun_out/prof_<tag>/t_kernel_trace.csv"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), name, (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["LDS_Block_Size"])))
rows.sort()
# the last training step: from the last k_gather_batch on
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_gather_batch")]
rows = rows[starts[-1]:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
queues = collections.Counter(r[2] for r in rows)
print("step span %.2f ms, launches per queue %s" % ((t1 - t0) / 1e6, dict(queues)))
main_q = queues.most_common(1)[0][0]
side = [(a, b) for (a, b, q, n, g) in rows if q != main_q]
main = [(a, b) for (a, b, q, n, g) in rows if q == main_q]


def overlap(a, b, ivs):
    tot = 0
    for (c, d) in ivs:
        if d <= a:
            continue
        if c >= b:
            break
        tot += min(b, d) - max(a, c)
    return tot


busy_side = sum(b - a for a, b in side)
busy_main = sum(b - a for a, b in main)
both = sum(overlap(a, b, side) for a, b in main)
print("main queue busy %.2f ms, side queue busy %.2f ms, both busy %.2f ms" % (busy_main / 1e6, busy_side / 1e6, both / 1e6))
per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
solo = collections.defaultdict(list)
ovl = collections.defaultdict(list)
for (a, b, q, n, g) in rows:
    o = overlap(a, b, side if q == main_q else main)
    e = per[(q != main_q, n)]
    e[0] += 1
    e[1] += (b - a) / 1e6
    e[2] += o / 1e6
    (solo if o < 0.1 * (b - a) else ovl if o > 0.8 * (b - a) else collections.defaultdict(list))[(n, g)].append(b - a)
print("%-50s %5s %9s %9s" % ("kernel (S = side queue)", "n", "total ms", "ovl ms"))
for (s, n), e in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-50s %5d %9.2f %9.2f" % (("S " if s else "  ") + n[:48], e[0], e[1], e[2]))
print("slowdown of fully overlapped launches against alone launches of the same kernel and grid (>= 2 samples each):")
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k in solo:
    if k in ovl and len(solo[k]) >= 2 and len(ovl[k]) >= 2:
        s_, o_ = sorted(solo[k])[len(solo[k]) // 2], sorted(ovl[k])[len(ovl[k]) // 2]
        a = agg[k[0]]
        a[0] += s_ * len(ovl[k]); a[1] += o_ * len(ovl[k]); a[2] += len(ovl[k])
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-48s alone %7.1f us  overlapped %7.1f us  x%.2f  (%d launches)" % (n[:48], a[0] / a[2] / 1e3, a[1] / a[2] / 1e3, a[1] / a[0], a[2]))
# idle gaps of the main queue (> 15 us): where lane 0 waits -- for a join with the side lane, or for the host
mrows = sorted([(a, b, n) for (a, b, q, n, g) in rows if q == main_q])
gaps = []
for (a0, b0, n0), (a1, b1, n1) in zip(mrows, mrows[1:]):
    if a1 - b0 > 15000:
        side_busy = overlap(b0, a1, side)
        gaps.append((a1 - b0, (b0 - t0) / 1e6, n0, n1, side_busy))
print("main-queue idle gaps > 15 us: %d, together %.2f ms (all gaps together %.2f ms)" % (
    len(gaps), sum(g[0] for g in gaps) / 1e6, sum(max(0, a1 - b0) for (a0, b0, n0), (a1, b1, n1) in zip(mrows, mrows[1:])) / 1e6))
for g in sorted(gaps, reverse=True)[:14]:
    print("  %7.1f us at %6.2f ms  after %-40s before %-40s side busy %5.1f us" % (g[0] / 1e3, g[1], g[2][:40], g[3][:40], g[4] / 1e3))
