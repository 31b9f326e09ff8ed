#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, summed per (previous kernel -> next kernel)
pair: where the stream sits empty while the host reads a count back or prepares the next launch.
usage: gap_report.py kernel_trace.csv [min_gap_us]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last sort only: from its k_pack to the last kernel of the library
packs = [i for i, r in enumerate(rows) if "k_pack" in r[2]]
if packs:
    rows = rows[packs[-1]:]
    last = max(i for i, r in enumerate(rows) if "k_induce" in r[2] or "k_chain" in r[2])
    rows = rows[:last + 1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:48]


gaps = defaultdict(lambda: [0, 0.0])
busy = 0.0
idle = 0.0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    busy += (e0 - s0) / 1e3
    g = (s1 - e0) / 1e3
    if g > 50000:  # between sorts (verification, text generation)
        continue
    if g > min_gap:
        k = (short(n0), short(n1))
        gaps[k][0] += 1
        gaps[k][1] += g
        idle += g
print("span %.2f ms" % ((rows[-1][1] - rows[0][0]) / 1e6))
busy += (rows[-1][1] - rows[-1][0]) / 1e3
print("kernels %d  busy %.1f ms  idle (gaps > %.1f us, < 50 ms) %.1f ms" % (len(rows), busy / 1e3, min_gap, idle / 1e3))
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us  %5d x %6.1f us   %s -> %s" % (t, c, t / c, k[0], k[1]))
