#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory and prints the device timeline of the backprojection
launches: per kernel name count / mean duration, and for the dominant backprojection kernel the idle gaps between consecutive
launches and how much other work ran inside them (are the fused launches back to back?).

  python tools/timeline.py <dir with *_kernel_trace.csv>
"""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
copies = []
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?")))
rows.sort()


def short(n):
    """kernel name without return type, namespaces and argument list"""
    n = n.replace("(anonymous namespace)::", "")
    n = n[5:] if n.startswith("void ") else n
    return n.split("(")[0][:90]


by = {}
for s, e, n in rows:
    key = short(n)
    c = by.setdefault(key, [0, 0])
    c[0] += 1
    c[1] += e - s
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
print("device span %.3f ms, %d kernel launches" % ((t1 - t0) / 1e6, len(rows)))
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%8d x %10.1f us = %9.3f ms  %s" % (c, t / c / 1e3, t / 1e6, k))
if copies:
    dirs = {}
    for s, e, k in copies:
        c = dirs.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e - s
    for k, (c, t) in dirs.items():
        print("copies %-28s %6d x %8.1f us = %9.3f ms" % (k, c, t / c / 1e3, t / 1e6))
dom = max((k for k in by if "bp_" in k), key=lambda k: by[k][1], default=None)
if dom:
    bp = [(s, e) for s, e, n in rows if short(n) == dom]
    gaps = [bp[i + 1][0] - bp[i][1] for i in range(len(bp) - 1)]
    busy = sum(e - s for s, e in bp)
    print("dominant: %s" % dom)
    print("  %d launches, busy %.3f ms of a %.3f ms span from its first start to its last end (%.1f %%)" % (
        len(bp), busy / 1e6, (bp[-1][1] - bp[0][0]) / 1e6, 100.0 * busy / max(1, bp[-1][1] - bp[0][0])))
    print("  first launch starts %.3f ms after the first kernel of the process" % ((bp[0][0] - t0) / 1e6))
    if gaps:
        gaps_ms = sorted(g / 1e6 for g in gaps)
        print("  gaps between launches: median %.3f ms, max %.3f ms, sum %.3f ms" % (gaps_ms[len(gaps_ms) // 2], gaps_ms[-1], sum(gaps_ms)))
        print("  durations (ms): " + " ".join("%.2f" % ((e - s) / 1e6) for s, e in bp[:24]))
        print("  gaps (ms):      " + " ".join("%.2f" % (g / 1e6) for g in gaps[:24]))
