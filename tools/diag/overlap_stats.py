#!/usr/bin/env python3
"""Occupancy of the device over time from a rocprofv3 --kernel-trace CSV of the headline mode (three streams): wall time of the trace's
steady part, time with at least one kernel running (union), the sum of kernel durations, time with two or more running - per forward - and
the kernels with the largest share of SOLO time (nothing beside them).  usage: overlap_stats.py <kernel_trace.csv> <forwards in the window>"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))
               if r["Kind"] == "KERNEL_DISPATCH"), key=lambda r: r[0])
# window: from the first to the last gather launch of the last N forwards
g = [i for i, r in enumerate(rows) if "::gather_" in r[2]]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lo, hi = rows[g[-N - 1]][0], rows[g[-1]][0]
ev = []
for s, e, n in rows:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        ev.append((s2, 1, n))
        ev.append((e2, -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
busy = multi = 0
solo = {}
active = {}
prev = lo
for t, d, n in ev:
    k = sum(active.values())
    if k >= 1:
        busy += t - prev
    if k >= 2:
        multi += t - prev
    if k == 1:
        nm = next(a for a, c in active.items() if c > 0)
        solo[nm] = solo.get(nm, 0) + (t - prev)
    active[n] = active.get(n, 0) + d
    prev = t
tot = sum(min(e, hi) - max(s, lo) for s, e, n in rows if min(e, hi) > max(s, lo))
wall = hi - lo
print(f"{N} forwards: wall {wall / N / 1e3:8.1f} us per forward, some kernel running {busy / N / 1e3:8.1f} ({busy / wall:.3f}), "
      f"two or more {multi / N / 1e3:8.1f}, sum of kernel durations {tot / N / 1e3:8.1f}, idle {(wall - busy) / N / 1e3:7.1f}")
print("solo time per forward (us), largest first:")
for n, v in sorted(solo.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / N / 1e3:8.1f}  {n[:120]}")
