"""Concurrency summary of a rocprofv3 kernel trace (CSV): how much of the busy time has kernels of two queues in flight.
usage: python tools/trace_overlap.py <..._kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_lo = int(rows[int(len(rows) * skip)]["Start_Timestamp"])       # steady state: the last part of the run
ev = []
per_queue = defaultdict(int)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t_lo:
        continue
    q = r.get("Queue_Id", "0")
    per_queue[q] += e - s
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
active = defaultdict(int)
last, busy1, busy2, idle = ev[0][0], 0, 0, 0
for t, d, q in ev:
    nq = sum(1 for v in active.values() if v > 0)
    dt = t - last
    if nq == 0:
        idle += dt
    elif nq == 1:
        busy1 += dt
    else:
        busy2 += dt
    active[q] += d
    last = t
tot = busy1 + busy2 + idle
print("window %.1f ms: one queue busy %.1f%%, two or more queues busy %.1f%%, idle %.1f%%" %
      (tot / 1e6, 100 * busy1 / tot, 100 * busy2 / tot, 100 * idle / tot))
for q, v in sorted(per_queue.items()):
    print("  queue %s: kernel time %.1f ms (%.0f%% of the window)" % (q, v / 1e6, 100 * v / tot))
