"""Per-kernel statistics of the DOMINANT launch shape from a rocprofv3 --kernel-trace CSV: a kernel that serves several
launch shapes in one run (call-group launches of the headline pipeline next to 1024-seed launches of the per-batch variant)
has a meaningless plain average; the launches lasting at least half as long as the kernel's longest one form the `large`
cluster, whose Calls / AverageNs / MinNs / MaxNs are written in the column layout of rocprofv3's own kernel_stats.csv.

usage: python tools/trace_large_launches.py <tag>_kernel_trace.csv out.csv
"""
import csv
import sys
from collections import defaultdict


def main(trace, out):
    dur = defaultdict(list)
    with open(trace, newline="") as f:
        for row in csv.DictReader(f):
            dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    rows = []
    for name, d in dur.items():
        big = [v for v in d if v >= 0.5 * max(d)]
        rows.append((sum(big), name, len(big), sum(big) / len(big), min(big), max(big), len(d)))
    rows.sort(reverse=True)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "AllCalls"])
        for tot, name, n, avg, lo, hi, n_all in rows:
            w.writerow([name, n, tot, "%.3f" % avg, lo, hi, n_all])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
