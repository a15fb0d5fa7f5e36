"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of gpurun_prof.sh (FETCH_SIZE, WRITE_SIZE; values are KB per
dispatch).  gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte units where the fabric
moves 128-byte requests, so read bytes = FETCH_SIZE x 2 (checked on a pure copy kernel: 2 x FETCH + WRITE = 2 x bytes copied).
Writes profiles/<round>/pmc_traffic.json, which bench.py quotes as roofline.traffic for the same workload.

usage: python tools/pmc_summary.py profiles/r01 r01i
"""
import csv, json, os, re, sys
from collections import defaultdict

FETCH_CORRECTION = 2.0


def short(name):
    m = re.search(r"(\w+)<", name) or re.search(r"(\w+)\(", name)
    base = m.group(1) if m else name
    tmpl = re.search(r"<([^()]*)>\(", name)
    return base + ("<" + tmpl.group(1).replace(" ", "") + ">" if tmpl else "")


def load(path):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main(folder, tag):
    fetch = load(os.path.join(folder, "%s_FETCH_SIZE_counter_collection.csv" % tag))
    write = load(os.path.join(folder, "%s_WRITE_SIZE_counter_collection.csv" % tag))
    out = {"source": ["%s_FETCH_SIZE_counter_collection.csv" % tag, "%s_WRITE_SIZE_counter_collection.csv" % tag],
           "fetch_correction": FETCH_CORRECTION, "unit": "bytes per launch (mean over the launches of the pass)", "kernels": {}}
    for k in sorted(set(fetch) & set(write)):
        fk, wk = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        out["kernels"][k] = {"launches": len(fetch[k]), "fetch_KB": round(fk, 1), "write_KB": round(wk, 1),
                             "traffic_bytes": int((fk * FETCH_CORRECTION + wk) * 1024)}
    with open(os.path.join(folder, "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print("%-70s %4d launches  %8.1f MB" % (k[:70], v["launches"], v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
