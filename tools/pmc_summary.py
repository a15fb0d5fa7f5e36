"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of gpurun_prof.sh (FETCH_SIZE, WRITE_SIZE; values are KB per
dispatch).  gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte units where the fabric
moves 128-byte requests, so read bytes = FETCH_SIZE x 2 (checked on a pure copy kernel: 2 x FETCH + WRITE = 2 x bytes copied).
Writes profiles/<round>/pmc_traffic.json, which bench.py quotes as roofline.traffic for the same workload.

usage: python tools/pmc_summary.py profiles/r01 r01i [pmc_traffic_rmat26.json]
(third argument: output file name inside the folder; default pmc_traffic.json = the products workload of the driver's command)
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
    """{kernel: [values per dispatch, in dispatch order]}"""
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return {k: [v for _, v in sorted(vals)] for k, vals in acc.items()}


def split_shapes(fetch, write):
    """The same aggregation kernel serves layer 1 and layer 2 of a call group with the same grid: its launches fall into
    two traffic clusters a factor of several apart.  Launches whose (fetch + write) is at least half the largest one form
    the `#large` entry (= the layer-1 launch shape), so a per-launch figure is never an average over different shapes."""
    out = {}
    for k in set(fetch) & set(write):
        f, w = fetch[k], write[k]
        if len(f) != len(w) or len(f) < 4:
            continue
        tot = [a + b for a, b in zip(f, w)]
        big = [i for i, t in enumerate(tot) if t >= 0.5 * max(tot)]
        if 0 < len(big) < len(tot) and min(tot) < 0.4 * max(tot):
            out[k + "#large"] = ([f[i] for i in big], [w[i] for i in big])
    return out


def main(folder, tag, out_name="pmc_traffic.json"):
    fetch = load(os.path.join(folder, "%s_FETCH_SIZE_counter_collection.csv" % tag))
    write = load(os.path.join(folder, "%s_WRITE_SIZE_counter_collection.csv" % tag))
    out = {"source": ["%s_FETCH_SIZE_counter_collection.csv" % tag, "%s_WRITE_SIZE_counter_collection.csv" % tag],
           "fetch_correction": FETCH_CORRECTION, "unit": "bytes per launch (mean over the launches of the pass)", "kernels": {}}
    pairs = {k: (fetch[k], write[k]) for k in set(fetch) & set(write)}
    pairs.update(split_shapes(fetch, write))
    for k in sorted(pairs):
        f, w = pairs[k]
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        # (the two passes run the same command: launch i of one is launch i of the other, so the largest single launch is
        #  the largest pairwise sum — what a pipeline with many launch shapes per kernel quotes for its dominant launch)
        biggest = max(a * FETCH_CORRECTION + b for a, b in zip(f, w)) if len(f) == len(w) else None
        out["kernels"][k] = {"launches": len(f), "fetch_KB": round(fk, 1), "write_KB": round(wk, 1),
                             "traffic_bytes": int((fk * FETCH_CORRECTION + wk) * 1024),
                             "max_traffic_bytes": None if biggest is None else int(biggest * 1024)}
    with open(os.path.join(folder, out_name), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print("%-70s %4d launches  %8.1f MB" % (k[:70], v["launches"], v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
