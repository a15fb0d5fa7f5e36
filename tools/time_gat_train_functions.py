"""HIP-event time of nn._GatAggregateHeads forward / backward inside the UN-PROFILED mag training loop
(tools/profile_mag_train.py 4 64): per call group and per launch of the last group."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "cugraph-gnn_amd")]
import torch
from wholegraph_amd import nn
ev = []
orig_b = nn._GatAggregateHeads.backward
orig_f = nn._GatAggregateHeads.forward
def timed(tag, fn):
    def wrap(ctx, *a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = fn(ctx, *a); e.record(); ev.append((tag, s, e)); return out
    return staticmethod(wrap)
nn._GatAggregateHeads.backward = timed("bwd", orig_b.__func__ if hasattr(orig_b, "__func__") else orig_b)
nn._GatAggregateHeads.forward = timed("fwd", orig_f.__func__ if hasattr(orig_f, "__func__") else orig_f)
sys.argv = ["x", "4", "64"]
import runpy
t0 = time.perf_counter()
runpy.run_path(os.path.join(ROOT, "tools/profile_mag_train.py"), run_name="__main__")
torch.cuda.synchronize()
n_groups = 6
for tag in ("fwd", "bwd"):
    ms = [s.elapsed_time(e) for t, s, e in ev if t == tag]
    print(tag, "calls", len(ms), "total ms per group", sum(ms) / n_groups, "last group:", [round(v, 2) for v in ms[-11:]])
