"""Rehearsal of bench.py's N > 1 control flow on ONE GPU: two ranks launched exactly like the driver launches them
(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ...), sharing cuda:0 with the gloo backend (RCCL refuses two
ranks on one device).  Checks what the driver relies on: exactly one JSON line (from rank 0), whole-job aggregation over the
ranks, weak scaling fields, no CPU baseline at N > 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--nodes", "200000", "--edges", "3000000", "--call-group", "16", "--no-variants"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_control_flow(hiplib):
    one = _run(1, ["--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline"])
    two = _run(2, ["--dist-backend", "gloo", "--share-gpu"])
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
        assert d["unit"] == "sampled-edges/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
        assert d["roofline"]["frac"] > 0 and d["value"] > 0 and d["ms_per_step"] > 0
        assert "workload" in d["config"]
        # the launch shape is fixed by --call-group / --groups-per-step, never derived from --steps
        assert d["call_group"] == 16 and d["batches_per_step"] == 32 and d["timed_call_groups"] == 6
    assert two["cpu_baseline"] is None            # timed at N = 1 only
    # whole-job aggregate: both ranks' edges are counted (each rank samples its own seed shard of the same graph)
    e1 = sum(v for k, v in one["edges_per_batch"].items() if k.startswith("hop"))
    e2 = sum(v for k, v in two["edges_per_batch"].items() if k.startswith("hop"))
    assert abs(e1 - e2) / e1 < 0.1
    assert 1.6 < (two["value"] * two["ms_per_step"]) / (one["value"] * one["ms_per_step"]) < 2.4
    assert "dp2" in two["config"]["parallelism"]
    # N > 1 with a small table: the headline is the collective-free replicated placement, the partitioned (exchange)
    # result is measured in the same run and reported next to it
    assert set(two["placements"]) == {"replicated", "partitioned"} and two["placements"]["partitioned"]["value"] > 0
    assert two["all_to_all_bytes_per_gpu"] > 0 and two["xgmi_frac"] > 0


def test_partitioned_feature_store_two_ranks(hiplib):
    """--feature-placement partitioned at N = 2: the all-to-all feature fetch over torch.distributed inside the pipeline."""
    d = _run(2, ["--dist-backend", "gloo", "--share-gpu", "--feature-placement", "partitioned"])
    assert "all-to-all" in d["config"]["parallelism"] and d["n_gpus"] == 2
    assert any(k.startswith("gather(") for k in d["stage_ms_per_call_group"])
    assert d["placements"]["replicated"] is None and d["all_to_all_bytes_per_gpu"] > 0


def test_stalled_extra_placement_does_not_cost_the_headline(hiplib):
    """One rank never enters the collectives of the also-measured partitioned pass: the watchdog prints the headline line
    (replicated placement) with the reason, every rank exits 0."""
    d = _run(2, ["--dist-backend", "gloo", "--share-gpu", "--extra-placement-timeout", "20"], env={"WGAMD_BENCH_TEST_STALL": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and "replicated" in d["config"]["parallelism"] or "dp2" in d["config"]["parallelism"]
    assert "timed out" in d["placement_errors"]["partitioned"] and "placements" not in d


def test_rank_dying_in_the_extra_placement_does_not_cost_the_headline(hiplib):
    """A rank that DIES in the also-measured partitioned pass (a GPU fault is not an exception): the launcher sends SIGTERM to
    the others and rank 0 answers with the headline line it already has (the launcher itself then reports the failed worker)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--nodes", "200000", "--edges", "3000000", "--call-group", "16", "--no-variants", "--dist-backend", "gloo", "--share-gpu"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WGAMD_BENCH_TEST_DIE="1"))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "died" in d["placement_errors"]["partitioned"]
