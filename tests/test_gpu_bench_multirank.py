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


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(nproc, extra, env=None, shape=("200000", "3000000", "16")):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--nodes", shape[0], "--edges", shape[1], "--call-group", shape[2], "--groups-per-step", "2", "--no-variants"] + extra
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
    assert "dp2" in two["config"]["parallelism"] and "all-to-all" in two["config"]["parallelism"]
    # N > 1: the HEADLINE is the north-star path — range-partitioned table + all-to-all exchange; the collective-free replicated
    # placement is measured first (it cannot hang) and reported next to it
    assert two["headline_placement"] == "partitioned"
    assert set(two["placements"]) == {"replicated", "partitioned"} and two["placements"]["replicated"]["value"] > 0
    assert two["value"] == two["placements"]["partitioned"]["value"]
    assert two["all_to_all_bytes_per_gpu"] > 0 and two["xgmi_frac"] > 0
    # per-rank values (so that N = 1 can be compared with a rank of N > 1) and the pre-flight's verdict
    assert len(two["per_rank_value"]) == 2 and all(v > 0 for v in two["per_rank_value"])
    assert sum(two["per_rank_value"]) >= two["value"] * 0.999
    assert two["selftest"]["partitioned"]["bit_exact"] is True and two["selftest"]["partitioned"]["ids_per_rank"] == 100001
    assert "rccl_ranks" in two      # None here: the gloo rehearsal has no RCCL communicator


@pytest.mark.parametrize("dedup", ["0", "1"])
def test_eight_ranks_control_flow_at_reduced_size(hiplib, dedup):
    """The line the driver's N = 8 run will produce, rehearsed with eight processes on ONE GPU (gloo; RCCL refuses several ranks
    per device, and eight processes time-slice the chip, so speeds mean nothing here): every rank joins the pre-flight and both
    placements, the whole-job value is the sum over eight seed shards, and the partitioned fetch is exercised with the
    de-duplicated exchange forced on and off (WGAMD_GATHER_DEDUP).  World sizes 2-8 of the C-level exchange itself — both
    memory types, split communicators, empty partitions — run as threads over the RCCL stand-in in
    tests/test_gpu_comm_multirank.py."""
    d = _run(8, ["--dist-backend", "gloo", "--share-gpu"], env={"WGAMD_GATHER_DEDUP": dedup}, shape=("120000", "1500000", "4"))
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["headline_placement"] == "partitioned" and set(d["placements"]) == {"replicated", "partitioned"}
    assert "dp8" in d["config"]["parallelism"] and "all-to-all" in d["config"]["parallelism"]
    assert d["selftest"]["partitioned"]["bit_exact"] is True and d["selftest"]["partitioned"]["dedup_bit_exact"] is True
    assert d["selftest"]["partitioned"]["rows"] == 65537 * 8 + 3
    for name in ("replicated", "partitioned"):
        pr = d["placements"][name]["per_rank_value"]
        assert len(pr) == 8 and all(v > 0 for v in pr)
        # the whole-job value = all ranks' edges over the SLOWEST rank's time: never above the sum of the per-rank rates (how far
        # below says nothing here: eight processes time-slice one GPU and finish at very different times)
        assert 0 < d["placements"][name]["value"] <= sum(pr) * 1.001
    part = d["placements"]["partitioned"]
    assert part["requested_rows_per_call_group"] > 0 and part["wire_rows_per_call_group"] <= part["requested_rows_per_call_group"]
    assert part["all_to_all_bytes_per_gpu"] > 0 and d["xgmi_peak_GBps"] == 7 * 153.0
    assert d["call_group"] == 4 and d["batches_per_step"] == 8 and d["timed_call_groups"] == 6


def test_partitioned_feature_store_two_ranks(hiplib):
    """--feature-placement partitioned at N = 2: the all-to-all feature fetch over torch.distributed inside the pipeline."""
    d = _run(2, ["--dist-backend", "gloo", "--share-gpu", "--feature-placement", "partitioned"])
    assert "all-to-all" in d["config"]["parallelism"] and d["n_gpus"] == 2
    assert any(k.startswith("gather(") for k in d["stage_ms_per_call_group"])
    assert set(d["placements"]) == {"partitioned"} and d["all_to_all_bytes_per_gpu"] > 0 and d["headline_placement"] == "partitioned"


def test_stalled_extra_placement_does_not_cost_the_headline(hiplib):
    """One rank never enters the collectives of the also-measured partitioned pass: the watchdog prints the headline line
    (replicated placement) with the reason, every rank exits 0."""
    d = _run(2, ["--dist-backend", "gloo", "--share-gpu", "--extra-placement-timeout", "20"], env={"WGAMD_BENCH_TEST_STALL": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and "replicated" in d["config"]["parallelism"] or "dp2" in d["config"]["parallelism"]
    assert "timed out" in d["placement_errors"]["partitioned"] and set(d["placements"]) == {"replicated"}
    assert d["headline_placement"] == "replicated"


def test_rank_dying_in_the_extra_placement_does_not_cost_the_headline(hiplib):
    """A rank that DIES in the also-measured partitioned pass (a GPU fault is not an exception): the launcher sends SIGTERM to
    the others and rank 0 answers with the headline line it already has (the launcher itself then reports the failed worker)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--nodes", "200000", "--edges", "3000000", "--call-group", "16", "--groups-per-step", "2", "--no-variants", "--dist-backend", "gloo",
           "--share-gpu"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WGAMD_BENCH_TEST_DIE="1"))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "died" in d["placement_errors"]["partitioned"]
    assert d["headline_placement"] == "replicated"


def test_single_rank_rccl_exchange_with_selftest(hiplib):
    """--force-partitioned at N = 1: the library's DISTRIBUTED handle over a REAL (single-rank) RCCL communicator, with the
    known-answer pre-flight; the line carries what RCCL itself reports (ncclCommCount)."""
    d = _run(1, ["--force-partitioned", "--selftest", "--no-cpu-baseline"])
    assert d["headline_placement"] == "partitioned" and d["rccl_ranks"] == 1
    assert d["selftest"]["partitioned"]["bit_exact"] is True and d["value"] > 0


def test_a_failing_selftest_is_loud(hiplib):
    """A pre-flight that does not come back bit-exact takes the placement out (here: the only one) — error line, exit code 1."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--nodes", "100000", "--edges", "1000000",
           "--call-group", "8", "--groups-per-step", "1", "--no-variants", "--force-partitioned", "--selftest", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WGAMD_BENCH_TEST_CORRUPT="1"))
    assert p.returncode == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert d["value"] is None and "selftest" in d["placement_errors"]["partitioned"]


def _run_bare(gpus, extra, env=None):
    """`python bench.py --gpus N ...` with NO launcher and no WORLD_SIZE: the shape of the driver's N = 1 command."""
    envd = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    envd.update(env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=envd)


def test_bare_gpus_2_launches_two_ranks(hiplib):
    """`--gpus` was parsed and never read, so a bare `python bench.py --gpus 8` printed a one-rank line.
    Now the bare command re-executes itself under torch.distributed.run with N ranks."""
    p = _run_bare(2, ["--nodes", "200000", "--edges", "3000000", "--call-group", "16", "--groups-per-step", "2", "--no-variants",
                      "--dist-backend", "gloo", "--share-gpu"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank_value"]) == 2 and "dp2" in d["config"]["parallelism"]


def test_world_size_disagreeing_with_gpus_fails_loudly(hiplib):
    p = _run_bare(4, ["--no-variants", "--dist-backend", "gloo", "--share-gpu"],
                  env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
