R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04e; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_renumber_gather.py tests/test_gpu_partitioned_csr.py tests/test_gpu_weighted_golden.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "^\.*s*\.* *\[" $OUT/pytest.log | tail -30
python - <<'PY'
import sys, time, torch
sys.path[:0]=["/root/repo","/root/repo/cugraph-gnn_amd","/root/repo/tests"]
import numpy as np
from graphgen import powerlaw_csr
from wholegraph_amd import GraphStructure
rp,col=powerlaw_csr(2_000_000, 30, seed=1, col_dtype=np.int64, max_deg=20000)
g=GraphStructure(); g.set_csr_graph(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda())
seeds=torch.randperm(2_000_000, device="cuda")[:1024]
for mode in ("captured","op_by_op"):
    g._captured_ok = mode=="captured"
    for i in range(20): g.multilayer_sample_without_replacement(seeds,[25,10],random_seeds=[i,i+1])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(200): out=g.multilayer_sample_without_replacement(seeds,[25,10],random_seeds=[i,i+1])
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/200
    print("L3 multilayer_sample_without_replacement %s: %.3f ms per batch (%d edges)"%(mode, dt*1e3, sum(int(c.shape[0]) for c in out[3])))
PY
bash gpurun_tune_lds.sh
