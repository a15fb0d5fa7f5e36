cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_aggregate.py -x -q -k "fused" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_call_group'])"
