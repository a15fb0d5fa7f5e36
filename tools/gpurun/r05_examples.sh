cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_example_training.py -m gpu -q -x 2>&1 | tail -3
python examples/sage_call_group_training.py --epochs 2 2>&1 | tail -2
python examples/sage_node_classification.py --epochs 2 2>&1 | tail -2
