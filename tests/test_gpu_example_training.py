"""The end-to-end training example (examples/sage_node_classification.py) learns: loader -> SAGEConv forward/backward ->
optimizer on the HIP kernels, loss goes down and the planted communities are recovered."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sage_example_learns(hiplib, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import sage_node_classification as ex
    monkeypatch.setattr(sys, "argv", ["x", "--nodes", "30000", "--epochs", "3", "--batch-size", "512", "--fanout", "10", "5"])
    loss, acc = ex.main()
    assert loss < 1.5 and acc > 0.6, (loss, acc)
