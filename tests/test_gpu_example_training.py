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


def test_link_prediction_example_learns(hiplib, monkeypatch):
    """examples/sage_link_prediction.py: LinkNeighborLoader (call groups, binary negatives) -> SAGE encoder -> dot-product
    decoder separates the planted intra-community edges from random negatives."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import sage_link_prediction as ex
    monkeypatch.setattr(sys, "argv", ["x", "--nodes", "8000", "--epochs", "3", "--batch-size", "256"])
    loss, acc = ex.main()
    assert loss < 0.6 and acc > 0.7, (loss, acc)


def test_call_group_training_example_learns(hiplib, monkeypatch):
    """examples/sage_call_group_training.py: the call-group loop — lazy features, one-kernel SAGE layers forward and backward,
    one optimizer step per call group — recovers the planted communities."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import sage_call_group_training as ex
    monkeypatch.setattr(sys, "argv", ["x", "--nodes", "30000", "--epochs", "6", "--batch-size", "256", "--group", "4",
                                      "--fanout", "10", "5"])
    loss, acc = ex.main()
    assert loss < 1.5 and acc > 0.6, (loss, acc)


def test_per_mini_batch_training_example_learns(hiplib, monkeypatch):
    """The same example with `--per-batch`: one Adam step per mini-batch of 256 seeds (the reference's semantics) through
    loader.PerBatchStep — sample per call group, one staging launch + one HIP-graph replay per step, a ragged last mini-batch."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import sage_call_group_training as ex
    monkeypatch.setattr(sys, "argv", ["x", "--nodes", "30000", "--epochs", "3", "--batch-size", "256", "--group", "4",
                                      "--fanout", "10", "5", "--per-batch"])
    loss, acc = ex.main()
    assert loss < 1.5 and acc > 0.6, (loss, acc)


def test_hetero_gat_call_group_example_learns_and_lazy_equals_gathered(hiplib, monkeypatch):
    """examples/hetero_gat_call_groups.py: HeteroConv{GATConv} over heterogeneous call groups — trains relation by relation
    (autograd), infers through the one-kernel relations reading the feature tables through the node lists; the class signal
    lives in the AUTHORS, so beating chance means the author -> paper relation is used."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import hetero_gat_call_groups as ex
    monkeypatch.setattr(sys, "argv", ["x", "--papers", "12000", "--authors", "6000", "--epochs", "6", "--batch-size", "256",
                                      "--group", "2"])
    loss, acc, same_bits = ex.main()
    assert same_bits and acc > 0.4 and loss < 1.7, (loss, acc, same_bits)       # (chance: 0.125, ln 8 = 2.08)
