"""Pins oracle/embedding_optimizer.py (the CPU restatement the GPU parity tests check against) to an independent
implementation: ``torch.optim`` on CPU computes the same updates when every row gets a gradient every step.
The reference has no golden vectors for this path (its test draws random tables and compares the device with its own CPU
optimizer, cpp/tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu:170-300)."""
import numpy as np
import pytest
import torch

from oracle import embedding_optimizer as eo


def _torch_opt(kind, param, lr, p):
    if kind == "sgd":
        return torch.optim.SGD([param], lr=lr, weight_decay=p.get("weight_decay", 0.0))
    if kind == "lazy_adam":
        cls = torch.optim.AdamW if p.get("adam_w", 0.0) > 0.5 else torch.optim.Adam
        return cls([param], lr=lr, betas=(p.get("beta1", 0.9), p.get("beta2", 0.999)), eps=p.get("epsilon", 1e-8),
                   weight_decay=p.get("weight_decay", 0.0))
    if kind == "adagrad":
        return torch.optim.Adagrad([param], lr=lr, eps=p.get("epsilon", 1e-8), weight_decay=p.get("weight_decay", 0.0))
    return torch.optim.RMSprop([param], lr=lr, alpha=p.get("alpha", 0.99), eps=p.get("epsilon", 1e-8),
                               weight_decay=p.get("weight_decay", 0.0))


@pytest.mark.parametrize("kind,params", [
    ("sgd", {}), ("sgd", {"weight_decay": 0.01}),
    ("lazy_adam", {}), ("lazy_adam", {"beta1": 0.8, "beta2": 0.9, "weight_decay": 0.02}),
    ("lazy_adam", {"adam_w": 1.0, "weight_decay": 0.05}),
    ("adagrad", {}), ("adagrad", {"weight_decay": 0.01, "epsilon": 1e-6}),
    ("rmsprop", {}), ("rmsprop", {"alpha": 0.9, "weight_decay": 0.01}),
])
def test_oracle_matches_torch_optim_when_every_row_is_touched(kind, params):
    rng = np.random.default_rng(7)
    n, dim, lr = 64, 19, 0.1
    table = rng.uniform(-10, 10, (n, dim)).astype(np.float32)
    param = torch.nn.Parameter(torch.from_numpy(table.copy()))
    topt = _torch_opt(kind, param, lr, params)
    opt = eo.SparseOptimizer(kind, n, dim, **params)
    for step in range(5):
        idx = rng.permutation(n)
        grads = rng.uniform(-5, 5, (n, dim)).astype(np.float32)
        opt.step(table, idx, grads, lr)
        dense = np.zeros_like(grads)
        dense[idx] = grads
        param.grad = torch.from_numpy(dense)
        topt.step()
        np.testing.assert_allclose(table, param.detach().numpy(), rtol=2e-5, atol=2e-5)


def test_dedup_sums_duplicates_in_arrival_order_and_skips_negative():
    idx = np.array([5, 2, 5, -1, 2, 5, 9])
    g = np.arange(7 * 3, dtype=np.float32).reshape(7, 3)
    rows, summed = eo.dedup(idx, g)
    assert rows.tolist() == [5, 2, 9]
    np.testing.assert_array_equal(summed[0], g[0] + g[2] + g[5])
    np.testing.assert_array_equal(summed[1], g[1] + g[4])
    np.testing.assert_array_equal(summed[2], g[6])


def test_lazy_rows_keep_their_own_step_count():
    """A row touched for the first time at step 3 is bias-corrected with t = 1 (per-row powers, not a global step)."""
    n, dim = 4, 8
    a = np.ones((n, dim), np.float32)
    opt = eo.SparseOptimizer("lazy_adam", n, dim)
    g = np.full((1, dim), 0.5, np.float32)
    for _ in range(2):
        opt.step(a, np.array([0]), g, 0.1)
    opt.step(a, np.array([1]), g, 0.1)
    b = np.ones((n, dim), np.float32)
    fresh = eo.SparseOptimizer("lazy_adam", n, dim)
    fresh.step(b, np.array([1]), g, 0.1)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(opt.states["beta12t"][0], [0.9 ** 2, 0.999 ** 2], rtol=1e-6)
    np.testing.assert_array_equal(opt.states["beta12t"][2], [1.0, 1.0])


@pytest.mark.parametrize("dtype", ["half", "bf16"])
def test_low_precision_tables_round_after_every_step(dtype):
    x = np.array([[1.0009765625, 3.14159]], np.float32)
    tdt = torch.float16 if dtype == "half" else torch.bfloat16
    want = torch.from_numpy(x).to(tdt).float().numpy()
    np.testing.assert_array_equal(eo._round_trip(x, dtype), want)
    rng = np.random.default_rng(0)
    big = rng.uniform(-100, 100, (1000,)).astype(np.float32)
    np.testing.assert_array_equal(eo._round_trip(big, dtype), torch.from_numpy(big).to(tdt).float().numpy())
