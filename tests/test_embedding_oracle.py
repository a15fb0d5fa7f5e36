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


def _reference_apply(kind, p, dtype, lr, indices, grads, embs, states, per_row):
    """The reference test's host optimizer followed STATEMENT BY STATEMENT in scalar Python on np.float32 values — `Apply` and
    `ApplyLazyAdam / ApplyAdaGrad / ApplyRMSProp / ApplySGD` of
    cpp/tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu:180-300 (the caller hands it de-duplicated
    indices with summed gradients, :450-481).  Deliberately NOT vectorised and NOT sharing a line with the oracle."""
    f = np.float32
    rt = (lambda v: f(np.float16(v))) if dtype == "half" else (lambda v: eo._round_trip(np.array([v], f), "bf16")[0]) \
        if dtype == "bf16" else (lambda v: v)
    beta1, beta2, eps, alpha, wd = (f(p[k]) for k in ("beta1", "beta2", "epsilon", "alpha", "weight_decay"))
    one = f(1)
    for i, index in enumerate(indices):
        grad_vec, emb_vec = grads[i], embs[index]
        if kind == "lazy_adam":
            beta1t, beta2t = per_row[0][index] * beta1, per_row[1][index] * beta2
            per_row[0][index], per_row[1][index] = beta1t, beta2t
        for d in range(len(emb_vec)):
            grad_value, emb_value = f(grad_vec[d]), f(emb_vec[d])
            if kind == "lazy_adam":
                if p["adam_w"] > 0.5:
                    emb_value = f(emb_value - f(f(lr * wd) * emb_value))
                else:
                    grad_value = f(grad_value + f(wd * emb_value))
                m = f(f(beta1 * states[0][index][d]) + f(f(one - beta1) * grad_value))
                v = f(f(beta2 * states[1][index][d]) + f(f(f(one - beta2) * grad_value) * grad_value))
                mhat, vhat = f(m / f(one - beta1t)), f(v / f(one - beta2t))
                emb_value = f(emb_value - f(f(lr * mhat) / f(f(np.sqrt(vhat)) + eps)))
                states[0][index][d], states[1][index][d] = m, v
            else:
                grad_value = f(grad_value + f(wd * emb_value))
                if kind == "adagrad":
                    state_sum = f(states[0][index][d] + f(grad_value * grad_value))
                    emb_value = f(emb_value - f(f(lr * grad_value) / f(f(np.sqrt(state_sum)) + eps)))
                    states[0][index][d] = state_sum
                elif kind == "rmsprop":
                    v = f(f(alpha * states[0][index][d]) + f(f(f(one - alpha) * grad_value) * grad_value))
                    emb_value = f(emb_value - f(f(lr * grad_value) / f(f(np.sqrt(v)) + eps)))
                    states[0][index][d] = v
                else:
                    emb_value = f(emb_value - f(lr * grad_value))
            emb_vec[d] = rt(emb_value)


@pytest.mark.parametrize("dtype", ["float", "half", "bf16"])
@pytest.mark.parametrize("kind,params", [("sgd", {"weight_decay": 0.01}), ("lazy_adam", {}), ("lazy_adam", {"adam_w": 1.0, "weight_decay": 0.05}),
                                         ("adagrad", {"weight_decay": 0.01}), ("rmsprop", {"alpha": 0.9})])
def test_oracle_follows_the_reference_host_optimizer_statement_by_statement(kind, params, dtype):
    """The vectorised oracle against a scalar transcription of the reference test's `CPUOptimizer` (the class the reference
    holds its device kernels to): sparse, duplicate-heavy index sets, three steps, all table dtypes.  fp32 arithmetic is
    the same operations in the same order on both sides up to the association of `lr * x / y` (ulp-level), so values agree
    to a few ulp and the optimizer states too."""
    rng = np.random.default_rng(11)
    n, dim, lr = 40, 7, np.float32(0.05)
    table = eo._round_trip(rng.uniform(-4, 4, (n, dim)).astype(np.float32), dtype)
    ref = [list(r) for r in table.copy()]
    p = dict(eo.DEFAULTS)
    p.update(params)
    n_states = {"sgd": 0, "lazy_adam": 2, "adagrad": 1, "rmsprop": 1}[kind]
    states = [[[np.float32(0)] * dim for _ in range(n)] for _ in range(n_states)]
    per_row = [[np.float32(1)] * n, [np.float32(1)] * n]
    opt = eo.SparseOptimizer(kind, n, dim, table_dtype=dtype, **params)
    for step in range(3):
        idx = rng.integers(-1, n // 2, 25)                       # duplicates and "skip" entries (-1)
        grads = rng.uniform(-2, 2, (25, dim)).astype(np.float32)
        opt.step(table, idx, grads, lr)
        rows, summed = eo.dedup(idx, grads)                      # the reference's test de-duplicates the same way (:450-481)
        _reference_apply(kind, p, dtype, lr, rows.tolist(), summed, ref, states, per_row)
        tol = dict(rtol=3e-6, atol=1e-6) if dtype == "float" else dict(rtol=1e-2 if dtype == "bf16" else 1e-3, atol=1e-3)
        np.testing.assert_allclose(table, np.array(ref, np.float32), **tol)
    for s, name in enumerate(eo.STATE_NAMES[kind][:n_states]):
        np.testing.assert_allclose(opt.states[name], np.array(states[s], np.float32), rtol=3e-6, atol=1e-7)
    if kind == "lazy_adam":
        np.testing.assert_allclose(opt.states["beta12t"], np.array(per_row, np.float32).T, rtol=1e-6)
