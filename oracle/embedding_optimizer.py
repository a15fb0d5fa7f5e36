"""CPU restatement of the sparse embedding optimizers (TEST INFRASTRUCTURE ONLY — see the package docstring).

Follows the CPU optimizer the reference's own test compares the device against
(/root/reference/cpp/tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu):
  * de-duplication of one step's (index, gradient) pairs, gradients of the same row summed in arrival order — :450-481
  * SGD :288-300, LazyAdam / AdamW :221-253, AdaGrad :254-270, RMSProp :271-287, all in fp32
  * the updated value is rounded through the table's dtype (half / bf16) after every step — :215-220
  * default hyper-parameters :368-372 (weight_decay 0, epsilon 1e-8, alpha 0.99, beta1 0.9, beta2 0.999)
Device formulas it must agree with: cpp/src/wholememory_ops/functions/embedding_optimizer_func.cu:203-214, :394-421,
:657-671, :867-881.

PARITY UNPINNED against the reference itself: it holds no golden vectors for these (its test draws random tables and
compares the device with the CPU class restated here) and cannot be built or run in this image; what
tests/test_embedding_oracle.py can and does check is this restatement against ``torch.optim.{SGD,Adam,AdamW,Adagrad,RMSprop}`` on CPU, which compute the same updates
when every row receives a gradient each step.
"""
import numpy as np

DEFAULTS = dict(weight_decay=0.0, epsilon=1e-8, alpha=0.99, beta1=0.9, beta2=0.999, adam_w=0.0)
STATE_NAMES = {"sgd": [], "lazy_adam": ["m", "v", "beta12t"], "adagrad": ["state_sum"], "rmsprop": ["v"]}


def _round_trip(x, dtype):
    """fp32 -> table dtype -> fp32 ("half" | "bf16" | "float")."""
    if dtype == "half":
        return x.astype(np.float16).astype(np.float32)
    if dtype == "bf16":  # round to nearest even on the upper 16 bits
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    return x.astype(np.float32)


def dedup(indices, grads):
    """Unique indices in first-arrival order and their summed gradients (sum in arrival order, fp32)."""
    indices = np.asarray(indices, dtype=np.int64)
    grads = np.asarray(grads, dtype=np.float32)
    keep = indices >= 0
    indices, grads = indices[keep], grads[keep]
    uniq, first, inverse = np.unique(indices, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")           # unique values in first-arrival order
    rank_of = np.empty_like(order)
    rank_of[order] = np.arange(order.size)
    slot = rank_of[inverse]
    summed = np.zeros((uniq.size, grads.shape[1]), dtype=np.float32)
    np.add.at(summed, slot, grads)                     # unbuffered: fp32 sums in arrival order
    return uniq[order], summed


class SparseOptimizer:
    """State + step over a full table held as fp32 (values always representable in the table dtype)."""

    def __init__(self, kind, n_rows, dim, table_dtype="float", **params):
        assert kind in STATE_NAMES
        self.kind, self.dtype = kind, table_dtype
        self.p = dict(DEFAULTS)
        self.p.update(params)
        f32 = np.float32
        self.states = {}
        if kind == "lazy_adam":
            self.states = {"m": np.zeros((n_rows, dim), f32), "v": np.zeros((n_rows, dim), f32),
                           "beta12t": np.ones((n_rows, 2), f32)}
        elif kind == "adagrad":
            self.states = {"state_sum": np.zeros((n_rows, dim), f32)}
        elif kind == "rmsprop":
            self.states = {"v": np.zeros((n_rows, dim), f32)}

    def step(self, table, indices, grads, lr):
        """In place on ``table`` (fp32 [n_rows, dim]); ``indices`` may repeat (gradients are summed first)."""
        f32 = np.float32
        p = {k: f32(v) for k, v in self.p.items()}
        lr = f32(lr)
        rows, g = dedup(indices, grads)
        if rows.size == 0:
            return
        x = table[rows].astype(f32)
        one = f32(1.0)
        if self.kind == "lazy_adam" and self.p["adam_w"] > 0.5:
            x = x - lr * p["weight_decay"] * x
        else:
            g = g + p["weight_decay"] * x
        if self.kind == "sgd":
            x = x - lr * g
        elif self.kind == "lazy_adam":
            b = self.states["beta12t"][rows] * np.array([p["beta1"], p["beta2"]], f32)
            self.states["beta12t"][rows] = b
            m = p["beta1"] * self.states["m"][rows] + (one - p["beta1"]) * g
            v = p["beta2"] * self.states["v"][rows] + (one - p["beta2"]) * g * g
            mhat = m / (one - b[:, 0:1])
            vhat = v / (one - b[:, 1:2])
            x = x - lr * mhat / (np.sqrt(vhat) + p["epsilon"])
            self.states["m"][rows], self.states["v"][rows] = m, v
        elif self.kind == "adagrad":
            s = self.states["state_sum"][rows] + g * g
            x = x - lr * g / (np.sqrt(s) + p["epsilon"])
            self.states["state_sum"][rows] = s
        else:
            v = p["alpha"] * self.states["v"][rows] + (one - p["alpha"]) * g * g
            x = x - lr * g / (np.sqrt(v) + p["epsilon"])
            self.states["v"][rows] = v
        table[rows] = _round_trip(x.astype(f32), self.dtype)
