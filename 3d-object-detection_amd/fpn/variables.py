"""Variable store: the stand-in for TF-1 variable scopes.

The reference creates its weights implicitly through `tf.get_variable` inside
`tf.variable_scope(scope)` blocks (models/tf_util.py:10-49).  Here a
`VariableStore` maps the same scoped names ('conv1/weights', 'conv1/biases',
'conv1/bn/gamma', ...) to device tensors; a layer asks the current store for its
variables and the store creates them on first use with the reference's
initialisers (Xavier-uniform weights :42, zero biases :105, batch-norm
gamma=1 / beta=0 / moving_mean=0 / moving_variance=1 as tf.contrib.layers
creates them).  Convolution kernels are held as (Cin,Cout); a TF checkpoint's
(1,1,Cin,Cout) arrays are reshaped on load.
"""
import contextlib
import math

import torch

BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon, models/tf_util.py:527-531


class VariableStore:
    def __init__(self, device="cuda", seed=1):
        self.device = torch.device(device)
        self._vars = {}
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(int(seed))
        self.version = 0      # bumped on every write; folded-parameter caches key on it
        self._cache = {}

    # -- dict-like access -------------------------------------------------
    def __contains__(self, name):
        return name in self._vars

    def __getitem__(self, name):
        return self._vars[name]

    def __setitem__(self, name, value):
        t = torch.as_tensor(value, dtype=torch.float32)
        if t.dim() == 4 and t.shape[0] == 1 and t.shape[1] == 1:   # TF conv kernel (1,1,Cin,Cout)
            t = t.reshape(t.shape[2], t.shape[3])
        self._vars[name] = t.to(self.device).contiguous()
        self.version += 1
        self._cache.clear()

    def keys(self):
        return self._vars.keys()

    def items(self):
        return self._vars.items()

    def numpy(self):
        """Host copy of every variable (what the CPU oracle consumes)."""
        return {k: v.detach().cpu().numpy() for k, v in self._vars.items()}

    def load(self, mapping):
        for k, v in mapping.items():
            self[k] = v
        return self

    # -- creation -----------------------------------------------------------
    def get_variable(self, name, shape, init):
        """Return the variable, creating it with `init` if absent
        ('xavier' | 'zeros' | 'ones' | float constant)."""
        if name in self._vars:
            v = self._vars[name]
            if tuple(v.shape) != tuple(shape):
                raise ValueError("variable %s has shape %s, layer expects %s"
                                 % (name, tuple(v.shape), tuple(shape)))
            return v
        if init == "xavier":
            fan_in, fan_out = shape[-2], shape[-1]
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=self._gen, dtype=torch.float32) * 2 - 1) * limit
        elif init == "zeros":
            t = torch.zeros(shape, dtype=torch.float32)
        elif init == "ones":
            t = torch.ones(shape, dtype=torch.float32)
        else:
            t = torch.full(shape, float(init), dtype=torch.float32)
        self[name] = t
        return self._vars[name]

    def randomize_batch_norm(self, seed=2):
        """Give every batch-norm variable a non-trivial value so that folding is
        exercised: gamma~U(0.5,1.5), beta~N(0,0.1), mean~N(0,0.1), var~U(0.5,1.5);
        biases ~N(0,0.1)."""
        g = torch.Generator(device="cpu")
        g.manual_seed(int(seed))
        for name in sorted(self._vars):
            shape = self._vars[name].shape
            if name.endswith("/bn/gamma") or name.endswith("/bn/moving_variance"):
                self[name] = torch.rand(shape, generator=g) + 0.5
            elif name.endswith("/bn/beta") or name.endswith("/bn/moving_mean") or name.endswith("/biases"):
                self[name] = torch.randn(shape, generator=g) * 0.1
        return self

    # -- derived, cached ------------------------------------------------------
    def cached(self, key, builder):
        hit = self._cache.get(key)
        if hit is None:
            hit = builder()
            self._cache[key] = hit
        return hit

    def folded(self, scope, bn):
        """(scale, shift) of the fused epilogue v = acc*scale + shift for one layer:
        bias_add then inference batch norm y = (x-mean)*gamma/sqrt(var+eps)+beta
        (models/tf_util.py:176-181) collapse to scale = gamma/sqrt(var+eps),
        shift = beta + (bias-mean)*scale; without batch norm scale is None
        (treated as 1) and shift = bias."""
        def build():
            bias = self._vars[scope + "/biases"]
            if not bn:
                return None, bias
            inv = self._vars[scope + "/bn/gamma"].double() / torch.sqrt(
                self._vars[scope + "/bn/moving_variance"].double() + BN_EPS)
            shift = self._vars[scope + "/bn/beta"].double() + \
                (bias.double() - self._vars[scope + "/bn/moving_mean"].double()) * inv
            return inv.float().contiguous(), shift.float().contiguous()
        return self.cached(("folded", scope, bool(bn)), build)


_default_store = None
_scope_stack = []


def get_default_store():
    """The store layers use when none is active (created on first use on the
    current ROCm device, like TF's default graph)."""
    global _default_store
    if _default_store is None:
        _default_store = VariableStore("cuda")
    return _default_store


def set_default_store(store):
    global _default_store
    _default_store = store
    return store


@contextlib.contextmanager
def use_store(store):
    global _default_store
    prev = _default_store
    _default_store = store
    try:
        yield store
    finally:
        _default_store = prev


@contextlib.contextmanager
def variable_scope(name):
    """Nested name prefix, the counterpart of tf.variable_scope."""
    _scope_stack.append(name)
    try:
        yield "/".join(_scope_stack)
    finally:
        _scope_stack.pop()


def scoped(name):
    return "/".join(_scope_stack + [name]) if _scope_stack else name
