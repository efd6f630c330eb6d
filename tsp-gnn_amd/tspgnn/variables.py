"""Trainable-variable store: the counterpart of the TF-1.x default graph's variable collection
that the reference relies on (tf.get_variable / tf.layers.Dense / tf.trainable_variables(),
model.py:47,163-167).

All variables live in ONE flat fp32 device buffer (``theta``) in declaration order, so that
(a) the kernels read whole weight packs (an MLP's W/b chain, an LSTM cell's kernel + LayerNorm
pairs) as contiguous slices without repacking, and (b) gradient all-reduce, global-norm clip
and Adam each run on one flat buffer.
"""
import contextlib
import math
from collections import OrderedDict

import numpy as np
import torch


def xavier_uniform(shape, gen):
    """tf.contrib.layers.xavier_initializer(): U(-l, l), l = sqrt(6/(fan_in+fan_out)); a 1-D
    shape uses fan_in = fan_out = shape[0] (what the reference gets for the message-MLP biases,
    graphnn.py:121)."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim


def zeros_init(shape, gen):
    return torch.zeros(shape, dtype=torch.float64)


def ones_init(shape, gen):
    return torch.ones(shape, dtype=torch.float64)


def normal_init(shape, gen):
    return torch.randn(shape, generator=gen, dtype=torch.float64)


class VariableStore(object):
    def __init__(self):
        self._decl = OrderedDict()  # name -> (shape, initializer)
        self._offsets = None
        self.theta = None
        self.device = None
        self.version = 0      # bumped whenever theta changes; keys the packed-weight cache
        self.assignments = 0  # bumped when theta is ASSIGNED from outside (initialise / restore): a data-parallel
                              # Session re-broadcasts from rank 0 before its next training step
        self._packed = {}
        self.grad = None      # flat gradient buffer, same layout as theta (allocated on first use)
        self._h2_guard = None
        self.h2_packs_pending = 0   # f16x2 weight packings enqueued since the guard's weight word was last read

    # -- f16x2 range guard (include/tspgnn.h, tspgnn_pack_weights_h2) ------------------------------------
    # int32[4] on the device: [0] bit 0 set by an f16x2 launch whose operand left the fp16 range (the tasks' range_flag,
    # also Adam's skip_flag); [1] IEEE bits of max |2^s W| over every f16x2 weight packing since it was last zeroed.
    H2_WEIGHT_LIMIT_BITS = 0x46ffe000   # 32752.0f: HALF of fp16's largest finite value -- margin for the steps a
                                        # training run takes between two looks at the word (Adam moves a weight by ~lr)

    def h2_guard(self):
        if self._h2_guard is None:
            self._h2_guard = torch.zeros(4, dtype=torch.int32, device=self.theta.device)
        return self._h2_guard

    def h2_flag_ptr(self):
        return self.h2_guard().data_ptr()

    def h2_absmax_ptr(self):
        self.h2_packs_pending += 1
        return self.h2_guard().data_ptr() + 4

    # -- declaration phase -------------------------------------------------------------
    def declare(self, name, shape, initializer):
        if self._offsets is not None:
            raise RuntimeError("variable store already finalised")
        if name in self._decl:
            raise ValueError("Variable %s already exists" % name)
        self._decl[name] = (tuple(int(s) for s in shape), initializer)
        return name

    def names(self):
        return list(self._decl.keys())

    def shape(self, name):
        return self._decl[name][0]

    @property
    def size(self):
        return sum(int(np.prod(s)) for s, _ in self._decl.values())

    # -- allocation --------------------------------------------------------------------
    def finalize(self, device):
        off, offsets = 0, OrderedDict()
        for name, (shape, _) in self._decl.items():
            n = int(np.prod(shape))
            off = (off + 3) // 4 * 4  # 16-byte alignment: the kernels read packs with float4 loads
            offsets[name] = (off, n)
            off += n
        self._offsets = offsets
        self.device = torch.device(device)
        self.theta = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.h2_guard()   # allocated (and zeroed) NOW: a first use inside a HIP-graph capture would capture the memset
        return self

    @property
    def finalized(self):
        return self._offsets is not None

    def initialize(self, seed=0):
        """tf.global_variables_initializer(): run every variable's initialiser (seeded CPU
        generator, then one upload)."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(seed))
        host = torch.zeros(self.theta.numel(), dtype=torch.float32)
        for name, (shape, init) in self._decl.items():
            off, n = self._offsets[name]
            host[off:off + n] = init(shape, gen).reshape(-1).to(torch.float32)
        self.theta.copy_(host)
        self.assignments += 1
        self.touch()

    def touch(self):
        """Call after modifying theta in place (initialise / restore / optimiser step)."""
        self.version += 1

    def packed(self, key, build):
        """Derived, kernel-ready copy of some variables (MFMA fragment order), rebuilt lazily when
        theta has changed since it was made.  ``build(out_or_None)`` returns the tensor."""
        ent = self._packed.get(key)
        if ent is None or ent[0] != self.version:
            t = build(None if ent is None else ent[1])
            self._packed[key] = (self.version, t)
            return t
        return ent[1]

    @contextlib.contextmanager
    def rounded_to_bf16(self, names):
        """Inside the block ``theta`` is a copy in which the listed variables are rounded to bf16 (nearest even) -- the
        GEMM weights the bf16-storage forward multiplied with -- and the packed-weight cache is a fresh one; the fp32
        master weights, their cache and the gradient buffer are untouched (mixed-precision training: the backward
        differentiates the function the forward evaluated, the optimiser updates the fp32 variables)."""
        saved = (self.theta, self._packed, self.version)
        theta = self.theta.clone()
        for name in names:
            off, n = self._offsets[name]
            theta[off:off + n] = theta[off:off + n].to(torch.bfloat16).to(torch.float32)
        self.theta, self._packed, self.version = theta, {}, self.version + 1
        try:
            yield self
        finally:
            self.theta, self._packed, self.version = saved

    def view(self, name):
        off, n = self._offsets[name]
        return self.theta[off:off + n].view(self._decl[name][0])

    def span(self, first, last):
        """Contiguous flat slice covering variables ``first`` .. ``last`` (inclusive)."""
        o0, _ = self._offsets[first]
        o1, n1 = self._offsets[last]
        return self.theta[o0:o1 + n1]

    def offset(self, name):
        return self._offsets[name][0]

    # -- gradients (same flat layout as theta: one buffer to all-reduce, clip and apply) ----
    # ``bucket`` = [grad (theta's layout) | BUCKET_TAIL floats]: the data-parallel step appends the local batch size
    # and the batch statistics to the gradient so that ONE all-reduce carries everything (Session.allreduce_grads).
    BUCKET_TAIL = 10   # B_r, B_r*loss, B_r*acc, TP, FP, TN, FN, guard bits 0 / 1 / 2 (csrc/train.hip, bucket_pack_kernel)

    def zero_grad(self):
        if self.grad is None:
            self.bucket = torch.zeros(self.theta.numel() + self.BUCKET_TAIL, dtype=torch.float32, device=self.theta.device)
            self.grad = self.bucket[:self.theta.numel()]
        else:
            self.bucket.zero_()
        return self.grad

    def grad_view(self, name):
        off, n = self._offsets[name]
        return self.grad[off:off + n].view(self._decl[name][0])

    def grad_span(self, first, last):
        o0, _ = self._offsets[first]
        o1, n1 = self._offsets[last]
        return self.grad[o0:o1 + n1]

    def grad_dict(self):
        host = self.grad.detach().cpu().numpy()
        return OrderedDict((name, host[off:off + n].reshape(self._decl[name][0]).copy())
                           for name, (off, n) in self._offsets.items())

    def load(self, values):
        """Assign variables from a {name: array} mapping (checkpoint restore / parity tests)."""
        host = self.theta.detach().cpu()
        for name, val in values.items():
            off, n = self._offsets[name]
            arr = np.asarray(val, dtype=np.float32).reshape(-1)
            if arr.size != n:
                raise ValueError("shape mismatch for %s: expected %s" % (name, self._decl[name][0]))
            host[off:off + n] = torch.from_numpy(arr.copy())
        self.theta.copy_(host)
        self.assignments += 1
        self.touch()

    def state_dict(self):
        host = self.theta.detach().cpu().numpy()
        return OrderedDict((name, host[off:off + n].reshape(self._decl[name][0]).copy())
                           for name, (off, n) in self._offsets.items())


# The "default graph": build_network()/Mlp()/GraphNN() declare into it unless given a store.
_default_store = VariableStore()


def get_default_store():
    return _default_store


def reset_default_store():
    """tf.reset_default_graph() analogue: start a fresh variable collection."""
    global _default_store
    _default_store = VariableStore()
    return _default_store
