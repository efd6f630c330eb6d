"""``Mlp``: the dense block of the hot path with the reference's constructor and call
signature (/root/reference/mlp.py:3-64), executed by hand-written HIP kernels.

Supported layer patterns (everything the reference instantiates; anything else raises
NotImplementedError rather than falling back to a generic slow path):
  * square chains d -> d -> ... -> d, d in {32,64,128}   (message MLPs, graphnn.py:114-125)
        -> tspgnn_mlp_fwd_f32 (MFMA, weights resident in LDS)
  * a square chain followed by Dense(1)                   (E_vote, model.py:107-115)
        -> tspgnn_mlp_fwd_f32 + tspgnn_rowdot_f32
  * 2 -> d/8 -> d/4 -> d/2 -> d with relu x3 + linear     (E_init_MLP, model.py:33-43)
        -> tspgnn_einit_fwd_f32
"""
import ctypes

import torch

from . import _lib
from . import variables as V


def _is_relu(act):
    if act is None:
        return False
    if act == "relu":
        return True
    name = getattr(act, "__name__", "")
    if name == "relu":
        return True
    raise NotImplementedError("Mlp activation %r: only relu and None have HIP kernels" % (act,))


def _bf16_flag(acts):
    """tspgnn_mlp_bwd_task.acts_bf16: the saved activations are the bf16 arrays of a bf16-storage tape."""
    return 1 if (acts is not None and acts.dtype == torch.bfloat16) else 0


def wgrad(X, dY, rows, kin, nout, gW, gb, ws):
    """gW[kin, nout] += X^T dY, gb += colsum(dY) (gb None: skip) over ``rows`` rows; X fp32, or a bf16 tape array read
    as it is (tspgnn_wgrad_bf16x_f32: widths in multiples of 64)."""
    if X.dtype == torch.bfloat16:
        if kin % 64 == 0 and nout % 64 == 0:
            _lib.call("tspgnn_wgrad_bf16x_f32", _lib.ptr(X), _lib.ptr(dY), rows, kin, nout, _lib.ptr(gW), _lib.ptr(gb),
                      _lib.ptr(ws), _lib.current_stream())
            return
        X = X.to(torch.float32)    # (narrow widths have no bf16-reading reduction: widened once)
    _lib.call("tspgnn_wgrad_f32", _lib.ptr(X), _lib.ptr(dY), rows, kin, nout, _lib.ptr(gW), _lib.ptr(gb), _lib.ptr(ws),
              _lib.current_stream())


class Mlp(object):
    def __init__(self, layer_sizes, output_size=None, activations=None, output_activation=None, use_bias=True,
                 kernel_initializer=None, bias_initializer=V.zeros_init, kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 trainable=True, name=None, name_internal_layers=True, input_size=None, store=None):
        """Stacks len(layer_sizes) dense layers, plus one of ``output_size`` units if given
        (mlp.py:22-55).  ``input_size`` is needed up front because variables are allocated at
        construction, not lazily at first call like tf.layers.Dense."""
        if not use_bias:
            raise NotImplementedError("Mlp(use_bias=False) has no HIP kernel")
        if any(x is not None for x in (kernel_regularizer, bias_regularizer, activity_regularizer,
                                       kernel_constraint, bias_constraint)):
            raise NotImplementedError("regularizers / constraints are not part of the hot path")
        if name is None or not name_internal_layers:
            raise ValueError("Mlp needs a name: layers are called <name>_MLP_layer_<i> (mlp.py:37)")
        layer_sizes = list(layer_sizes)
        if not isinstance(activations, list):
            activations = [activations for _ in layer_sizes]
        if output_size is not None:
            layer_sizes = layer_sizes + [output_size]
            activations = activations + [output_activation]
        # tf.layers.Dense does int(units): model.py:34 passes d/8, d/4, d/2 as floats
        self.sizes = [int(s) for s in layer_sizes]
        self.relu = [_is_relu(a) for a in activations]
        self.name = name
        self.store = store if store is not None else V.get_default_store()
        self.input_size = int(input_size) if input_size is not None else self.sizes[0]
        kinit = kernel_initializer if kernel_initializer is not None else V.xavier_uniform
        binit = bias_initializer if bias_initializer is not None else V.zeros_init
        self.layer_names = []
        fan_in = self.input_size
        for i, size in enumerate(self.sizes):
            lname = "%s_MLP_layer_%d" % (name, i + 1)
            self.store.declare(lname + "/kernel", (fan_in, size), kinit)
            self.store.declare(lname + "/bias", (size,), binit)
            self.layer_names.append(lname)
            fan_in = size
        self._plan = self._make_plan()

    # ------------------------------------------------------------------ kernel selection
    def _make_plan(self):
        dims = [self.input_size] + self.sizes
        d = dims[1] if len(dims) > 1 else None
        if all(x == dims[0] for x in dims) and dims[0] in (32, 64, 128):
            return ("square", dims[0], len(self.sizes), False)
        if len(dims) >= 3 and dims[-1] == 1 and all(x == dims[0] for x in dims[:-1]) and dims[0] in (32, 64, 128):
            if self.relu[-1]:
                raise NotImplementedError("Dense(1) head with an activation")
            return ("square", dims[0], len(self.sizes) - 1, True)
        dd = dims[-1]
        if dims == [2, int(dd / 8), int(dd / 4), int(dd / 2), dd] and dd in (32, 64, 128) \
                and self.relu == [True, True, True, False]:
            return ("einit", dd, 4, False)
        raise NotImplementedError("Mlp with layer widths %s has no HIP kernel (supported: square d->d chains, "
                                  "a square chain + Dense(1), and 2->d/8->d/4->d/2->d)" % (dims,))

    def wb(self, first=0, last=None):
        """Flat [W,b,W,b,...] slice of theta for layers first..last (inclusive)."""
        last = len(self.layer_names) - 1 if last is None else last
        return self.store.span(self.layer_names[first] + "/kernel", self.layer_names[last] + "/bias")

    def wb_packed(self, first, last, d):
        """[pack(W),b,...] for square layers first..last: the weights in MFMA fragment order
        (tspgnn_pack_weights_f32), cached until the variables change."""
        def build(out):
            src = self.wb(first, last)
            if out is None:
                out = torch.empty_like(src)
            st = _lib.current_stream()
            per = d * d + d
            for j in range(last - first + 1):
                o = j * per
                _lib.call("tspgnn_pack_weights_f32", _lib.ptr(src[o:o + d * d]), _lib.ptr(out[o:o + d * d]), d, d, 0, st)
                out[o + d * d:o + per].copy_(src[o + d * d:o + per])
            return out
        return self.store.packed(("mlp", self.name, first, last), build)

    def wb_packed_split(self, arith, first, last, d):
        """Byte tensor of {split pack(W), bias [d*4 B]} per square layer first..last for the split-operand kernels,
        cached until the variables change.  ``arith`` = "x3": three bf16 pieces (3*d*d*2 B, tspgnn_pack_weights_x3);
        "h2": two fp16 pieces of 2^s W (2*d*d*2 B, tspgnn_pack_weights_h2) next to 2^s b."""
        nb = {"x3": 6, "h2": 4}[arith]

        def build(out):
            src = self.wb(first, last)
            per_src, per = d * d + d, nb * d * d + 4 * d
            if out is None:
                out = torch.empty((last - first + 1) * per, dtype=torch.uint8, device=src.device)
            st = _lib.current_stream()
            if arith == "h2":      # every layer, weights and scaled biases, in one launch
                _lib.call("tspgnn_pack_mlp_h2", _lib.ptr(src), _lib.ptr(out), d, last - first + 1, 0,
                          self.store.h2_absmax_ptr(), st)
                return out
            for j in range(last - first + 1):
                o, q = j * per_src, j * per
                if arith == "h2":
                    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(src[o:o + d * d]), _lib.ptr(out[q:q + nb * d * d]), d, d,
                              self.store.h2_absmax_ptr(), st)
                else:
                    _lib.call("tspgnn_pack_weights_" + arith, _lib.ptr(src[o:o + d * d]), _lib.ptr(out[q:q + nb * d * d]), d, d, st)
                bias = src[o + d * d:o + per_src]
                if arith == "h2":
                    bias = bias * _lib.lib.tspgnn_h2_weight_scale()
                out[q + nb * d * d:q + per].copy_(bias.view(torch.uint8))
            return out
        return self.store.packed(("mlp." + arith, self.name, first, last), build)

    def wb_packed_x3(self, first, last, d):
        return self.wb_packed_split("x3", first, last, d)

    def wb_packed_bf16(self, first, last, d, interleave_last=False):
        """Byte tensor of {bf16 pack(W) [d*d*2 B] (weights rounded to bf16, fragment order), b [d*4 B]} per square
        layer first..last for the bf16-storage kernels, cached until the variables change.  ``interleave_last``: the
        last layer's output columns permuted for tspgnn_mlp_task_bf16.y_interleaved (packed column 16t+4g+j holds true
        column 32(t/2)+8g+4(t%2)+j)."""
        def build(out):
            src = self.wb_packed_x3(first, last, d)
            per3, per = 6 * d * d + 4 * d, 2 * d * d + 4 * d
            if out is None:
                out = torch.empty((last - first + 1) * per, dtype=torch.uint8, device=src.device)
            for j in range(last - first + 1):
                out[j * per:j * per + 2 * d * d].copy_(src[j * per3:j * per3 + 2 * d * d])            # piece 0
                out[j * per + 2 * d * d:(j + 1) * per].copy_(src[j * per3 + 6 * d * d:(j + 1) * per3])  # bias
            if interleave_last:
                j = last - first
                c = torch.arange(d, device=src.device)
                t, g, r = c // 16, (c % 16) // 4, c % 4
                perm = 32 * (t // 2) + 8 * g + 4 * (t % 2) + r
                wb = self.wb(last, last)
                W = wb[:d * d].view(d, d)[:, perm].contiguous()
                tmp = torch.empty(6 * d * d, dtype=torch.uint8, device=src.device)
                _lib.call("tspgnn_pack_weights_x3", _lib.ptr(W), _lib.ptr(tmp), d, d, _lib.current_stream())
                out[j * per:j * per + 2 * d * d].copy_(tmp[:2 * d * d])
                out[j * per + 2 * d * d:(j + 1) * per].copy_(wb[d * d:d * d + d][perm].contiguous().view(torch.uint8))
            return out
        return self.store.packed(("mlp.bf16" + (".il" if interleave_last else ""), self.name, first, last), build)

    def wt_packed(self, first, last, d):
        """pack(W_l^T) for square layers first..last back to back (data-gradient kernels)."""
        def build(out):
            if out is None:
                out = torch.empty((last - first + 1) * d * d, dtype=torch.float32, device=self.store.theta.device)
            st = _lib.current_stream()
            for j in range(last - first + 1):
                W = self.store.view(self.layer_names[first + j] + "/kernel")
                _lib.call("tspgnn_pack_weights_f32", _lib.ptr(W), _lib.ptr(out[j * d * d:(j + 1) * d * d]), d, d, 1, st)
            return out
        return self.store.packed(("mlpT", self.name, first, last), build)

    def wt_packed_h2(self, first, last, d):
        """tspgnn_pack_weights_h2(W_l^T) (two fp16 pieces of 2^s W_l^T, 4 d d bytes) for square layers first..last back
        to back: the data-gradient operand of tspgnn_mlp_bwd_rc_h2."""
        def build(out):
            if out is None:
                out = torch.empty((last - first + 1) * 4 * d * d, dtype=torch.uint8, device=self.store.theta.device)
            _lib.call("tspgnn_pack_mlp_h2", _lib.ptr(self.wb(first, last)), _lib.ptr(out), d, last - first + 1, 1,
                      self.store.h2_absmax_ptr(), _lib.current_stream())     # (max |2^s W| joins the store's range guard)
            return out
        return self.store.packed(("mlpT.h2", self.name, first, last), build)

    def recompute_ok(self, n_layers):
        """tspgnn_mlp_bwd_rc_h2 covers the first ``n_layers`` square layers of this Mlp (d = 64, one to three layers)."""
        kind, d, n_sq, head = self._plan
        return kind == "square" and d == 64 and 1 <= n_layers <= min(3, n_sq) and len(self._chunks()) == 1

    def backward_rc_task(self, n_layers, x, y_out, dY, dX, accumulate, gather_uv=None, partial=None):
        """An _lib.MlpBwdRcTask for the first ``n_layers`` square layers: the data gradient from the chain's input ``x``,
        its output ``y_out`` and the incoming gradient -- the hidden activations are recomputed (csrc/mlp_bwd_rc.hip) and
        the weight gradients formed in the same launch: they accumulate in ``partial`` (backward_rc_partial()) over the
        launches of a backward pass and are folded by backward_rc_finish()."""
        kind, d, n_sq, head = self._plan
        return _lib.MlpBwdRcTask(_lib.ptr(x), _lib.ptr(self.wb_packed_split("h2", 0, n_layers - 1, d)),
                                 _lib.ptr(self.wt_packed_h2(0, n_layers - 1, d)), _lib.ptr(y_out), _lib.ptr(dY),
                                 _lib.ptr(gather_uv), _lib.ptr(dX), 1 if accumulate else 0,
                                 dY.shape[0] if gather_uv is None else gather_uv.shape[0], n_layers,
                                 self.relu_mask(0, n_layers), None, 0, None, 0, _lib.ptr(partial))

    def backward_rc_partial(self, n_layers):
        kind, d, n_sq, head = self._plan
        n = int(_lib.lib.tspgnn_mlp_bwd_rc_partial_floats(d, n_layers))
        return torch.zeros(n, dtype=torch.float32, device=self.store.theta.device)

    def backward_rc_finish(self, n_layers, partial):
        """grad[W_0, b_0, .., W_{n-1}, b_{n-1}] += fixed-order sum of the workgroup partials."""
        kind, d, n_sq, head = self._plan
        g = self.store.grad_span(self.layer_names[0] + "/kernel", self.layer_names[n_layers - 1] + "/bias")
        _lib.call("tspgnn_mlp_bwd_rc_finish_f32", _lib.ptr(partial), _lib.ptr(g), d, n_layers, _lib.current_stream())

    @property
    def n_square(self):
        return self._plan[2] if self._plan[0] == "square" else 0

    def relu_mask(self, first, n):
        mask = 0
        for j in range(n):
            if self.relu[first + j]:
                mask |= 1 << j
        return mask

    def task(self, x, out, acts=None, acts_stride=0, proj=None, arith=None):
        """An _lib.MlpTask for a single-kernel square chain (None if this Mlp needs several kernels).
        ``proj`` = (packed [d,4d] matrix, output [rows,4d]): also emit out @ P from the same launch.
        ``arith``: "x3" / "h2" = weights in that split packing (for tspgnn_mlp_fwd_multi_x3 / _h2; proj packed
        likewise)."""
        kind, d, n_sq, head = self._plan
        if kind != "square" or head or len(self._chunks()) != 1:
            return None
        pw, po = (proj if proj is not None else (None, None))
        wb = self.wb_packed_split(arith, 0, n_sq - 1, d) if arith else self.wb_packed(0, n_sq - 1, d)
        return _lib.MlpTask(_lib.ptr(x), _lib.ptr(wb), _lib.ptr(out), _lib.ptr(acts),
                            acts_stride, x.shape[0], n_sq, self.relu_mask(0, n_sq), _lib.ptr(pw), _lib.ptr(po),
                            self.store.h2_flag_ptr() if arith == "h2" else None)

    def prefix_task(self, x, out, n_layers, arith=None, acts=None, acts_stride=0):
        """Task running only the first ``n_layers`` square layers (the rest is folded elsewhere); ``acts``: the hidden
        activations of those layers (all but the last, which is ``out``) are saved as by task()."""
        kind, d, n_sq, head = self._plan
        if kind != "square" or head or len(self._chunks()) != 1 or not (1 <= n_layers <= n_sq):
            return None
        wb = self.wb_packed_split(arith, 0, n_layers - 1, d) if arith else self.wb_packed(0, n_layers - 1, d)
        return _lib.MlpTask(_lib.ptr(x), _lib.ptr(wb), _lib.ptr(out), _lib.ptr(acts) if n_layers > 1 else None, acts_stride,
                            x.shape[0], n_layers, self.relu_mask(0, n_layers), None, None,
                            self.store.h2_flag_ptr() if arith == "h2" else None)

    def _chunks(self):
        kind, d, n_sq, head = self._plan
        step = 2 if d == 128 else 4   # layers whose weights fit LDS together
        return [(l0, min(step, n_sq - l0)) for l0 in range(0, n_sq, step)]

    def forward_saving(self, x, out, acts, acts_stride):
        """Square chain forward writing into caller-owned buffers: ``out`` [rows,d] and the hidden
        activations at acts + l*acts_stride (layer-major, so that all time steps of one layer are
        contiguous for the batched weight gradient).  ``acts`` is a tensor whose dim 0 indexes the layer."""
        kind, d, n_sq, head = self._plan
        if kind != "square":
            raise NotImplementedError("forward_saving: square Dense chains only")
        st, rows = _lib.current_stream(), x.shape[0]
        chunks = self._chunks()
        for l0, n in chunks:
            last = l0 + n == n_sq
            src = x if l0 == 0 else acts[l0 - 1]
            dst = out if last else acts[l0 + n - 1]
            _lib.call("tspgnn_mlp_fwd_f32", _lib.ptr(src), _lib.ptr(self.wb_packed(l0, l0 + n - 1, d)), _lib.ptr(dst),
                      _lib.ptr(acts[l0]) if n > 1 else None, acts_stride, rows, d, n, self.relu_mask(l0, n), st)
        return out

    def backward_data(self, dY, acts, acts_stride, y_out, dpre, dpre_stride, dX, accumulate, h2=False):
        """Data gradient of the square chain (tspgnn_mlp_bwd_f32; ``h2``: tspgnn_mlp_bwd_multi_h2), chunk by chunk in
        reverse."""
        kind, d, n_sq, head = self._plan
        st, rows = _lib.current_stream(), dY.shape[0]
        g = dY
        for l0, n in reversed(self._chunks()):
            last = l0 + n == n_sq
            first = l0 == 0
            yo = y_out if last else acts[l0 + n - 1]
            dst = dX if first else torch.empty_like(dY)
            wt = self.wt_packed_h2(l0, l0 + n - 1, d) if h2 else self.wt_packed(l0, l0 + n - 1, d)
            task = _lib.MlpBwdTask(_lib.ptr(g), _lib.ptr(wt),
                                   _lib.ptr(acts[l0]) if n > 1 else None, acts_stride, _lib.ptr(yo), _lib.ptr(dpre[l0]),
                                   dpre_stride, _lib.ptr(dst), 1 if (accumulate and first) else 0, rows, n,
                                   self.relu_mask(l0, n), None, _bf16_flag(acts))
            _lib.call_multi("tspgnn_mlp_bwd_multi_" + ("h2" if h2 else "f32"), [task], d)
            g = dst

    def backward_h2_ok(self, acts):
        """tspgnn_mlp_bwd_multi_h2 (data gradient on the fp16 matrix cores) covers this Mlp's square chain (width 64:
        one kernel; width 128: the two-layer chunks of backward_data); fp32 and bf16 tapes."""
        kind, d, n_sq, head = self._plan
        return kind == "square" and d in (64, 128)

    def backward_task(self, dY, acts, acts_stride, y_out, dpre, dpre_stride, dX, accumulate, gather_uv=None, h2=False,
                      pre=None):
        """An _lib.MlpBwdTask for a single-kernel chain (None if several kernels are needed).
        ``gather_uv`` (int32 [rows,2]): dY holds SOURCE rows and the chain starts from dY[u] + dY[v] per row.
        ``h2``: the task is for tspgnn_mlp_bwd_multi_h2 (weights in the f16x2 packing of W^T).
        ``pre`` = (X [rows, k], f16x2 packing of P^T [k, d]), h2 only: the chain starts from X P^T and dY is not read
        (tspgnn_mlp_bwd_task.pre_X)."""
        kind, d, n_sq, head = self._plan
        if len(self._chunks()) != 1:
            return None
        wt = self.wt_packed_h2(0, n_sq - 1, d) if h2 else self.wt_packed(0, n_sq - 1, d)
        if pre is not None:
            return _lib.MlpBwdTask(None, _lib.ptr(wt), _lib.ptr(acts), acts_stride, _lib.ptr(y_out), _lib.ptr(dpre),
                                   dpre_stride, _lib.ptr(dX), 1 if accumulate else 0, pre[0].shape[0], n_sq,
                                   self.relu_mask(0, n_sq), None, _bf16_flag(acts), _lib.ptr(pre[0]), _lib.ptr(pre[1]),
                                   pre[0].shape[1])
        return _lib.MlpBwdTask(_lib.ptr(dY), _lib.ptr(wt), _lib.ptr(acts), acts_stride,
                               _lib.ptr(y_out), _lib.ptr(dpre), dpre_stride, _lib.ptr(dX), 1 if accumulate else 0,
                               dY.shape[0] if gather_uv is None else gather_uv.shape[0], n_sq, self.relu_mask(0, n_sq),
                               _lib.ptr(gather_uv), _bf16_flag(acts))

    def backward_task_takes_projection(self, acts, k):
        """backward_task(..., pre=...) is available (tspgnn_mlp_bwd_multi_h2 at width 64, one kernel, fp32 tape)."""
        return self.sizes[-1] == 64 and len(self._chunks()) == 1 and self.backward_h2_ok(acts) and k % 32 == 0 \
            and 32 <= k <= 256 and acts.dtype == torch.float32

    def backward_prefix_task(self, n_layers, dY, acts, acts_stride, y_out, dpre, dpre_stride, dX, accumulate, gather_uv=None,
                             h2=False):
        """backward_task for the first ``n_layers`` square layers only (the rest was pushed elsewhere); ``y_out`` = the
        prefix's output (read when its last layer has relu)."""
        kind, d, n_sq, head = self._plan
        if len(self._chunks()) != 1 or not (1 <= n_layers <= n_sq):
            return None
        wt = self.wt_packed_h2(0, n_layers - 1, d) if h2 else self.wt_packed(0, n_layers - 1, d)
        return _lib.MlpBwdTask(_lib.ptr(dY), _lib.ptr(wt), _lib.ptr(acts), acts_stride,
                               _lib.ptr(y_out), _lib.ptr(dpre), dpre_stride, _lib.ptr(dX), 1 if accumulate else 0,
                               dY.shape[0] if gather_uv is None else gather_uv.shape[0], n_layers,
                               self.relu_mask(0, n_layers), _lib.ptr(gather_uv), _bf16_flag(acts))

    def backward_task_fuses_gather(self, dY):
        """backward_task(..., gather_uv=...) is available: one kernel covers the chain and dY is a plain fp32 array."""
        return len(self._chunks()) == 1 and dY.dtype == torch.float32 and dY.is_contiguous()

    def backward_weights(self, layer_inputs, layer_dpre, rows, n_layers=None):
        """dW_l += X_l^T dPre_l, db_l += colsum(dPre_l) for the square layers (the first ``n_layers`` of them); ``rows``
        may span all time steps (inputs / dpre are [T*rows_per_step, d] contiguous)."""
        kind, d, n_sq, head = self._plan
        ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows, d, d, device=layer_dpre[0].device)
        st = _lib.current_stream()
        for l in range(n_sq if n_layers is None else n_layers):
            name = self.layer_names[l]
            wgrad(layer_inputs[l], layer_dpre[l], rows, d, d, self.store.grad_view(name + "/kernel"),
                  self.store.grad_view(name + "/bias"), ws)

    # ------------------------------------------------------------------ forward
    def forward_split(self, x, arith="h2"):
        """Inference forward with the square layers on a split-operand kernel (``arith`` = "h2" / "x3": fp32-class
        accuracy on the fp16 / bf16 matrix cores, see csrc/dense_h2.hip, dense_x3.hip); falls back to __call__ for
        shapes those kernels do not cover."""
        kind, d, n_sq, head = self._plan
        if kind != "square" or d not in (32, 64) or len(self._chunks()) != 1 or x.dtype != torch.float32 \
                or not x.is_contiguous() or x.shape[1] != self.input_size:
            return self(x)
        last = self.layer_names[-1]
        flag = self.store.h2_flag_ptr() if arith == "h2" else None
        wb = _lib.ptr(self.wb_packed_split(arith, 0, n_sq - 1, d))
        if head and arith == "h2":     # Dense(1) taken in the same launch, the hidden rows never written
            y = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
            task = _lib.MlpTask(_lib.ptr(x), wb, None, None, 0, x.shape[0], n_sq, self.relu_mask(0, n_sq), None, None, flag)
            _lib.call("tspgnn_mlp_head_fwd_h2", ctypes.cast(ctypes.pointer(task), ctypes.c_void_p), _lib.ptr(self.store.view(last + "/kernel")),
                      _lib.ptr(self.store.view(last + "/bias")), _lib.ptr(y), d, _lib.current_stream())
            return y
        out = torch.empty((x.shape[0], d), dtype=torch.float32, device=x.device)
        task = _lib.MlpTask(_lib.ptr(x), wb, _lib.ptr(out), None, 0, x.shape[0], n_sq, self.relu_mask(0, n_sq), None, None,
                            flag)
        _lib.call_multi("tspgnn_mlp_fwd_multi_" + arith, [task], d)
        if not head:
            return out
        y = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
        _lib.call("tspgnn_rowdot_f32", _lib.ptr(out), _lib.ptr(self.store.view(last + "/kernel")),
                  _lib.ptr(self.store.view(last + "/bias")), _lib.ptr(y), x.shape[0], d, _lib.current_stream())
        return y

    def forward_x3(self, x):
        return self.forward_split(x, "x3")

    def __call__(self, inputs, save=None):
        """inputs: fp32 device tensor [rows, input_size].  ``save`` (optional list) receives the
        tensors a later backward needs."""
        if not self.store.finalized:
            raise RuntimeError("variables not allocated: create a Session (or store.finalize) first")
        x = inputs
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float32).contiguous()
        rows = x.shape[0]
        if x.shape[1] != self.input_size:
            raise ValueError("Mlp %s expects %d input features, got %d" % (self.name, self.input_size, x.shape[1]))
        st = _lib.current_stream()
        kind, d, n_sq, head = self._plan
        if kind == "einit":
            out = torch.empty((rows, d), dtype=torch.float32, device=x.device)
            _lib.call("tspgnn_einit_fwd_f32", _lib.ptr(x), _lib.ptr(self.wb()), _lib.ptr(out), rows, d, st)
            return out
        max_chunk = 2 if d == 128 else 4
        l0 = 0
        while l0 < n_sq:
            n = min(max_chunk, n_sq - l0)
            mask = 0
            for j in range(n):
                if self.relu[l0 + j]:
                    mask |= 1 << j
            out = torch.empty((rows, d), dtype=torch.float32, device=x.device)
            acts = None
            if save is not None and n > 1:
                acts = torch.empty((n - 1, rows, d), dtype=torch.float32, device=x.device)
            _lib.call("tspgnn_mlp_fwd_f32", _lib.ptr(x), _lib.ptr(self.wb_packed(l0, l0 + n - 1, d)), _lib.ptr(out),
                      _lib.ptr(acts), 0, rows, d, n, mask, st)
            if save is not None:
                save.append((l0, n, x, acts, out))
            x = out
            l0 += n
        if head:
            y = torch.empty((rows, 1), dtype=torch.float32, device=x.device)
            last = self.layer_names[-1]
            _lib.call("tspgnn_rowdot_f32", _lib.ptr(x), _lib.ptr(self.store.view(last + "/kernel")),
                      _lib.ptr(self.store.view(last + "/bias")), _lib.ptr(y), rows, d, st)
            if save is not None:
                save.append(("head", x))
            return y
        return x
