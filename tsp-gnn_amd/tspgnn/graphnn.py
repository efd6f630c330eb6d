"""``GraphNN``: the typed-graph recurrent message-passing engine with the reference's
constructor / call contract (/root/reference/graphnn.py:4-272), running on hand-written HIP
kernels (libtspgnn.so) instead of TensorFlow ops.

Per time step and per variable v (graphnn.py:142-173), for each entry of loop[v]:
    y = states[var].h  ->  optional fun(y)  ->  optional msg MLP  ->  optional mat (x) y
the entries are concatenated on axis 1 and fed to v's LayerNorm-LSTM cell; every update reads
the OLD states (synchronous update, graphnn.py:143).  The adjacency product never touches a
dense matrix: matrices are kept as CSR (both orientations) on the device and multiplied by the
aggregation kernels; a 0/1 matrix with exactly two ones per row (the TSP EV matrix) takes the
gather path.
"""
from collections import namedtuple

import numpy as np
import torch

from . import _lib
from . import variables as V
from .instance_loader import SparseEV
from .mlp import Mlp

# Field order of tf.contrib.rnn.LSTMStateTuple is (c, h); the reference constructs it by keyword
# (graphnn.py:138).
LSTMStateTuple = namedtuple("LSTMStateTuple", ("c", "h"))

LN_GATES = ("input", "transform", "forget", "output", "state")


def _dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


class DeviceAdjacency(object):
    """A sparse [R, C] matrix resident on the device in CSR, both orientations."""

    def __init__(self, shape, device, csr, csr_t, uv=None):
        self.shape = tuple(shape)
        self.device = device
        self.csr = csr      # (rowptr[R+1], col[nnz], val[nnz] or None)   rows of the matrix
        self.csr_t = csr_t  # same for the transpose
        self.uv = uv        # int32 [R,2] if the matrix is 0/1 with exactly two ones per row

    @staticmethod
    def from_sparse_ev(ev, device):
        rowptr, eid = ev.csr_by_vertex()
        M, N = ev.shape
        uv = _dev_i32(ev.uv, device)
        csr = (torch.arange(0, 2 * M + 1, 2, dtype=torch.int32, device=device), uv.view(-1), None)
        csr_t = (_dev_i32(rowptr, device), _dev_i32(eid, device), None)
        return DeviceAdjacency((M, N), device, csr, csr_t, uv=uv)

    @staticmethod
    def from_dense(mat, device):
        a = mat.detach().cpu().numpy() if torch.is_tensor(mat) else np.asarray(mat)
        if a.ndim != 2:
            raise ValueError("adjacency matrix must be 2-D")
        try:
            return DeviceAdjacency.from_sparse_ev(SparseEV.fromdense(a), device)
        except ValueError:
            pass

        def one(m):
            r, c = np.nonzero(m)
            vals = m[r, c].astype(np.float32)
            rowptr = np.zeros(m.shape[0] + 1, dtype=np.int64)
            np.cumsum(np.bincount(r, minlength=m.shape[0]), out=rowptr[1:])
            pattern = bool(np.all(vals == 1.0))
            v = None if pattern else torch.from_numpy(vals).to(device)
            return (_dev_i32(rowptr, device), _dev_i32(c, device), v)

        return DeviceAdjacency(a.shape, device, one(a), one(np.ascontiguousarray(a.T)))

    @staticmethod
    def wrap(mat, device):
        if isinstance(mat, DeviceAdjacency):
            return mat
        if isinstance(mat, SparseEV):
            return DeviceAdjacency.from_sparse_ev(mat, device)
        return DeviceAdjacency.from_dense(mat, device)

    def matmul(self, y, transpose=False):
        """mat (x) y  or  mat^T (x) y  (tf.matmul(..., adjoint_a=transpose), graphnn.py:156-160)."""
        R, C = self.shape
        rows_in = R if transpose else C
        rows_out = C if transpose else R
        if y.shape[0] != rows_in:
            raise ValueError("matrix/embedding size mismatch: %d vs %d" % (rows_in, y.shape[0]))
        d = y.shape[1]
        if d % 4 != 0:
            raise NotImplementedError("aggregation kernels need d %% 4 == 0 (got %d)" % d)
        out = torch.empty((rows_out, d), dtype=torch.float32, device=y.device)
        st = _lib.current_stream()
        if not transpose and self.uv is not None:
            _lib.call("tspgnn_gather2_sum_f32", _lib.ptr(self.uv), _lib.ptr(y), _lib.ptr(out), R, C, d, st)
            return out
        rowptr, col, val = self.csr_t if transpose else self.csr
        if val is None:
            _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(y), _lib.ptr(out),
                      rows_out, rows_in, d, st)
        else:
            _lib.call("tspgnn_csr_spmm_f32", _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.ptr(y),
                      _lib.ptr(out), rows_out, rows_in, d, st)
        return out


class LayerNormBasicLSTMCell(object):
    """tf.contrib.rnn.LayerNormBasicLSTMCell(num_units, activation=relu) with its defaults
    (forget_bias=1, layer_norm=True, gain 1, shift 0, no dropout) -- graphnn.py:107-112.
    Variables: <scope>/layer_norm_basic_lstm_cell/kernel [dx+d,4d] (glorot uniform, TF's default
    initialiser), .../{input,transform,forget,output,state}/{gamma,beta}."""

    def __init__(self, num_units, input_size, scope, activation="relu", store=None):
        if getattr(activation, "__name__", activation) != "relu":
            raise NotImplementedError("LayerNormBasicLSTMCell: only activation=relu has a HIP kernel")
        self.d, self.dx = int(num_units), int(input_size)
        if self.d not in (32, 64, 128):
            raise NotImplementedError("LSTM width %d: HIP kernels exist for 32, 64, 128" % self.d)
        if self.dx % 16 != 0:
            raise NotImplementedError("LSTM input width %d must be a multiple of 16" % self.dx)
        self.store = store if store is not None else V.get_default_store()
        self.base = "%s/layer_norm_basic_lstm_cell" % scope
        self.store.declare(self.base + "/kernel", (self.dx + self.d, 4 * self.d), V.xavier_uniform)
        for g in LN_GATES:
            self.store.declare("%s/%s/gamma" % (self.base, g), (self.d,), V.ones_init)
            self.store.declare("%s/%s/beta" % (self.base, g), (self.d,), V.zeros_init)

    def kernel(self):
        return self.store.view(self.base + "/kernel")

    def kernel_packed(self):
        """The kernel in MFMA fragment order (tspgnn_pack_weights_f32), cached per weight version."""
        def build(out):
            K = self.kernel()
            if out is None:
                out = torch.empty_like(K)
            _lib.call("tspgnn_pack_weights_f32", _lib.ptr(K), _lib.ptr(out), self.dx + self.d, 4 * self.d,
                      0, _lib.current_stream())
            return out
        return self.store.packed(("lstm", self.base), build)

    def ln(self):
        return self.store.span(self.base + "/input/gamma", self.base + "/state/beta")

    def __call__(self, inputs, state):
        """Returns (new_h, LSTMStateTuple(new_c, new_h)) like the TF cell."""
        c, h = state.c, state.h
        rows = h.shape[0]
        if inputs.shape[0] != rows or inputs.shape[1] != self.dx:
            raise ValueError("cell input must be [%d,%d], got %s" % (rows, self.dx, tuple(inputs.shape)))
        x = inputs if inputs.is_contiguous() else inputs.contiguous()
        h_out = torch.empty_like(h)
        c_out = torch.empty_like(c)
        _lib.call("tspgnn_lnlstm_fwd_f32", _lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(self.kernel_packed()),
                  _lib.ptr(self.ln()), _lib.ptr(h_out), _lib.ptr(c_out), rows, self.d, _lib.current_stream())
        return h_out, LSTMStateTuple(c=c_out, h=h_out)


class GraphNN(object):
    def __init__(self, var, mat, msg, loop, MLP_depth=3, MLP_weight_initializer=None, MLP_bias_initializer=None,
                 RNN_cell=LayerNormBasicLSTMCell, Cell_activation="relu", Msg_activation="relu",
                 Msg_last_activation=None, float_dtype=torch.float32, name="GraphNN", store=None):
        """Same four dictionaries as the reference (graphnn.py:21-48):
        var: name -> embedding size;  mat: name -> (row var, column var or int);
        msg: name -> (source var, target var);  loop: var -> list of update dicts with the
        optional keys 'mat', 'transpose?', 'fun', 'msg', 'var'."""
        self.var, self.mat, self.msg, self.loop, self.name = var, mat, msg, loop, name
        self.MLP_depth = MLP_depth
        # The reference initialises the message-MLP *biases* with the weight initialiser
        # (graphnn.py:121 passes MLP_weight_initializer() as bias_initializer).
        self.MLP_weight_initializer = MLP_weight_initializer or V.xavier_uniform
        self.MLP_bias_initializer = MLP_bias_initializer
        self.RNN_cell = RNN_cell
        self.Cell_activation = Cell_activation
        self.Msg_activation = Msg_activation
        self.Msg_last_activation = Msg_last_activation
        if float_dtype != torch.float32:
            raise NotImplementedError("GraphNN: the HIP path computes in fp32")
        self.float_dtype = float_dtype
        self.store = store if store is not None else V.get_default_store()
        self.check_model()
        self._init_parameters()

    # ---------------------------------------------------------------- static checks
    def check_model(self):
        """graphnn.py:72-103, same exception types and messages."""
        for v in self.var:
            if v not in self.loop:
                raise Warning("Variable {v} is not updated anywhere! Consider removing it from the model".format(v=v))
        for v in self.loop:
            if v not in self.var:
                raise Exception("Updating variable {v}, which has not been declared!".format(v=v))
        for mat, (v1, v2) in self.mat.items():
            if v1 not in self.var:
                raise Exception("Matrix {mat} definition depends on undeclared variable {v}".format(mat=mat, v=v1))
            if v2 not in self.var and type(v2) is not int:
                raise Exception("Matrix {mat} definition depends on undeclared variable {v}".format(mat=mat, v=v2))
        for msg, (v1, v2) in self.msg.items():
            if v1 not in self.var:
                raise Exception("Message {msg} maps from undeclared variable {v}".format(msg=msg, v=v1))
            if v2 not in self.var:
                raise Exception("Message {msg} maps to undeclared variable {v}".format(msg=msg, v=v2))

    def _update_width(self, update):
        """Number of columns one loop entry contributes to the cell input."""
        if "var" in update:
            width = self.var[update["var"]]
            if "msg" in update:
                width = self.var[self.msg[update["msg"]][1]]
            return width
        v2 = self.mat[update["mat"]][1]
        if type(v2) is not int:
            raise NotImplementedError("a loop entry without 'var' needs a matrix with an integer second "
                                      "dimension (graphnn.py:163-165)")
        return v2

    def _init_parameters(self):
        """graphnn.py:105-126: message MLPs ([d_in]*depth + [d_out], relu, xavier weights AND
        biases) and one LayerNorm-LSTM cell per variable."""
        self._msg_MLPs = {}
        for msg, (vin, vout) in self.msg.items():
            self._msg_MLPs[msg] = Mlp(
                layer_sizes=[self.var[vin] for _ in range(self.MLP_depth)],
                output_size=self.var[vout],
                activations=[self.Msg_activation for _ in range(self.MLP_depth)],
                output_activation=self.Msg_last_activation,
                kernel_initializer=self.MLP_weight_initializer,
                bias_initializer=self.MLP_weight_initializer,
                name="%s/%s" % (self.name, msg),
                name_internal_layers=True,
                input_size=self.var[vin],
                store=self.store,
            )
        self._RNN_cells = {}
        for v, d in self.var.items():
            dx = sum(self._update_width(u) for u in self.loop[v])
            self._RNN_cells[v] = self.RNN_cell(d, dx, "%s/%s_cell" % (self.name, v),
                                               activation=self.Cell_activation, store=self.store)

    # ---------------------------------------------------------------- run-time checks
    def check_run(self, adjacency_matrices, initial_embeddings, time_steps, LSTM_initial_states):
        """graphnn.py:185-271 (tf.assert_equal -> ValueError with the reference's messages)."""
        num_vars = {}
        for v, d in self.var.items():
            shape = tuple(initial_embeddings[v].shape)
            num_vars[v] = shape[0]
            if shape[1] != d:
                raise ValueError("Initial embedding of variable {v} doesn't have the same dimensionality {d} as "
                                 "declared".format(v=v, d=d))
            if v in LSTM_initial_states:
                ls = tuple(LSTM_initial_states[v].shape)
                if ls[1] != d:
                    raise ValueError("Initial hidden state of variable {v}'s LSTM doesn't have the same "
                                     "dimensionality {d} as declared".format(v=v, d=d))
                if ls != shape:
                    raise ValueError("Initial embeddings of variable {v} don't have the same shape as the its "
                                     "LSTM's initial hidden state".format(v=v))
        for mat, (v1, v2) in self.mat.items():
            ms = tuple(adjacency_matrices[mat].shape)
            if ms[0] != num_vars[v1]:
                raise ValueError("Matrix {m} doesn't have the same number of nodes as the initial embeddings of "
                                 "its variable {v}".format(v=v1, m=mat))
            if type(v2) is int:
                if ms[1] != v2:
                    raise ValueError("Matrix {m} doesn't have the same dimensionality {d} on the second variable "
                                     "as declared".format(m=mat, d=v2))
            elif ms[1] != num_vars[v2]:
                raise ValueError("Matrix {m} doesn't have the same number of nodes as the initial embeddings of "
                                 "its variable {v}".format(v=v2, m=mat))

    # ---------------------------------------------------------------- forward
    def __call__(self, adjacency_matrices, initial_embeddings, time_steps, LSTM_initial_states={}):
        """-> {var: LSTMStateTuple(c, h)} after ``time_steps`` synchronous steps
        (graphnn.py:128-183).  Embeddings are fp32 device tensors; matrices may be SparseEV,
        DeviceAdjacency, or dense numpy / torch arrays (converted once per call)."""
        self.check_run(adjacency_matrices, initial_embeddings, time_steps, LSTM_initial_states)
        some = next(iter(initial_embeddings.values()))
        device = some.device
        mats, dense_mats = {}, {}
        for v in self.var:
            for update in self.loop[v]:
                m = update.get("mat")
                if m is None:
                    continue
                if "var" in update and m not in mats:
                    mats[m] = DeviceAdjacency.wrap(adjacency_matrices[m], device)
                elif "var" not in update and m not in dense_mats:
                    a = adjacency_matrices[m]
                    if isinstance(a, (SparseEV, DeviceAdjacency)):
                        raise NotImplementedError("a matrix appended as a cell input must be dense")
                    dense_mats[m] = torch.as_tensor(a, dtype=torch.float32).to(device).contiguous()
        states = {}
        for v, init in initial_embeddings.items():
            h0 = init.to(torch.float32).contiguous()
            c0 = torch.zeros_like(h0) if v not in LSTM_initial_states \
                else LSTM_initial_states[v].to(torch.float32).contiguous()
            states[v] = LSTMStateTuple(c=c0, h=h0)
        for _ in range(int(time_steps)):
            new_states = {}
            for v in self.var:
                inputs = []
                for update in self.loop[v]:
                    if "var" in update:
                        y = states[update["var"]].h
                        if "fun" in update:
                            y = update["fun"](y)
                        if "msg" in update:
                            y = self._msg_MLPs[update["msg"]](y)
                        if "mat" in update:
                            y = mats[update["mat"]].matmul(y, transpose=update.get("transpose?", False))
                        inputs.append(y)
                    else:
                        inputs.append(dense_mats[update["mat"]])
                x = inputs[0] if len(inputs) == 1 else torch.cat(inputs, dim=1)
                _, new_states[v] = self._RNN_cells[v](x, states[v])
            states = new_states
        return states
