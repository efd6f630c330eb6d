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

import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import loop_plan
from . import resident_plan
from . import variables as V
from .instance_loader import SparseEV
from .mlp import Mlp, wgrad

# Field order of tf.contrib.rnn.LSTMStateTuple is (c, h); the reference constructs it by keyword
# (graphnn.py:138).
LSTMStateTuple = namedtuple("LSTMStateTuple", ("c", "h"))

LN_GATES = ("input", "transform", "forget", "output", "state")


# GraphNN.gemm -> suffix of the split-operand entry points (None: the fp32-MFMA kernels) and bytes per packed weight
GEMM_ARITH = {"f16x2": "h2", "bf16x3": "x3", "f32": None}
SPLIT_BYTES = {"x3": 6, "h2": 4}


def _pad16(rows):
    """Row count of a projected-message buffer of the f16x2 kernels (blocked by 16 source rows, include/tspgnn.h)."""
    return (int(rows) + 15) // 16 * 16


def to_storage(x, dtype, out=None):
    """``x`` (fp32) in the storage type of the embeddings: itself for fp32, rounded to bf16 (nearest even) by the library's
    own kernel on the device (tspgnn_convert_f32_to_bf16) -- torch only converts on the CPU plumbing path."""
    if dtype == torch.float32 or x.dtype == dtype:
        x = x.contiguous()
        if out is None:
            return x
        out.copy_(x)
        return out
    if dtype == torch.bfloat16 and x.is_cuda and x.dtype == torch.float32:
        x = x.contiguous()
        if out is None:
            out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        _lib.call("tspgnn_convert_f32_to_bf16", _lib.ptr(x), _lib.ptr(out), x.numel(), _lib.current_stream())
        return out
    y = x.to(dtype).contiguous()
    if out is None:
        return y
    out.copy_(y)
    return out


def to_f32(x):
    """Stored embeddings as fp32 (bf16 storage: widened once by tspgnn_convert_bf16_to_f32)."""
    if x.dtype == torch.float32:
        return x
    if x.dtype == torch.bfloat16 and x.is_cuda and x.is_contiguous():
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _lib.call("tspgnn_convert_bf16_to_f32", _lib.ptr(x), _lib.ptr(out), x.numel(), _lib.current_stream())
        return out
    return x.to(torch.float32)


def _pack_split(store, arith, W, out, krows, ncols):
    """tspgnn_pack_weights_x3 / _h2 of W [krows, ncols] into the byte tensor ``out``; an f16x2 packing also raises the
    store's range guard word to max |2^s W| (VariableStore.h2_guard)."""
    if arith == "h2":
        _lib.call("tspgnn_pack_weights_h2", _lib.ptr(W), _lib.ptr(out), krows, ncols, store.h2_absmax_ptr(),
                  _lib.current_stream())
    else:
        _lib.call("tspgnn_pack_weights_" + arith, _lib.ptr(W), _lib.ptr(out), krows, ncols, _lib.current_stream())


def _center_gates(K, d):
    """K [rows, 4d] with the mean over each gate's d columns subtracted per row (float64 arithmetic): LayerNorm's mean
    subtraction moved from every z row of every step into the weights (tspgnn_lstm_task.z_centered)."""
    k64 = K.to(torch.float64).reshape(K.shape[0], 4, d)
    return (k64 - k64.mean(dim=2, keepdim=True)).reshape(K.shape[0], 4 * d).to(torch.float32).contiguous()


def _dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


def loop_enabled():
    """The one-launch T-step loop (tspgnn_mp_loop_h2) is on unless TSPGNN_LOOP=0 (A/B runs, parity tests of the stepwise
    launches)."""
    return os.environ.get("TSPGNN_LOOP", "1") != "0"


def loop_kind():
    """Which one-launch form a batch may get: "auto" (default) = tspgnn_mp_loop_h2 (edge states in registers) for batches of
    <= loop_plan.max_edge_tiles() tiles per wavefront, tspgnn_mp_resident_h2 (edge states through memory, LDS ticket) for
    larger ones; TSPGNN_LOOP_KIND=loop / resident force one form (tests, A/B runs)."""
    return os.environ.get("TSPGNN_LOOP_KIND", "auto")


def choose_loop_plan(e_start, v_start, grid):
    """-> (plan int32 ndarray, meta) or None; meta = (n_groups, grid, kind, n_slots, lds_words, n_active), kind "loop"
    (loop_plan.py, tspgnn_mp_loop_h2) or "resident" (resident_plan.py, tspgnn_mp_resident_h2)."""
    kind = loop_kind()
    if kind in ("auto", "loop"):
        built = loop_plan.build(e_start, v_start, grid=grid)
        if built is not None:
            return built[0], (built[1], grid, "loop", 0, 0, 0)
    if kind in ("auto", "resident"):
        built = resident_plan.build(e_start, v_start, grid=grid)
        if built is not None and kind == "auto" and not resident_plan.in_auto_window(int(e_start[-1])):
            built = None
        if built is not None:
            return built["plan"], (built["n_groups"], grid, "resident", built["n_slots"], built["lds_words"],
                                   built["edge_wgs"] + built["vertex_wgs"])
    return None


def _make_loop_plan(ev, device):
    """Work plan of the one-launch T-step loop for a SparseEV on a GPU: built where the batch is packed (host side, cached
    per block structure), uploaded with the adjacency -- a captured forward then serves any batch copied into its buffers
    (DeviceBatch.copy_from copies the plan too).  None: no GPU, switched off, or the batch does not fit a resident
    design.  -> (int32 device tensor, n_groups, grid, kind, n_slots, lds_words, n_active)."""
    dev = torch.device(device)
    if dev.type != "cuda" or not loop_enabled() or ev.shape[0] == 0:
        return None
    blocks = getattr(ev, "blocks", None)
    if blocks is None:
        blocks = loop_plan.block_structure(ev.uv, ev.shape[1])
        if blocks is None:
            return None
    grid = torch.cuda.get_device_properties(dev).multi_processor_count
    grid -= grid % 8
    built = choose_loop_plan(blocks[0], blocks[1], grid)
    if built is None:
        return None
    plan, meta = built
    return (torch.from_numpy(plan).to(dev),) + meta


class _States(dict):
    """{var: LSTMStateTuple} as returned by GraphNN.__call__, carrying the scratch buffers its launch plan points at:
    the task structures hold raw device pointers, so the buffers must live as long as anything that may replay those
    launches -- a captured HIP graph keeps the outputs (Session.capture_forward's closure), hence the buffers, alive,
    independently of later calls that build other plans."""

    def __init__(self, states, keep):
        dict.__init__(self, states)
        self._keep = keep


class Tape(object):
    """What GraphNN.forward_train keeps for the backward pass: every step's states H, C ([T+1, rows, d]), cell inputs
    X (for a folded cell: the message y per SOURCE row, with ZX = y Kx), hidden MLP activations acts
    ([layers-1, T, rows, d]).  fp32 in the default mode; in the bf16-storage mode H, X, ZX, acts are the bf16 arrays
    the forward stored (C stays fp32) and the accessors below widen a step -- or a range of steps for the weight
    gradients -- to the fp32 operands the backward kernels take."""

    native = False   # bf16 tape consumed as it is by the bf16-reading backward kernels (no widened copies)

    def _f32(self, x):
        return x if (x.dtype == torch.float32 or self.native) else x.to(torch.float32)

    def h(self, v, t):
        return self._f32(self.H[v][t])

    def x(self, v, t):
        return self._f32(self.X[v][t])

    def zx(self, v, t):
        z = self.ZX[v][t]
        if z.dtype == torch.float32 or self.native:
            return z
        pad, w = z.shape     # bf16 projected messages, blocked by 16 rows (include/tspgnn.h) -> fp32 row-major
        return z.view(pad // 16, w // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(pad, w).to(torch.float32)

    def acts_at(self, key, t):
        """(hidden activations of step t [layers-1, rows, d], element stride between layers)."""
        a = self.acts[key]
        if a.dtype == torch.float32 or self.native:
            return a[:, t], a.stride(0)
        w = a[:, t].to(torch.float32)
        return w, w.stride(0)

    def h_steps(self, v, t0, t1):
        return self._f32(self.H[v][t0:t1]).reshape(-1, self.H[v].shape[2])

    def x_steps(self, v, t0, t1):
        return self._f32(self.X[v][t0:t1]).reshape(-1, self.X[v].shape[2])

    def acts_steps(self, key, layer, t0, t1):
        a = self.acts[key]
        return self._f32(a[layer, t0:t1]).reshape(-1, a.shape[3])


class DeviceAdjacency(object):
    """A sparse [R, C] matrix resident on the device in CSR, both orientations."""

    def __init__(self, shape, device, csr, csr_t, uv=None):
        self.shape = tuple(shape)
        self.device = device
        self.csr = csr      # (rowptr[R+1], col[nnz], val[nnz] or None)   rows of the matrix
        self.csr_t = csr_t  # same for the transpose
        self.uv = uv        # int32 [R,2] if the matrix is 0/1 with exactly two ones per row
        self._degrees = {}
        self.loop_plan = None   # (int32 device tensor, n_groups, grid, kind, n_slots, lds_words, n_active): _make_loop_plan

    def row_degrees(self, transpose=False):
        """Stored entries per row (of the transpose) as fp32, computed once per matrix."""
        if transpose not in self._degrees:
            rowptr = (self.csr_t if transpose else self.csr)[0]
            self._degrees[transpose] = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
        return self._degrees[transpose]

    @staticmethod
    def from_sparse_ev(ev, device):
        rowptr, eid = ev.csr_by_vertex()
        M, N = ev.shape
        uv = _dev_i32(ev.uv, device)
        csr = (torch.arange(0, 2 * M + 1, 2, dtype=torch.int32, device=device), uv.view(-1), None)
        csr_t = (_dev_i32(rowptr, device), _dev_i32(eid, device), None)
        adj = DeviceAdjacency((M, N), device, csr, csr_t, uv=uv)
        adj.loop_plan = _make_loop_plan(ev, device)
        return adj

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

    def matmul(self, y, transpose=False, out=None):
        """mat (x) y  or  mat^T (x) y  (tf.matmul(..., adjoint_a=transpose), graphnn.py:156-160)."""
        R, C = self.shape
        rows_in = R if transpose else C
        rows_out = C if transpose else R
        if y.shape[0] != rows_in:
            raise ValueError("matrix/embedding size mismatch: %d vs %d" % (rows_in, y.shape[0]))
        d = y.shape[1]
        st = _lib.current_stream()
        if y.dtype == torch.bfloat16:   # bf16 rows, fp32 sums, one rounding at the store (pattern matrices only)
            if out is None:
                out = torch.empty((rows_out, d), dtype=torch.bfloat16, device=y.device)
            if not transpose and self.uv is not None:
                _lib.call("tspgnn_gather2_sum_bf16", _lib.ptr(self.uv), _lib.ptr(y), _lib.ptr(out), R, C, d, st)
                return out
            rowptr, col, val = self.csr_t if transpose else self.csr
            if val is not None:
                raise NotImplementedError("bf16 aggregation of a valued matrix")
            _lib.call("tspgnn_csr_rowsum_bf16", _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(y), _lib.ptr(out),
                      rows_out, rows_in, d, st)
            return out
        if d % 4 != 0:
            raise NotImplementedError("aggregation kernels need d %% 4 == 0 (got %d)" % d)
        if out is None:
            out = torch.empty((rows_out, d), dtype=torch.float32, device=y.device)
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

    def __call__(self, inputs, state, out=None):
        """Returns (new_h, LSTMStateTuple(new_c, new_h)) like the TF cell.  ``out`` = (h_out, c_out)
        lets the caller own the output buffers (training keeps every step's state)."""
        c, h = state.c, state.h
        rows = h.shape[0]
        if inputs.shape[0] != rows or inputs.shape[1] != self.dx:
            raise ValueError("cell input must be [%d,%d], got %s" % (rows, self.dx, tuple(inputs.shape)))
        x = inputs if inputs.is_contiguous() else inputs.contiguous()
        h_out, c_out = out if out is not None else (torch.empty_like(h), torch.empty_like(c))
        _lib.call("tspgnn_lnlstm_fwd_f32", _lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(self.kernel_packed()),
                  _lib.ptr(self.ln()), _lib.ptr(h_out), _lib.ptr(c_out), rows, self.d, _lib.current_stream())
        return h_out, LSTMStateTuple(c=c_out, h=h_out)

    # ------------------------------------------------------------------ backward
    def kernel_t_packed(self):
        """pack(K^T) ([4d, dx+d]) for the data gradient [dx | dh] = dz K^T."""
        if (self.dx + self.d) not in (64, 128, 256):
            raise NotImplementedError("LSTM backward needs dx+d in {64,128,256} (got %d)" % (self.dx + self.d))

        def build(out):
            K = self.kernel()
            if out is None:
                out = torch.empty_like(K)
            _lib.call("tspgnn_pack_weights_f32", _lib.ptr(K), _lib.ptr(out), 4 * self.d, self.dx + self.d, 1,
                      _lib.current_stream())
            return out
        return self.store.packed(("lstmT", self.base), build)

    def ln_grad(self):
        return self.store.grad_span(self.base + "/input/gamma", self.base + "/state/beta")

    # ------------------------------------------------------------------ folded adjacency product
    # (EV y) Kx = EV (y Kx): when the cell's only input is a gather over a two-ones-per-row matrix, the
    # x-half of the GEMM is applied on the (fewer) source rows and the cell kernel adds the two gathered
    # rows of Zx = y Kx to h Kh (tspgnn_lnlstm_gather_fwd_f32).
    def can_fold(self):
        return self.d == 64 and self.dx == 64

    def _packed_slice(self, key, rows_lo, rows_hi, transposed):
        def build(out):
            K = self.kernel()[rows_lo:rows_hi]
            if out is None:
                out = torch.empty_like(K)
            kr, nc = (4 * self.d, rows_hi - rows_lo) if transposed else (rows_hi - rows_lo, 4 * self.d)
            _lib.call("tspgnn_pack_weights_f32", _lib.ptr(K), _lib.ptr(out), kr, nc, 1 if transposed else 0,
                      _lib.current_stream())
            return out
        return self.store.packed((key, self.base), build)

    def _packed_split(self, arith, key, rows_lo, rows_hi, centered=False):
        """Split-operand packing of kernel rows [rows_lo, rows_hi) as a byte tensor: ``arith`` = "x3" (three bf16
        pieces, tspgnn_pack_weights_x3) or "h2" (two fp16 pieces of 2^s K, tspgnn_pack_weights_h2).  ``centered``: of
        the kernel with each gate's columns centred (_center_gates) -- a packing of its own."""
        def build(out):
            K = self.kernel()[rows_lo:rows_hi]
            if centered:
                K = _center_gates(K, self.d)
            if out is None:
                out = torch.empty(SPLIT_BYTES[arith] * K.numel(), dtype=torch.uint8, device=K.device)
            _pack_split(self.store, arith, K, out, rows_hi - rows_lo, 4 * self.d)
            return out
        return self.store.packed((key + "." + arith + (".c" if centered else ""), self.base), build)

    def _packed_x3(self, key, rows_lo, rows_hi):
        return self._packed_split("x3", key, rows_lo, rows_hi)

    def _packed_bf16(self, key, rows_lo, rows_hi):
        """Kernel rows [rows_lo, rows_hi) rounded to bf16 in MFMA fragment order: piece 0 of the bf16x3 packing."""
        n = (rows_hi - rows_lo) * 4 * self.d
        return self._packed_x3(key, rows_lo, rows_hi)[:2 * n]

    def task_bf16(self, x, h, c, h_out, c_out, rows, adj=None, state_in_blocked=False, state_out_blocked=False):
        """tspgnn_lstm_task_bf16 over ``rows`` rows: x, h, h_out bf16; c, c_out fp32; the states blocked by 16 rows when
        flagged (padded buffers).  With adj, the gather-init form: x is the blocked bf16 Zx of the source rows, K = Kh."""
        if adj is None:
            K = self._packed_bf16("lstm.x3", 0, self.dx + self.d)
            return _lib.LstmTaskB(_lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(K), _lib.ptr(self.ln()),
                                  _lib.ptr(h_out), _lib.ptr(c_out), rows, None, None, int(state_in_blocked),
                                  int(state_out_blocked))
        K = self._packed_bf16("lstm.kh.x3", self.dx, self.dx + self.d)
        return _lib.LstmTaskB(None, 0, _lib.ptr(h), _lib.ptr(c), _lib.ptr(K), _lib.ptr(self.ln()), _lib.ptr(h_out),
                              _lib.ptr(c_out), rows, _lib.ptr(adj.uv), _lib.ptr(x), int(state_in_blocked),
                              int(state_out_blocked))

    def x3_ok(self):
        """The split-operand cell kernels cover this shape (tspgnn_lnlstm_fwd_multi_x3 / _h2)."""
        return self.d in (32, 64) and self.dx % 32 == 0

    def kx_packed(self):
        return self._packed_slice("lstm.kx", 0, self.dx, False)

    def kh_packed(self):
        return self._packed_slice("lstm.kh", self.dx, self.dx + self.d, False)

    def kx_t_packed(self):
        return self._packed_slice("lstm.kxT", 0, self.dx, True)

    def kh_t_packed(self):
        return self._packed_slice("lstm.khT", self.dx, self.dx + self.d, True)

    def task(self, x, state, out, arith=None, centered=False):
        K = self._packed_split(arith, "lstm", 0, self.dx + self.d, centered) if arith else self.kernel_packed()
        return _lib.LstmTask(_lib.ptr(x), self.dx, _lib.ptr(state.h), _lib.ptr(state.c), _lib.ptr(K),
                             _lib.ptr(self.ln()), _lib.ptr(out[0]), _lib.ptr(out[1]), state.h.shape[0], None, None,
                             None, None, self._flag(arith), int(centered))

    def gather_task(self, adj, zx, state, out, arith=None, centered=False):
        """``centered``: Kh AND the Kx behind zx were centred per gate (the caller projects with
        _packed_split(..., "lstm.kx", ..., centered=True))."""
        K = self._packed_split(arith, "lstm.kh", self.dx, self.dx + self.d, centered) if arith else self.kh_packed()
        return _lib.LstmTask(None, 0, _lib.ptr(state.h), _lib.ptr(state.c), _lib.ptr(K),
                             _lib.ptr(self.ln()), _lib.ptr(out[0]), _lib.ptr(out[1]), state.h.shape[0],
                             _lib.ptr(adj.uv), _lib.ptr(zx), None, None, self._flag(arith), int(centered))

    def _flag(self, arith):
        """The range_flag of an f16x2 task (include/tspgnn.h): the store's guard word."""
        return self.store.h2_flag_ptr() if arith == "h2" else None

    def pushed_kernel(self, mlp):
        """(K' = [W Kx ; Kh] as fp32 [dx+d, 4d], b Kx as [1, 4d], pack(K'^T) for the data gradient) of the message MLP's
        last (linear) layer W, b pushed through a row-sum aggregation and through Kx:
            (sum_e (a_e W + b)) Kx = (sum_e a_e) (W Kx) + degree (b Kx)."""
        d, dx = self.d, self.dx
        last = mlp.layer_names[-1]

        def build(out):
            W, b = self.store.view(last + "/kernel"), self.store.view(last + "/bias")
            if out is None:
                out = (torch.empty((dx + d, 4 * d), dtype=torch.float32, device=W.device),
                       torch.empty((1, 4 * d), dtype=torch.float32, device=W.device),
                       torch.empty((dx + d, 4 * d), dtype=torch.float32, device=W.device))
            kfull, zb, kt = out
            st = _lib.current_stream()
            kfull[dx:].copy_(self.kernel()[dx:])
            _lib.call("tspgnn_linear_f32", _lib.ptr(W), dx, _lib.ptr(self.kx_packed()), None, 0, _lib.ptr(kfull[:dx]),
                      4 * d, 0, W.shape[0], st)
            _lib.call("tspgnn_linear_f32", _lib.ptr(b.view(1, -1)), dx, _lib.ptr(self.kx_packed()), None, 0, _lib.ptr(zb),
                      4 * d, 0, 1, st)
            _lib.call("tspgnn_pack_weights_f32", _lib.ptr(kfull), _lib.ptr(kt), 4 * d, dx + d, 1, st)
            return (kfull, zb, kt)
        return self.store.packed(("lstm.pushed.kernel", self.base, last), build)

    def pushed_bias_pack(self, mlp, arith=None, centered=False):
        """For a cell whose input is a row-sum aggregation of ``mlp``'s output: (pack(K'), b Kx) of pushed_kernel in the
        packing of ``arith``.  The cell then takes the row-sum of the LAST HIDDEN activation as its input and starts z
        at degree * (b Kx).  One Dense(d) layer less on every edge row per step."""
        d, dx = self.d, self.dx
        last = mlp.layer_names[-1]

        def build(out):
            kfull, zb, _ = self.pushed_kernel(mlp)
            if centered:   # (K' and b Kx centred per gate: tspgnn_lstm_task.z_centered)
                kfull = _center_gates(kfull, d)
            if out is None:
                out = torch.empty(SPLIT_BYTES[arith] * (dx + d) * 4 * d, dtype=torch.uint8, device=kfull.device) if arith \
                    else torch.empty((dx + d, 4 * d), dtype=torch.float32, device=kfull.device)
            st = _lib.current_stream()
            if arith:
                _pack_split(self.store, arith, kfull, out, dx + d, 4 * d)
            else:
                _lib.call("tspgnn_pack_weights_f32", _lib.ptr(kfull), _lib.ptr(out), dx + d, 4 * d, 0, st)
            return out
        packed = self.store.packed(("lstm.pushed." + (arith or "f32") + (".c" if centered else ""), self.base, last), build)
        if not centered:
            return packed, self.pushed_kernel(mlp)[1]
        zb_c = self.store.packed(("lstm.pushed.zb.c", self.base, last),
                                 lambda out: _center_gates(self.pushed_kernel(mlp)[1], d) if out is None
                                 else out.copy_(_center_gates(self.pushed_kernel(mlp)[1], d)))
        return packed, zb_c

    def pushed_backward_task(self, x, h, c, dh_out, dc_out, dz, dc_in, ws, kp, zb, deg, defer=False, data=None):
        """Backward task (tspgnn_lnlstm_bwd_multi_h2) of pushed_task: K = pushed_bias_pack's f16x2 K', z restarts at
        deg * zb.  ``data`` = (f16x2 packing of K'^T, dx_out, dh_in): the data gradient [dx_out | dh_in] = dz K'^T formed in
        the same launch (tspgnn_lstm_bwd_task.KTg) instead of by pushed_backward_data."""
        ktg, dx_out, dh_in = data if data is not None else (None, None, None)
        return _lib.LstmBwdTask(_lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(kp), _lib.ptr(self.ln()),
                                _lib.ptr(dh_out), _lib.ptr(dc_out), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(self.ln_grad()),
                                _lib.ptr(ws), h.shape[0], None, None, None, _lib.ptr(dh_in), 1 if defer else 0, _lib.ptr(zb),
                                _lib.ptr(deg), _lib.ptr(ktg), _lib.ptr(dx_out))

    def pushed_kernel_t_h2(self, mlp):
        """f16x2 packing of K'^T ([4d, dx+d], K' of pushed_kernel): the weight operand of the data gradient formed inside
        tspgnn_lnlstm_bwd_multi_h2 (tspgnn_lstm_bwd_task.KTg)."""
        last = mlp.layer_names[-1]

        def build(out):
            KT = self.pushed_kernel(mlp)[0].t().contiguous()
            if out is None:
                out = torch.empty(SPLIT_BYTES["h2"] * KT.numel(), dtype=torch.uint8, device=KT.device)
            _pack_split(self.store, "h2", KT, out, 4 * self.d, self.dx + self.d)
            return out
        return self.store.packed(("lstm.pushed.kT.h2", self.base, last), build)

    def fuses_pushed_data_gradient(self):
        """The pushed cell's data gradient can ride in its backward launch (f16x2, d == dx == 64)."""
        return self.d == 64 and self.dx == 64 and os.environ.get("TSPGNN_FUSE_DATA_GRADIENTS", "1") != "0"

    def pushed_backward_data(self, mlp, dz, dx_out, dh_in):
        """[d(aggregate) | dh] = dz K'^T."""
        _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(self.pushed_kernel(mlp)[2]), _lib.ptr(dx_out),
                  self.dx, _lib.ptr(dh_in), self.d, 0, dz.shape[0], _lib.current_stream())

    def pushed_backward_weights(self, agg_all, h_all, dz_all, rows, deg_all, g_wkx, g_zb):
        """Over rows = steps * rows_per_step: g_wkx[dx,4d] += agg^T dz (gradient w.r.t. the product W Kx), dKh += h^T dz,
        g_zb[4d] += sum_r deg[r] dz[r] (gradient w.r.t. b Kx); pushed_backward_finish turns the first and the last
        into the gradients of W, b and Kx."""
        gK = self.store.grad_view(self.base + "/kernel")
        st = _lib.current_stream()
        ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows, max(self.dx, self.d), 4 * self.d, device=dz_all.device)
        _lib.call("tspgnn_wgrad_f32", _lib.ptr(agg_all), _lib.ptr(dz_all), rows, self.dx, 4 * self.d, _lib.ptr(g_wkx), None,
                  _lib.ptr(ws), st)
        _lib.call("tspgnn_wgrad_f32", _lib.ptr(h_all), _lib.ptr(dz_all), rows, self.d, 4 * self.d, _lib.ptr(gK[self.dx:]), None,
                  _lib.ptr(ws), st)
        ws = _lib.workspace("tspgnn_wcolsum_workspace_floats", rows, 4 * self.d, device=dz_all.device)
        _lib.call("tspgnn_wcolsum_f32", _lib.ptr(dz_all), _lib.ptr(deg_all), rows, 4 * self.d, 1.0, _lib.ptr(g_zb), None,
                  _lib.ptr(ws), st)

    def pushed_backward_finish(self, mlp, g_wkx, g_zb):
        """With P = W Kx and q = b Kx:  dW += dP Kx^T,  db += dq Kx^T,  dKx += W^T dP + b^T dq  (once per training step,
        on [dx, 4d]-sized operands)."""
        d, dx, st = self.d, self.dx, _lib.current_stream()
        last = mlp.layer_names[-1]
        W, b = self.store.view(last + "/kernel"), self.store.view(last + "/bias")
        f32 = dict(dtype=torch.float32, device=W.device)
        dW, db = torch.empty((W.shape[0], dx), **f32), torch.empty((1, dx), **f32)
        _lib.call("tspgnn_linear_f32", _lib.ptr(g_wkx), 4 * d, _lib.ptr(self.kx_t_packed()), _lib.ptr(dW), dx, None, 0, 0,
                  g_wkx.shape[0], st)
        _lib.call("tspgnn_linear_f32", _lib.ptr(g_zb), 4 * d, _lib.ptr(self.kx_t_packed()), _lib.ptr(db), dx, None, 0, 0, 1, st)
        self.store.grad_view(last + "/kernel").add_(dW)
        self.store.grad_view(last + "/bias").add_(db.view(-1))
        packed = torch.empty_like(g_wkx)
        _lib.call("tspgnn_pack_weights_f32", _lib.ptr(g_wkx), _lib.ptr(packed), g_wkx.shape[0], 4 * d, 0, st)
        wt = W.t().contiguous()
        dkx = torch.empty((dx, 4 * d), **f32)
        _lib.call("tspgnn_linear_f32", _lib.ptr(wt), wt.shape[1], _lib.ptr(packed), None, 0, _lib.ptr(dkx), 4 * d, 0, dx, st)
        dkx.addcmul_(b.view(-1, 1), g_zb.view(1, -1))
        self.store.grad_view(self.base + "/kernel")[:dx].add_(dkx)

    def pushed_task(self, x, state, out, kp, zb, deg, arith=None, centered=False):
        """Cell task whose kernel operand is pushed_bias_pack's K' (either packing) and z starts at deg * zb."""
        return _lib.LstmTask(_lib.ptr(x), self.dx, _lib.ptr(state.h), _lib.ptr(state.c), _lib.ptr(kp), _lib.ptr(self.ln()),
                             _lib.ptr(out[0]), _lib.ptr(out[1]), state.h.shape[0], None, None, _lib.ptr(zb), _lib.ptr(deg),
                             self._flag(arith), int(centered))

    def premultiply(self, y, out=None, scale=None):
        """Zx = y Kx  ([n_src, 4d]); ``scale``: times 2^s for an f16x2 cell, whose z carries that factor."""
        if out is None:
            out = torch.empty((y.shape[0], 4 * self.d), dtype=torch.float32, device=y.device)
        if scale is None:
            _lib.call("tspgnn_linear_f32", _lib.ptr(y), self.dx, _lib.ptr(self.kx_packed()), None, 0, _lib.ptr(out),
                      4 * self.d, 0, y.shape[0], _lib.current_stream())
            return out
        # the f16x2 cells' projected-message format: times 2^s, blocked by 16 source rows (include/tspgnn.h)
        n, d4 = y.shape[0], 4 * self.d
        flat = torch.zeros((_pad16(n), d4), dtype=torch.float32, device=y.device)
        _lib.call("tspgnn_linear_f32", _lib.ptr(y), self.dx, _lib.ptr(self.kx_packed()), None, 0, _lib.ptr(flat), d4, 0, n,
                  _lib.current_stream())
        out.view(-1, d4 // 16, 4, 16, 4).copy_(flat.mul_(scale).view(-1, 16, d4 // 16, 4, 4).permute(0, 2, 3, 1, 4))
        return out

    def gather_call(self, adj, zx, state, out=None):
        c, h = state.c, state.h
        rows = h.shape[0]
        h_out, c_out = out if out is not None else (torch.empty_like(h), torch.empty_like(c))
        _lib.call("tspgnn_lnlstm_gather_fwd_f32", _lib.ptr(adj.uv), _lib.ptr(zx), _lib.ptr(h), _lib.ptr(c),
                  _lib.ptr(self.kh_packed()), _lib.ptr(self.ln()), _lib.ptr(h_out), _lib.ptr(c_out), rows, zx.shape[0],
                  self.d, _lib.current_stream())
        return h_out, LSTMStateTuple(c=c_out, h=h_out)

    def _packed_h2_t(self, key, rows_lo, rows_hi):
        """f16x2 packing of the TRANSPOSE of kernel rows [rows_lo, rows_hi) ([4d, rows]): the weight operand of the data
        gradient dz K^T in tspgnn_lnlstm_bwd_multi_h2."""
        def build(out):
            KT = self.kernel()[rows_lo:rows_hi].t().contiguous()
            if out is None:
                out = torch.empty(SPLIT_BYTES["h2"] * KT.numel(), dtype=torch.uint8, device=KT.device)
            _pack_split(self.store, "h2", KT, out, 4 * self.d, rows_hi - rows_lo)
            return out
        return self.store.packed((key + ".h2", self.base), build)

    def backward_task(self, x, h, c, dh_out, dc_out, dz, dc_in, ws, defer=False, arith=None):
        K = self._packed_split("h2", "lstm", 0, self.dx + self.d) if arith == "h2" else self.kernel_packed()
        return _lib.LstmBwdTask(_lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(K),
                                _lib.ptr(self.ln()), _lib.ptr(dh_out), _lib.ptr(dc_out), _lib.ptr(dz), _lib.ptr(dc_in),
                                _lib.ptr(self.ln_grad()), _lib.ptr(ws), h.shape[0], None, None, None, None,
                                1 if defer else 0)

    def gather_backward_task(self, adj, zx, h, c, dh_out, dc_out, dz, dc_in, ws, dh_in=None, defer=False, arith=None):
        """``dh_in`` given (d == 64): dh_in = dz Kh^T is formed in the same launch, from dz in registers.
        ``defer``: LayerNorm-gradient partials accumulate in ``ws`` (see backward_finish).
        ``arith`` = "h2": operands for tspgnn_lnlstm_bwd_multi_h2 (``zx`` as the f16x2 forward wrote it)."""
        fuse = dh_in is not None and self.d == 64
        if arith == "h2":
            K = self._packed_split("h2", "lstm.kh", self.dx, self.dx + self.d)
            KT = self._packed_h2_t("lstm.khT", self.dx, self.dx + self.d) if fuse else None
        else:
            K, KT = self.kh_packed(), (self.kh_t_packed() if fuse else None)
        return _lib.LstmBwdTask(None, 0, _lib.ptr(h), _lib.ptr(c), _lib.ptr(K), _lib.ptr(self.ln()),
                                _lib.ptr(dh_out), _lib.ptr(dc_out), _lib.ptr(dz), _lib.ptr(dc_in),
                                _lib.ptr(self.ln_grad()), _lib.ptr(ws), h.shape[0], _lib.ptr(adj.uv), _lib.ptr(zx),
                                _lib.ptr(KT), _lib.ptr(dh_in) if fuse else None, 1 if defer else 0)

    # ---- bf16-storage tape (tspgnn_lnlstm_bwd_multi_bf16 and friends, include/tspgnn.h)
    def _packed_bf16_t(self, key, rows_lo, rows_hi):
        """bf16 packing of the TRANSPOSE of kernel rows [rows_lo, rows_hi) ([4d, rows]): W of tspgnn_linear_bf16w_f32 for
        the data gradient dz K^T."""
        def build(out):
            KT = self.kernel()[rows_lo:rows_hi].t().contiguous()
            if out is None:
                out = torch.empty(SPLIT_BYTES["x3"] * KT.numel(), dtype=torch.uint8, device=KT.device)
            _pack_split(self.store, "x3", KT, out, 4 * self.d, rows_hi - rows_lo)
            return out
        return self.store.packed((key + ".x3", self.base), build)[:2 * 4 * self.d * (rows_hi - rows_lo)]

    def backward_task_bf16(self, x, h, c, dh_out, dc_out, dz, dc_in, ws, adj=None, zx=None):
        """tspgnn_lnlstm_bwd_multi_bf16 task: x / h (and zx: the blocked projected messages, gather-init mode with adj)
        are the tape's bf16 arrays."""
        if adj is not None:
            K = self._packed_bf16("lstm.kh.x3", self.dx, self.dx + self.d)
            return _lib.LstmBwdTask(None, 0, _lib.ptr(h), _lib.ptr(c), _lib.ptr(K), _lib.ptr(self.ln()), _lib.ptr(dh_out),
                                    _lib.ptr(dc_out), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(self.ln_grad()), _lib.ptr(ws),
                                    h.shape[0], _lib.ptr(adj.uv), _lib.ptr(zx), None, None, 1)
        K = self._packed_bf16("lstm.x3", 0, self.dx + self.d)
        return _lib.LstmBwdTask(_lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(K), _lib.ptr(self.ln()),
                                _lib.ptr(dh_out), _lib.ptr(dc_out), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(self.ln_grad()),
                                _lib.ptr(ws), h.shape[0], None, None, None, None, 1)

    def backward_data_bf16(self, dz, dx_out, dh_in):
        """[dx | dh] = dz K^T on the bf16 matrix cores (the weights are bf16-exact in this mode)."""
        W = self._packed_bf16_t("lstm.T", 0, self.dx + self.d)
        _lib.call("tspgnn_linear_bf16w_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(W), _lib.ptr(dx_out), self.dx,
                  _lib.ptr(dh_in), self.d, 0, dz.shape[0], _lib.current_stream())

    def gather_backward_data_bf16(self, adj, dz, dh_in, dzx, dy):
        """dh = dz Kh^T, dZx = EV^T dz, dy = dZx Kx^T (gather_backward_data with bf16-exact weights)."""
        st = _lib.current_stream()
        _lib.call("tspgnn_linear_bf16w_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(self._packed_bf16_t("lstm.khT", self.dx,
                  self.dx + self.d)), None, 0, _lib.ptr(dh_in), self.d, 0, dz.shape[0], st)
        adj.matmul(dz, transpose=True, out=dzx)
        _lib.call("tspgnn_linear_bf16w_f32", _lib.ptr(dzx), 4 * self.d, _lib.ptr(self._packed_bf16_t("lstm.kxT", 0, self.dx)),
                  None, 0, _lib.ptr(dy), self.dx, 0, dzx.shape[0], st)

    def backward_finish(self, ws):
        """Fold the LayerNorm-gradient partials that the deferred backward launches of all time steps left in ws."""
        _lib.call("tspgnn_lnlstm_bwd_finish_f32", _lib.ptr(ws), _lib.ptr(self.ln_grad()), self.d, _lib.current_stream())

    def backward_data(self, dz, dx_out, dh_in):
        """[dx | dh] = dz K^T."""
        _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(self.kernel_t_packed()), _lib.ptr(dx_out),
                  self.dx, _lib.ptr(dh_in), self.d, 0, dz.shape[0], _lib.current_stream())

    def kx_t_packed_h2(self):
        """f16x2 packing of Kx^T ([4d, dx]): dy = dZx Kx^T inside the source MLP's backward launch
        (tspgnn_mlp_bwd_task.pre_X)."""
        return self._packed_h2_t("lstm.kxT", 0, self.dx)

    def gather_backward_data(self, adj, dz, dh_in, dzx, dy):
        """dh = dz Kh^T (unless the cell launch already formed it: dh_in None), dZx = EV^T dz, dy = dZx Kx^T (``dy`` None:
        left to the source MLP's backward launch)."""
        st = _lib.current_stream()
        if dh_in is not None:
            _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(self.kh_t_packed()), None, 0, _lib.ptr(dh_in),
                      self.d, 0, dz.shape[0], st)
        adj.matmul(dz, transpose=True, out=dzx)
        if dy is None:
            return
        _lib.call("tspgnn_linear_f32", _lib.ptr(dzx), 4 * self.d, _lib.ptr(self.kx_t_packed()), None, 0, _lib.ptr(dy),
                  self.dx, 0, dzx.shape[0], st)

    def gather_backward(self, adj, zx, h, c, dh_out, dc_out, dz, dc_in, dh_in, dzx, dy, ws):
        """Backward of gather_call + premultiply: dz, dc_in, dh_in = dz Kh^T, dzx = EV^T dz, dy = dzx Kx^T."""
        rows, st = h.shape[0], _lib.current_stream()
        _lib.call("tspgnn_lnlstm_gather_bwd_f32", _lib.ptr(adj.uv), _lib.ptr(zx), _lib.ptr(h), _lib.ptr(c),
                  _lib.ptr(self.kh_packed()), _lib.ptr(self.ln()), _lib.ptr(dh_out), _lib.ptr(dc_out), _lib.ptr(dz),
                  _lib.ptr(dc_in), _lib.ptr(self.ln_grad()), _lib.ptr(ws), rows, self.d, st)
        _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(self.kh_t_packed()), None, 0, _lib.ptr(dh_in),
                  self.d, 0, rows, st)
        adj.matmul(dz, transpose=True, out=dzx)
        _lib.call("tspgnn_linear_f32", _lib.ptr(dzx), 4 * self.d, _lib.ptr(self.kx_t_packed()), None, 0, _lib.ptr(dy),
                  self.dx, 0, dzx.shape[0], st)

    def backward_weights_folded(self, y_all, dzx_all, rows_src, h_all, dz_all, rows):
        """dKx += y^T dZx over T*n_src rows (instead of T*M), dKh += h^T dz over T*M rows."""
        gK = self.store.grad_view(self.base + "/kernel")
        ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows_src, self.dx, 4 * self.d, device=dz_all.device)
        wgrad(y_all, dzx_all, rows_src, self.dx, 4 * self.d, gK[:self.dx], None, ws)
        ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows, self.d, 4 * self.d, device=dz_all.device)
        wgrad(h_all, dz_all, rows, self.d, 4 * self.d, gK[self.dx:], None, ws)

    def backward(self, x, h, c, dh_out, dc_out, dz, dc_in, dx_out, dh_in, ws):
        """One step: (dh_out, dc_out) -> dz (kept for the weight gradient), dc_in, dx_out, dh_in; the
        LayerNorm parameter gradients are accumulated into the store's gradient buffer."""
        rows, st = h.shape[0], _lib.current_stream()
        _lib.call("tspgnn_lnlstm_bwd_f32", _lib.ptr(x), self.dx, _lib.ptr(h), _lib.ptr(c), _lib.ptr(self.kernel_packed()),
                  _lib.ptr(self.ln()), _lib.ptr(dh_out), _lib.ptr(dc_out), _lib.ptr(dz), _lib.ptr(dc_in),
                  _lib.ptr(self.ln_grad()), _lib.ptr(ws), rows, self.d, st)
        _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * self.d, _lib.ptr(self.kernel_t_packed()), _lib.ptr(dx_out),
                  self.dx, _lib.ptr(dh_in), self.d, 0, rows, st)

    def backward_weights(self, x_all, h_all, dz_all, rows):
        """dK += [x|h]^T dz over all time steps at once (rows = T * rows_per_step)."""
        gK = self.store.grad_view(self.base + "/kernel")
        if self.dx:
            ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows, self.dx, 4 * self.d, device=dz_all.device)
            wgrad(x_all, dz_all, rows, self.dx, 4 * self.d, gK[:self.dx], None, ws)
        ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows, self.d, 4 * self.d, device=dz_all.device)
        wgrad(h_all, dz_all, rows, self.d, 4 * self.d, gK[self.dx:], None, ws)


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
        if float_dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError("GraphNN: float_dtype must be torch.float32 or torch.bfloat16 (bf16 storage of the "
                                      "embeddings with fp32 accumulation)")
        self.float_dtype = float_dtype
        self.store = store if store is not None else V.get_default_store()
        self.fold_adjacency = True   # (EV y) Kx = EV (y Kx) fast path; False = op-for-op reference order
        # f16x2 fused forward: LayerNorm's mean subtraction folded into the cell kernels (TSPGNN_CENTER_GATES=0: A/B)
        self.center_gates = os.environ.get("TSPGNN_CENTER_GATES", "1") != "0"
        # f16x2 inference forward: the whole T-step loop as ONE launch of resident workgroups (tspgnn_mp_loop_h2) where the
        # wiring and the batch allow it (_loop_launch); False / TSPGNN_LOOP=0 = one row-sum + one cell launch per step
        self.persistent_loop = True
        # training (f16x2): a message MLP's last linear layer pushed through the row-sum into the receiving cell, as in the
        # inference plan (one Dense layer less per edge row in the forward, the backward and the weight gradients)
        self.push_training = os.environ.get("TSPGNN_PUSH_TRAINING", "1") != "0"
        self.wgrad_chunk_bytes = 24 * 2 ** 30   # backward: budget for the pre-activation gradients kept per weight-gradient chunk
        # training forward: message MLPs of step t+1 inside the cell launch of step t (TSPGNN_FUSE_TRAINING=1).  Off by
        # default: the tape makes the training forward HBM-write-bound, and there the two plain launches at full occupancy
        # measured 0.2 ms per step FASTER than the fused one (C2: 11.75 vs 11.95 ms)
        self.fuse_training_messages = os.environ.get("TSPGNN_FUSE_TRAINING", "0") == "1"
        # training (f16x2 and bf16 storage, widths 64 / 128): the message MLPs' data gradient on the fp16 matrix cores
        # (tspgnn_mlp_bwd_multi_h2) instead of the fp32 matrix instruction (TSPGNN_MLP_BWD_H2=0: A/B)
        self.mlp_backward_h2 = os.environ.get("TSPGNN_MLP_BWD_H2", "1") != "0"
        # training (f16x2, pushed message MLPs of width 64), OPT-IN (TSPGNN_RECOMPUTE=1): the backward recomputes the MLP's
        # hidden activations and forms the MLP's weight gradients in the same launch (tspgnn_mlp_bwd_rc_h2), so that the
        # forward tapes only the messages and runs the message MLP inside the cell launch as the inference plan does.
        # Parity-green and deterministic; off by default because it does not pay at C2 (round 5, one box: taped 10.98-11.19 ms
        # per training step, this form 11.6-11.7 ms).  The variant that handed the recomputed activations to tspgnn_wgrad
        # instead (11.23-11.33 ms) was removed in round 6 -- DESIGN_HISTORY
        self.recompute_messages = os.environ.get("TSPGNN_RECOMPUTE", "0") == "1"
        # GEMM arithmetic of the inference forward, all fp32-class in accuracy: "f16x2" = fp16 matrix cores on
        # two-piece splits of the fp32 operands (csrc/dense_h2.hip, the default); "bf16x3" = bf16 matrix cores on
        # exact three-piece splits (csrc/dense_x3.hip); "f32" = fp32 MFMA.  TSPGNN_GEMM in the environment selects
        # one; shapes the split kernels do not cover fall back to "f32".
        self.gemm = os.environ.get("TSPGNN_GEMM", "f16x2")
        if self.gemm not in GEMM_ARITH:
            raise ValueError("TSPGNN_GEMM must be one of %s, got %r" % (sorted(GEMM_ARITH), self.gemm))
        self._h2_off_at = None       # store.assignments at which the weights were found outside the f16x2 range
        self._h2_force_off = False
        self._mlp_h2_native_ok = False   # bf16-storage backward: the last eager pass found the MLP weights inside the f16x2 range
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

    def _pushable(self, v, mats, folded):
        """True if v's cell input is a pattern (all-ones) adjacency product of a message MLP whose last
        layer is linear: that layer can be pushed through the row-sum into the cell (pushed_bias_pack)."""
        if not self.fold_adjacency or folded[v] is not None or len(self.loop[v]) != 1:
            return False
        u = self.loop[v][0]
        if "var" not in u or "fun" in u or "mat" not in u or "msg" not in u:
            return False
        mlp, cell = self._msg_MLPs[u["msg"]], self._RNN_cells[v]
        adj = mats[u["mat"]]
        csr = adj.csr_t if u.get("transpose?", False) else adj.csr
        if csr[2] is not None or mlp.relu[-1] or mlp.n_square < 2 or len(mlp._chunks()) != 1:
            return False
        return cell.d == 64 and cell.dx == 64 and mlp.sizes[-1] == cell.dx

    def _folded(self, v, mats):
        """The loop entry of v if its cell input is a single gather over a two-ones-per-row matrix
        (then the adjacency product is folded through the cell's GEMM), else None."""
        if not self.fold_adjacency or len(self.loop[v]) != 1:
            return None
        u = self.loop[v][0]
        if "var" not in u or "fun" in u or "mat" not in u or u.get("transpose?", False):
            return None
        if mats[u["mat"]].uv is None or not self._RNN_cells[v].can_fold():
            return None
        return u

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
        T = int(time_steps)
        h0 = {v: to_storage(init, self.float_dtype) for v, init in initial_embeddings.items()}   # storage type
        c0 = {v: LSTM_initial_states[v].to(torch.float32).contiguous() if v in LSTM_initial_states else None for v in h0}
        folded = {v: self._folded(v, mats) for v in self.var}
        if T > 0 and self.float_dtype == torch.float32:
            # (the fused plan reads the caller's embeddings in place and takes "no initial cell state" as such: no copies,
            # no zero fill)
            plan = self._plan_fused({v: LSTMStateTuple(c=c0[v], h=h0[v]) for v in h0}, mats, folded)
            if plan is not None and not self.check_h2_weights():   # (the plan's packings raised the guard: bf16x3 instead)
                plan = self._plan_fused({v: LSTMStateTuple(c=c0[v], h=h0[v]) for v in h0}, mats, folded)
            if plan is not None:
                return _States(plan(T), self._plan_keep)
        if self.float_dtype == torch.bfloat16:   # (no initial cell state = a null pointer to the bf16 cell kernel: no zero fill)
            return self._run_bf16({v: LSTMStateTuple(c=c0[v], h=h0[v]) for v in h0}, mats, dense_mats, T)
        states = {v: LSTMStateTuple(c=torch.zeros(h0[v].shape, dtype=torch.float32, device=h0[v].device) if c0[v] is None
                                    else c0[v], h=h0[v]) for v in h0}     # the cell state stays fp32
        if T > 0:
            plan = self._plan(states, mats, folded)
            if plan is not None and not self.check_h2_weights():
                plan = self._plan(states, mats, folded)
            if plan is not None:
                for t in range(T):
                    plan[t & 1]()
                return _States(plan[2][T & 1], self._plan_keep)
        for _ in range(T):
            states = self._step(states, mats, dense_mats, folded)
        return states

    def _run_bf16(self, states, mats, dense_mats, T):
        """bf16-storage forward (float_dtype=torch.bfloat16; BASELINE config 5): h, messages, aggregates and the
        projected messages Zx are bf16 in HBM, GEMMs are bf16 MFMA with fp32 accumulation, the cell state, LayerNorm
        and gate arithmetic fp32.  Three launches per step over ping-pong state buffers: message MLPs (vertex task
        with its Kx projection), adjacency products, cells (edge cell in gather-init mode)."""
        if dense_mats:
            raise NotImplementedError("bf16 storage: dense matrices appended as cell inputs")
        bf = dict(dtype=torch.bfloat16, device=self.store.theta.device)
        for v in self.var:
            if self.var[v] not in (32, 64, 128) or self._RNN_cells[v].dx % 32 != 0:
                raise NotImplementedError("bf16 storage needs widths 32/64/128 and cell inputs in multiples of 32")
            for u in self.loop[v]:
                if "var" not in u or "fun" in u:
                    raise NotImplementedError("bf16 storage supports loop entries made of var / msg / mat only")
                if "msg" in u:
                    m = self._msg_MLPs[u["msg"]]
                    if m._plan[0] != "square" or m._plan[3] or not (1 <= m.n_square <= 4) or m.input_size != m.sizes[-1]:
                        raise NotImplementedError("bf16 storage needs square message MLPs of at most 4 layers")
        if T == 0:
            return {v: LSTMStateTuple(c=st.c if st.c is not None else torch.zeros(st.h.shape, dtype=torch.float32,
                                                                                    device=st.h.device), h=st.h)
                    for v, st in states.items()}
        # Between the steps the states live in two ping-pong buffers BLOCKED by 16 rows (include/tspgnn.h: the lstm
        # task's state_*_blocked, the mlp task's x_blocked) -- only the loop's own launches read them, and a tile is then
        # loaded and stored in contiguous runs instead of as pieces of 16 rows.  The first step reads the caller's
        # row-major states in place, the last one writes row-major results.  An h that a loop entry uses WITHOUT a message
        # MLP (it goes to an aggregation or straight into a cell input) stays row-major.
        f32 = dict(dtype=torch.float32, device=self.store.theta.device)
        blocked = {v: all("msg" in u for w in self.var for u in self.loop[w] if u["var"] == v) for v in self.var}

        def buffers(dtype):
            return [{v: torch.empty((_pad16(st.h.shape[0]) if blocked[v] else st.h.shape[0], st.h.shape[1]),
                                    dtype=dtype, device=st.h.device) for v, st in states.items()} for _ in (0, 1)]
        hbuf, cbuf = buffers(torch.bfloat16), buffers(torch.float32)
        last = {v: LSTMStateTuple(c=torch.empty(st.h.shape, **f32), h=torch.empty_like(st.h)) for v, st in states.items()}

        def folds(v):   # single gather over a two-ones-per-row matrix behind a message MLP: Zx = msg(y) Kx on source rows
            if not self.fold_adjacency or len(self.loop[v]) != 1:
                return None
            u = self.loop[v][0]
            if "mat" not in u or "msg" not in u or u.get("transpose?", False) or mats[u["mat"]].uv is None:
                return None
            return u if self._RNN_cells[v].dx == self._msg_MLPs[u["msg"]].sizes[-1] == self.var[v] else None
        folded = {v: folds(v) for v in self.var}
        keep = [hbuf, cbuf, last, states]
        built = {}

        def step_launches(p, first, final):
            """(message-MLP launches, adjacency products, cell launches) of a step reading parity-p buffers."""
            key = (p, first, final)
            if key in built:
                return built[key]
            h_in = {v: states[v].h if first else hbuf[p][v] for v in self.var}
            c_in = {v: states[v].c if first else cbuf[p][v] for v in self.var}
            h_out = {v: last[v].h if final else hbuf[1 - p][v] for v in self.var}
            c_out = {v: last[v].c if final else cbuf[1 - p][v] for v in self.var}
            blk_in = {v: blocked[v] and not first for v in self.var}
            blk_out = {v: blocked[v] and not final for v in self.var}
            mlp_tasks, mid, msg_out, zxs, cell_tasks = {}, [], {}, {}, {}
            for v in self.var:
                for i, u in enumerate(self.loop[v]):
                    src = u["var"]
                    y = h_in[src]
                    if "msg" in u:
                        mlp = self._msg_MLPs[u["msg"]]
                        d = mlp.sizes[-1]
                        rows = states[src].h.shape[0]
                        out = torch.empty((rows, d), **bf)
                        pw = po = None
                        if folded[v] is not None:
                            cv = self._RNN_cells[v]
                            zxs[v] = torch.empty((_pad16(rows), 4 * self.var[v]), **bf)
                            pw, po = cv._packed_bf16("lstm.kx", 0, cv.dx), zxs[v]
                        n = mlp.n_square
                        # a plain message (no projection rides behind it): the last layer's columns interleaved in the
                        # packing, 16-byte stores of Y (tspgnn_mlp_task_bf16.y_interleaved)
                        il = pw is None and d % 32 == 0 and os.environ.get("TSPGNN_BF16_INTERLEAVE", "1") != "0"
                        mlp_tasks.setdefault(d, []).append(_lib.MlpTaskB(
                            _lib.ptr(y), _lib.ptr(mlp.wb_packed_bf16(0, n - 1, d, interleave_last=il)), _lib.ptr(out), rows, n,
                            mlp.relu_mask(0, n), _lib.ptr(pw), _lib.ptr(po), None, 0, int(blk_in[src]), int(il)))
                        y = out
                    msg_out[(v, i)] = y
            for v, d in self.var.items():
                cell = self._RNN_cells[v]
                rows = states[v].h.shape[0]
                if folded[v] is not None:
                    adj, x = mats[folded[v]["mat"]], zxs[v]
                else:
                    inputs = []
                    for i, u in enumerate(self.loop[v]):
                        y = msg_out[(v, i)]
                        if "mat" in u:
                            adj_, tr = mats[u["mat"]], u.get("transpose?", False)
                            o = torch.empty((adj_.shape[1] if tr else adj_.shape[0], y.shape[1]), **bf)
                            mid.append((adj_.matmul, (y, tr, o)))
                            y = o
                        inputs.append(y)
                    if len(inputs) == 1:
                        x = inputs[0]
                    else:
                        x = torch.empty((rows, cell.dx), **bf)
                        mid.append((lambda ins, o: torch.cat(ins, dim=1, out=o), (inputs, x)))
                    if x.shape[0] != rows or x.shape[1] != cell.dx:
                        raise ValueError("cell input must be [%d,%d], got %s" % (rows, cell.dx, tuple(x.shape)))
                    adj = None
                    keep.append(x)
                cell_tasks.setdefault(d, []).append(cell.task_bf16(
                    x, h_in[v], c_in[v], h_out[v], c_out[v], rows, adj=adj, state_in_blocked=blk_in[v],
                    state_out_blocked=blk_out[v]))
            keep.extend([msg_out, zxs])
            # tasks with a projection go to their own launch: without them the MLP kernel needs a third of the
            # registers (the projection's 4d accumulators) and the big edge task runs at twice the occupancy
            mlp_calls = []
            for d, ts in mlp_tasks.items():
                for group in ([t for t in ts if not t.proj_w], [t for t in ts if t.proj_w]):
                    mlp_calls += [(_lib.task_array(group[k:k + 4]), d) for k in range(0, len(group), 4)]
            cell_calls = [(_lib.task_array(ts[k:k + 4]), d) for d, ts in cell_tasks.items() for k in range(0, len(ts), 4)]
            built[key] = (mlp_calls, mid, cell_calls)
            return built[key]
        keep.append(built)
        self._plan_keep = keep
        for t in range(T):
            mlp_calls, mid, cell_calls = step_launches(t & 1, t == 0, t == T - 1)
            for arr, d in mlp_calls:
                _lib.call_multi("tspgnn_mlp_fwd_multi_bf16", arr, d)
            for fn, args in mid:
                fn(*args)
            for arr, d in cell_calls:
                _lib.call_multi("tspgnn_lnlstm_fwd_multi_bf16", arr, d)
        return _States(last, keep)

    def _split_arith(self, n_rows=None):
        """"h2" / "x3" when the selected split-operand kernels cover this network (widths 32/64, cell inputs in
        multiples of 32) and, when the row counts are given, this batch (they address rows with 32-bit element
        offsets: rows * 4d < 2^30); None = the fp32-MFMA kernels."""
        if n_rows is not None and any(int(n) * 4 * self.var[v] >= 2 ** 30 for v, n in n_rows.items()):
            return None
        if GEMM_ARITH[self.gemm] and all(c.x3_ok() for c in self._RNN_cells.values()) \
                and all(m.sizes[-1] in (32, 64) for m in self._msg_MLPs.values()):
            return self.active_arith()
        return None

    # ---- f16x2 range guard.  fp16 pieces cannot hold what fp32 -- the reference's type, graphnn.py:18 -- can: a weight
    # with 2^6 |w| >= 65504 or an activation >= 65504 overflows to inf.  Weights are checked where they are packed (a
    # device word raised by tspgnn_pack_weights_h2, read here when packings were refreshed); activations where the
    # kernels split them (the tasks' range_flag, read by Session.run next to the statistics it fetches anyway).  Either
    # way the network falls back to bf16x3, whose pieces have fp32's exponent range.
    def active_arith(self):
        """Suffix of the split-operand entry points in force: "h2" / "x3" / None, after the range guard's say."""
        arith = GEMM_ARITH[self.gemm]
        if arith == "h2" and (self._h2_force_off or self._h2_off_at == self.store.assignments):
            return "x3"
        return arith

    def training_packs_h2(self):
        """A training step of this network packs weights into fp16 pieces (the f16x2 forward and backward, or the bf16-storage
        mode's message-MLP backward): a replayed training graph has to watch the range guard's weight word."""
        if self.float_dtype == torch.bfloat16:
            return bool(self.mlp_backward_h2 and self._mlp_h2_native_ok)
        return self.active_arith() == "h2"

    def forced_off_h2(self):
        """Context manager: f16x2 disabled inside (Session's re-run of a batch whose activations overflowed)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._h2_force_off = self._h2_force_off, True
            try:
                yield self
            finally:
                self._h2_force_off = prev
        return cm()

    def check_h2_weights(self):
        """False if the weights just packed veto f16x2 (the caller rebuilds its launch plan: _split_arith now answers
        "x3"), True otherwise.  Reads the guard's weight word when packings were enqueued
        since the last look (one 4-byte device read; skipped while a HIP graph is being captured -- a captured
        sequence is checked by its replay closure); past HALF the fp16 range the network is latched to bf16x3 until
        the variables are assigned anew."""
        store = self.store
        if GEMM_ARITH[self.gemm] != "h2" or self._h2_force_off or self._h2_off_at == store.assignments:
            return True    # (not in use, or already vetoed for these variables: the caller's plan stands)
        if store.h2_packs_pending and store.theta.is_cuda and not torch.cuda.is_current_stream_capturing():
            guard = store.h2_guard()
            bits = int(guard[1].item())
            guard[1:2].zero_()
            store.h2_packs_pending = 0
            if bits >= store.H2_WEIGHT_LIMIT_BITS:
                self._h2_off_at = store.assignments
                return False
        return True

    def _single_consumers(self):
        """{source variable: (v, i)} when every variable's h feeds exactly one loop entry and that entry has a
        single-kernel square message MLP of the variable's own width (the wiring the fused cell + message launch
        covers), else None."""
        consumers = {u: [] for u in self.var}
        for v in self.var:
            for i, u in enumerate(self.loop[v]):
                if "var" not in u or "fun" in u or "msg" not in u:
                    return None
                mlp = self._msg_MLPs[u["msg"]]
                if mlp._plan[0] != "square" or len(mlp._chunks()) != 1 or mlp.n_square < 1 or mlp._plan[3] \
                        or mlp.sizes[-1] != self.var[u["var"]]:
                    return None
                consumers[u["var"]].append((v, i))
        if any(len(c) != 1 for c in consumers.values()):
            return None
        return {u: c[0] for u, c in consumers.items()}

    def _plan_fused(self, states, mats, folded):
        """bf16x3 plan with every message MLP fused behind the cell of its SOURCE variable: the launch that
        updates the states of step t also evaluates msg(h') -- the messages of step t+1 -- on the rows it
        still holds in registers (tspgnn_lnlstm_mlp_fwd_multi_x3), so a step is {adjacency products, one
        cell+message launch}.  The messages of step 0 come from one plain MLP launch, the last step runs
        the cells alone.  Applies when every variable's h feeds exactly one loop entry and that entry has a
        single-kernel message MLP; returns run(T) -> states, or None."""
        arith = self._split_arith({v: st.h.shape[0] for v, st in states.items()})
        if arith is None:
            return None
        consumers = {u: [] for u in self.var}
        for v in self.var:
            for i, u in enumerate(self.loop[v]):
                if "var" not in u or "fun" in u or "msg" not in u:
                    return None
                mlp = self._msg_MLPs[u["msg"]]
                if len(mlp._chunks()) != 1 or mlp.n_square < 1 or mlp._plan[3]:
                    return None
                consumers[u["var"]].append((v, i))
        if any(len(c) != 1 for c in consumers.values()):
            return None
        f32 = dict(dtype=torch.float32, device=self.store.theta.device)
        # f16x2: the states of a folded (edge-side) variable live BLOCKED by 16 rows between the steps (include/tspgnn.h,
        # tspgnn_cell_mlp_task.state_*_blocked): nothing but the variable's own cell task reads them -- the message MLP
        # takes h' from registers -- so the loop's ping-pong buffers are loaded and stored 1 KiB contiguous per
        # instruction; the first step reads the caller's row-major states, the last one writes row-major again.
        blocked = {v: arith == "h2" and folded[v] is not None for v in self.var}
        # f16x2: the cells' kernels (and with Kx the projected messages, and a pushed bias) are centred per gate, so that
        # the four gate LayerNorms of every row and step skip their mean pass (tspgnn_lstm_task.z_centered)
        cen = arith == "h2" and self.center_gates
        rows_of = {v: st.h.shape[0] for v, st in states.items()}

        def state_buffers():
            out = {}
            for v, st in states.items():
                rows = _pad16(rows_of[v]) if blocked[v] else rows_of[v]
                out[v] = LSTMStateTuple(c=torch.empty((rows, st.h.shape[1]), **f32), h=torch.empty((rows, st.h.shape[1]), **f32))
            return out
        # ping-pong buffers of the loop; the FIRST step reads the caller's states in place (row-major; c None = the zero
        # cell state, which the f16x2 kernels take as a null pointer: nothing is read), so nothing is copied or filled
        buf = [state_buffers(), state_buffers()]
        first_state = {v: LSTMStateTuple(c=st.c if st.c is not None or arith == "h2" else torch.zeros_like(st.h), h=st.h)
                       for v, st in states.items()}
        pushed = {v: self._pushable(v, mats, folded) for v in self.var}
        # message outputs / projected messages, double-buffered by step parity (a launch reads one set and
        # writes the other)
        mo = [{}, {}]
        zxs = [{}, {}]
        for v in self.var:
            for i, u in enumerate(self.loop[v]):
                rows, width = states[u["var"]].h.shape[0], self._msg_MLPs[u["msg"]].sizes[-1]
                for p in (0, 1):
                    if folded[v] is not None:
                        zxs[p][v] = torch.empty((_pad16(rows), 4 * self.var[v]), **f32)   # (fused plan: f16x2 / x3 only)
                    else:
                        mo[p][(v, i)] = torch.empty((rows, width), **f32)
        keep = [buf, mo, zxs, first_state]

        def message(v, i, p):
            """(wb, n_layers, relu_mask, out, proj_w, proj_out) of loop entry (v, i) writing parity-p buffers."""
            mlp = self._msg_MLPs[self.loop[v][i]["msg"]]
            d = mlp.sizes[-1]
            n = mlp.n_square - 1 if pushed[v] else mlp.n_square   # >= 1 (_pushable needs two square layers)
            pw = po = None
            if folded[v] is not None:
                cv = self._RNN_cells[v]
                pw, po = cv._packed_split(arith, "lstm.kx", 0, cv.dx, cen), zxs[p][v]
            out = mo[p].get((v, i))
            return (mlp.wb_packed_split(arith, 0, n - 1, d), n, mlp.relu_mask(0, n), out, pw, po)

        def cell_tasks(p, with_messages, first):
            """Launches of a step of parity p; ``first``: the step reads the caller's (row-major) states; a step
            without messages is the last one and writes row-major states."""
            src, dst = buf[p], buf[1 - p]
            mid, tasks = [], {}
            for v, d in self.var.items():
                cell = self._RNN_cells[v]
                n_v = rows_of[v]
                st = first_state[v] if first else LSTMStateTuple(c=src[v].c[:n_v], h=src[v].h[:n_v])
                out = (dst[v].h, dst[v].c)
                if folded[v] is not None:
                    t = cell.gather_task(mats[folded[v]["mat"]], zxs[p][v], st, out, arith=arith, centered=cen)
                else:
                    inputs = []
                    for i, u in enumerate(self.loop[v]):
                        y = mo[p][(v, i)]
                        if "mat" in u:
                            adj, tr = mats[u["mat"]], u.get("transpose?", False)
                            o = torch.empty((adj.shape[1] if tr else adj.shape[0], y.shape[1]), **f32)
                            mid.append((adj.matmul, (y, tr, o)))
                            y = o
                        inputs.append(y)
                    if len(inputs) == 1:
                        x = inputs[0]
                    else:
                        x = torch.empty((st.h.shape[0], cell.dx), **f32)
                        mid.append((lambda ins, o: torch.cat(ins, dim=1, out=o), (inputs, x)))
                    if x.shape[0] != st.h.shape[0] or x.shape[1] != cell.dx:
                        raise ValueError("cell input must be [%d,%d], got %s" % (st.h.shape[0], cell.dx, tuple(x.shape)))
                    keep.append(x)
                    if pushed[v]:
                        u0 = self.loop[v][0]
                        kp, zb = cell.pushed_bias_pack(self._msg_MLPs[u0["msg"]], arith=arith, centered=cen)
                        deg = mats[u0["mat"]].row_degrees(bool(u0.get("transpose?", False)))
                        t = cell.pushed_task(x, st, out, kp, zb, deg, arith=arith, centered=cen)
                    else:
                        t = cell.task(x, st, out, arith=arith, centered=cen)
                s_in = 1 if blocked[v] and not first else 0
                s_out = 1 if blocked[v] and with_messages else 0
                if with_messages:   # the message MLP that reads this variable's new h in the next step
                    (cv, ci), = consumers[v]
                    wb, n, mask, mout, pw, po = message(cv, ci, 1 - p)
                    ct = _lib.CellMlpTask(t, _lib.ptr(wb), n, mask, _lib.ptr(mout), _lib.ptr(pw), _lib.ptr(po), s_in, s_out)
                else:
                    ct = _lib.CellMlpTask(t, None, 0, 0, None, None, None, s_in, s_out)
                tasks.setdefault(d, []).append(ct)
            calls = [(_lib.task_array(ts[k:k + 4]), d) for d, ts in tasks.items() for k in range(0, len(ts), 4)]
            return mid, calls

        # messages of step 0 from the initial states
        pre = {}
        for v in self.var:
            for i, u in enumerate(self.loop[v]):
                wb, n, mask, mout, pw, po = message(v, i, 0)
                y = first_state[u["var"]].h
                if mout is None:
                    mout = torch.empty((y.shape[0], self._msg_MLPs[u["msg"]].sizes[-1]), **f32)
                    keep.append(mout)
                pre.setdefault(self._msg_MLPs[u["msg"]].sizes[-1], []).append(
                    _lib.MlpTask(_lib.ptr(y), _lib.ptr(wb), _lib.ptr(mout), None, 0, y.shape[0], n, mask,
                                 _lib.ptr(pw), _lib.ptr(po), self.store.h2_flag_ptr() if arith == "h2" else None))
        pre_calls = [(_lib.task_array(ts[k:k + 4]), d) for d, ts in pre.items() for k in range(0, len(ts), 4)]
        built = {}
        self._plan_keep = keep
        # the cells' packings are made here, not at the first step: the caller vets their range (check_h2_weights) between
        # the construction of the plan and its first launch
        for v in self.var:
            cell = self._RNN_cells[v]
            if folded[v] is not None:
                cell._packed_split(arith, "lstm.kh", cell.dx, cell.dx + cell.d, cen)
            elif pushed[v]:
                cell.pushed_bias_pack(self._msg_MLPs[self.loop[v][0]["msg"]], arith=arith, centered=cen)
            else:
                cell._packed_split(arith, "lstm", 0, cell.dx + cell.d, cen)

        loop_launch = self._loop_launch(states, mats, folded, pushed, consumers, message, first_state, arith, cen, keep) \
            if arith == "h2" else None

        def run(T):
            for arr, d in pre_calls:
                _lib.call_multi("tspgnn_mlp_fwd_multi_" + arith, arr, d)
            if loop_launch is not None:
                done = loop_launch(T)
                if done is not None:
                    return done
            for t in range(T):
                kind = (t & 1, t < T - 1, t == 0)
                if kind not in built:
                    built[kind] = cell_tasks(*kind)
                mid, calls = built[kind]
                for fn, args in mid:
                    fn(*args)
                for arr, d in calls:
                    _lib.call_multi("tspgnn_lnlstm_mlp_fwd_multi_" + arith, arr, d)
            final = buf[T & 1]
            return {v: LSTMStateTuple(c=final[v].c[:rows_of[v]], h=final[v].h[:rows_of[v]]) for v in self.var}
        return run

    def _loop_launch(self, states, mats, folded, pushed, consumers, message, first_state, arith, cen, keep):
        """run(T) -> states through ONE launch of tspgnn_mp_loop_h2 (csrc/mp_loop_h2.hip: resident workgroups, edge
        states in registers, per-group synchronisation), or None when the wiring or the batch is not the one that kernel
        is written for: two variables of width 64, the "edge" one folded over a two-ones-per-row matrix that carries a
        work plan (DeviceAdjacency.loop_plan), the "vertex" one fed by the row-sum over the same matrix's transpose.
        Results are bit-identical to the stepwise launches of _plan_fused (tests/test_gpu_loop.py)."""
        if not loop_enabled() or not self.persistent_loop or len(self.var) != 2:
            return None
        ve = [v for v in self.var if folded[v] is not None]
        if len(ve) != 1:
            return None
        ve = ve[0]
        vv = [v for v in self.var if v != ve][0]
        if self.var[ve] != 64 or self.var[vv] != 64:
            return None
        ue, uvx = folded[ve], self.loop[vv]
        if len(uvx) != 1:
            return None
        uvx = uvx[0]
        if ue.get("var") != vv or uvx.get("var") != ve or uvx.get("mat") != ue["mat"] or not uvx.get("transpose?", False) \
                or "fun" in uvx or "msg" not in uvx or "msg" not in ue:
            return None
        adj = mats[ue["mat"]]
        if adj.loop_plan is None or adj.csr_t[2] is not None:
            return None
        plan_t, n_groups, grid, kind, n_slots, lds_words, n_active = adj.loop_plan
        cell_e, cell_v = self._RNN_cells[ve], self._RNN_cells[vv]
        if cell_v.dx != 64 or cell_e.dx != 64:
            return None
        M, N = states[ve].h.shape[0], states[vv].h.shape[0]
        f32 = dict(dtype=torch.float32, device=self.store.theta.device)
        # the edge side's message MLP (consumed by the vertex cell) and the vertex side's (consumed, projected, by the edge cell)
        e_wb, e_n, e_mask, e_out0, _, _ = message(vv, 0, 0)
        _, _, _, e_out1, _, _ = message(vv, 0, 1)
        v_wb, v_n, v_mask, _, v_pw, zx0 = message(ve, 0, 0)
        _, _, _, _, _, zx1 = message(ve, 0, 1)
        if e_out0 is None or zx0 is None or e_n > 3 or v_n > 4 or v_n < 1:   # (e_n <= 3: resident next to Kh in LDS)
            return None
        if pushed[vv]:
            mlp_e = self._msg_MLPs[uvx["msg"]]
            v_K, zb = cell_v.pushed_bias_pack(mlp_e, arith=arith, centered=cen)
            deg = adj.row_degrees(True)
        else:
            v_K, zb, deg = cell_v._packed_split(arith, "lstm", 0, cell_v.dx + cell_v.d, cen), None, None
        e_K = cell_e._packed_split(arith, "lstm.kh", cell_e.dx, cell_e.dx + cell_e.d, cen)
        out_e = LSTMStateTuple(c=torch.empty((M, 64), **f32), h=torch.empty((M, 64), **f32))
        out_v = LSTMStateTuple(c=torch.empty((N, 64), **f32), h=torch.empty((N, 64), **f32))
        vagg = [torch.empty((N, 64), **f32), torch.empty((N, 64), **f32)]
        counters = torch.zeros(4 * 32 * n_groups + 32, dtype=torch.int32, device=f32["device"])
        guard = self.store.h2_guard()
        fs_e, fs_v = first_state[ve], first_state[vv]
        resident = kind == "resident"
        a = _lib.MpResidentArgs() if resident else _lib.MpLoopArgs()
        if resident:   # the edge states between the steps, by tile slot (private to the launch)
            slots = [torch.empty((n_slots * 16, 64), **f32), torch.empty((n_slots * 16, 64), **f32)]
            a.e_hs, a.e_cs, a.n_slots, a.lds_words = _lib.ptr(slots[0]), _lib.ptr(slots[1]), n_slots, lds_words
            a.n_active = n_active
            a.flags = 1 if os.environ.get("TSPGNN_RES_SAFE") == "1" else 0   # (tests: the placement-independent acquire)
            vh = [torch.empty((N, 64), **f32), torch.empty((N, 64), **f32)]
            a.vh[0], a.vh[1] = _lib.ptr(vh[0]), _lib.ptr(vh[1])
            keep.extend([slots, vh])
        a.e_h0, a.e_c0, a.e_h, a.e_c = _lib.ptr(fs_e.h), _lib.ptr(fs_e.c), _lib.ptr(out_e.h), _lib.ptr(out_e.c)
        a.uv, a.e_K, a.e_ln = _lib.ptr(adj.uv), _lib.ptr(e_K), _lib.ptr(cell_e.ln())
        a.e_mlp_wb, a.e_mlp_layers, a.e_relu_mask = _lib.ptr(e_wb), e_n, e_mask
        a.msg[0], a.msg[1] = _lib.ptr(e_out0), _lib.ptr(e_out1)
        a.v_h0, a.v_c0, a.v_h, a.v_c = _lib.ptr(fs_v.h), _lib.ptr(fs_v.c), _lib.ptr(out_v.h), _lib.ptr(out_v.c)
        a.rowptr, a.eid = _lib.ptr(adj.csr_t[0]), _lib.ptr(adj.csr_t[1])
        a.v_K, a.v_ln, a.v_zbias, a.v_zscale = _lib.ptr(v_K), _lib.ptr(cell_v.ln()), _lib.ptr(zb), _lib.ptr(deg)
        a.v_mlp_wb, a.v_mlp_layers, a.v_relu_mask, a.v_proj_w = _lib.ptr(v_wb), v_n, v_mask, _lib.ptr(v_pw)
        a.zx[0], a.zx[1] = _lib.ptr(zx0), _lib.ptr(zx1)
        a.vagg[0], a.vagg[1] = _lib.ptr(vagg[0]), _lib.ptr(vagg[1])
        a.plan, a.counters, a.n_groups, a.grid = _lib.ptr(plan_t), _lib.ptr(counters), n_groups, grid
        a.M, a.N, a.z_centered = M, N, int(cen)
        a.range_flag, a.status = guard.data_ptr(), guard.data_ptr() + 8
        trace = None
        if os.environ.get("TSPGNN_LOOP_TRACE"):   # development: per-wavefront phase times (tools/loop_trace.py)
            trace = self.loop_trace = torch.zeros((grid, resident_plan.WAVES if resident else loop_plan.WAVES, 32 if resident else 16),
                                                  dtype=torch.int64, device=f32["device"])
        a.trace = _lib.ptr(trace)
        keep.extend([out_e, out_v, vagg, counters, plan_t, v_K, zb, deg, e_K, a, trace])

        unsupported = [False]

        def launch(T):
            """-> the final states, or None when the device cannot keep the launch's workgroups resident (the entry point
            asks the runtime before it launches anything and answers TSPGNN_EUNSUPPORTED): the caller then runs the
            stepwise launches."""
            if unsupported[0]:
                return None
            a.T = int(T)
            counters.zero_()
            try:
                _lib.call("tspgnn_mp_resident_h2" if resident else "tspgnn_mp_loop_h2", ctypes.byref(a), 64,
                          _lib.current_stream())
            except _lib.TspgnnError as e:
                if e.status != -2:
                    raise
                unsupported[0] = True
                return None
            return {ve: out_e, vv: out_v}
        return launch

    def _plan(self, states, mats, folded):
        """Pre-builds the launches of an even and an odd step over two ping-pong state buffers, so that
        the T-step loop is a handful of ctypes calls per step (the host stays ahead of the GPU even
        without HIP-graph replay).  Returns (run_even, run_odd, (states_if_T_even, states_if_T_odd)), or
        None when the wiring needs the general path (Python 'fun' entries, dense matrices, MLPs that do
        not fit one kernel)."""
        f32 = dict(dtype=torch.float32, device=self.store.theta.device)
        for v in self.var:
            for u in self.loop[v]:
                if "var" not in u or "fun" in u:
                    return None
                if "msg" in u and len(self._msg_MLPs[u["msg"]]._chunks()) != 1:
                    return None
        # buf[p] holds the states a step of parity p READS; it writes buf[1-p]
        buf = [{v: LSTMStateTuple(c=st.c.clone(), h=st.h.clone()) for v, st in states.items()},
               {v: LSTMStateTuple(c=torch.empty_like(st.c), h=torch.empty_like(st.h)) for v, st in states.items()}]
        runs, keep = [], []
        arith = self._split_arith({v: st.h.shape[0] for v, st in states.items()})
        zx_scale = _lib.lib.tspgnn_h2_weight_scale() if arith == "h2" else None
        for p in (0, 1):
            src_states, dst_states = buf[p], buf[1 - p]
            mlp_tasks, lstm_tasks, mid, msg_out, zxs = {}, {}, [], {}, {}
            pushed = {v: self._pushable(v, mats, folded) for v in self.var}
            for v in self.var:
                for i, u in enumerate(self.loop[v]):
                    y = src_states[u["var"]].h
                    if "msg" in u:
                        mlp = self._msg_MLPs[u["msg"]]
                        out = torch.empty((y.shape[0], mlp.sizes[-1]), **f32)
                        if pushed[v]:   # last hidden activation only; the last layer is folded into the cell
                            mlp_tasks.setdefault(mlp.sizes[-1], []).append(mlp.prefix_task(y, out, mlp.n_square - 1, arith=arith))
                            msg_out[(v, i)] = out
                            continue
                        proj = None
                        if folded[v] is not None:   # Zx = msg(y) Kx rides in the MLP launch
                            zxs[v] = torch.empty((_pad16(y.shape[0]), 4 * self.var[v]), **f32)
                            cv = self._RNN_cells[v]
                            proj = (cv._packed_split(arith, "lstm.kx", 0, cv.dx) if arith else cv.kx_packed(), zxs[v])
                        mlp_tasks.setdefault(mlp.sizes[-1], []).append(mlp.task(y, out, proj=proj, arith=arith))
                        y = out
                    msg_out[(v, i)] = y
            for v, d in self.var.items():
                cell, st = self._RNN_cells[v], src_states[v]
                out = (dst_states[v].h, dst_states[v].c)
                if folded[v] is not None:
                    if v in zxs:
                        zx = zxs[v]
                    else:
                        zx = torch.empty((_pad16(msg_out[(v, 0)].shape[0]), 4 * d), **f32)
                        mid.append((cell.premultiply, (msg_out[(v, 0)], zx, zx_scale)))
                    lstm_tasks.setdefault(d, []).append(cell.gather_task(mats[folded[v]["mat"]], zx, st, out, arith=arith))
                    keep.append(zx)
                    continue
                inputs = []
                for i, u in enumerate(self.loop[v]):
                    y = msg_out[(v, i)]
                    if "mat" in u:
                        adj, tr = mats[u["mat"]], u.get("transpose?", False)
                        o = torch.empty((adj.shape[1] if tr else adj.shape[0], y.shape[1]), **f32)
                        mid.append((adj.matmul, (y, tr, o)))
                        y = o
                    inputs.append(y)
                if len(inputs) == 1:
                    x = inputs[0]
                else:
                    x = torch.empty((st.h.shape[0], cell.dx), **f32)
                    mid.append((lambda ins, o: torch.cat(ins, dim=1, out=o), (inputs, x)))
                if x.shape[0] != st.h.shape[0] or x.shape[1] != cell.dx:
                    raise ValueError("cell input must be [%d,%d], got %s" % (st.h.shape[0], cell.dx, tuple(x.shape)))
                if pushed[v]:
                    u0 = self.loop[v][0]
                    kp, zb = cell.pushed_bias_pack(self._msg_MLPs[u0["msg"]], arith=arith)
                    deg = mats[u0["mat"]].row_degrees(bool(u0.get("transpose?", False)))
                    lstm_tasks.setdefault(d, []).append(cell.pushed_task(x, st, out, kp, zb, deg, arith=arith))
                else:
                    lstm_tasks.setdefault(d, []).append(cell.task(x, st, out, arith=arith))
                keep.append(x)
            keep.append(msg_out)

            mlp_calls = [(_lib.task_array(ts[k:k + 4]), d) for d, ts in mlp_tasks.items() for k in range(0, len(ts), 4)]
            lstm_calls = [(_lib.task_array(ts[k:k + 4]), d) for d, ts in lstm_tasks.items() for k in range(0, len(ts), 4)]

            mlp_fn = "tspgnn_mlp_fwd_multi_" + (arith or "f32")
            lstm_fn = "tspgnn_lnlstm_fwd_multi_" + (arith or "f32")

            def run(mlp_calls=mlp_calls, mid=mid, lstm_calls=lstm_calls, mlp_fn=mlp_fn, lstm_fn=lstm_fn):
                for arr, d in mlp_calls:
                    _lib.call_multi(mlp_fn, arr, d)
                for fn, args in mid:
                    fn(*args)
                for arr, d in lstm_calls:
                    _lib.call_multi(lstm_fn, arr, d)
            runs.append(run)
        self._plan_keep = keep   # buffers referenced by raw pointers inside the task structures
        return runs[0], runs[1], (buf[0], buf[1])

    def _step(self, states, mats, dense_mats, folded):
        """One synchronous step (graphnn.py:142-173) in three phases, so that independent work of the
        same kind shares a launch: (A) every message MLP, (B) the adjacency products / vertex-side
        pre-multiplications, (C) every cell."""
        f32 = dict(dtype=torch.float32, device=self.store.theta.device)
        # ---- A: message MLPs of all loop entries
        msg_out, tasks, keep = {}, {}, []
        for v in self.var:
            for i, u in enumerate(self.loop[v]):
                if "var" not in u:
                    continue
                y = states[u["var"]].h
                if "fun" in u:
                    y = u["fun"](y)
                if "msg" in u:
                    mlp = self._msg_MLPs[u["msg"]]
                    out = torch.empty((y.shape[0], mlp.sizes[-1]), **f32)
                    yc = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.to(torch.float32).contiguous()
                    t = mlp.task(yc, out)
                    if t is None:
                        out = mlp(y)
                    else:
                        tasks.setdefault(mlp.sizes[-1], []).append(t)
                        keep.append(yc)
                    y = out
                msg_out[(v, i)] = y
        for d, ts in tasks.items():
            for k in range(0, len(ts), 4):
                _lib.call_multi("tspgnn_mlp_fwd_multi_f32", ts[k:k + 4], d)
        # ---- B: adjacency products (or, for a folded cell, Zx = y Kx on the source rows)
        cell_in = {}
        for v in self.var:
            if folded[v] is not None:
                cell_in[v] = self._RNN_cells[v].premultiply(msg_out[(v, 0)])
                continue
            inputs = []
            for i, u in enumerate(self.loop[v]):
                if "var" in u:
                    y = msg_out[(v, i)]
                    if "mat" in u:
                        y = mats[u["mat"]].matmul(y, transpose=u.get("transpose?", False))
                    inputs.append(y)
                else:
                    inputs.append(dense_mats[u["mat"]])
            cell_in[v] = inputs[0] if len(inputs) == 1 else torch.cat(inputs, dim=1)
        # ---- C: all cells
        new_states, tasks = {}, {}
        for v, d in self.var.items():
            cell, st = self._RNN_cells[v], states[v]
            out = (torch.empty_like(st.h), torch.empty_like(st.c))
            new_states[v] = LSTMStateTuple(c=out[1], h=out[0])
            if folded[v] is not None:
                t = cell.gather_task(mats[folded[v]["mat"]], cell_in[v], st, out)
            else:
                x = cell_in[v]
                if x.shape[0] != st.h.shape[0] or x.shape[1] != cell.dx:
                    raise ValueError("cell input must be [%d,%d], got %s" % (st.h.shape[0], cell.dx, tuple(x.shape)))
                x = x if x.is_contiguous() else x.contiguous()
                keep.append(x)
                t = cell.task(x, st, out)
            tasks.setdefault(d, []).append(t)
        for d, ts in tasks.items():
            for k in range(0, len(ts), 4):
                _lib.call_multi("tspgnn_lnlstm_fwd_multi_f32", ts[k:k + 4], d)
        return new_states

    # ---------------------------------------------------------------- training: forward with a tape
    def forward_train(self, adjacency_matrices, initial_embeddings, time_steps):
        """Same computation as __call__, keeping what the backward pass needs: every step's states,
        cell inputs and hidden MLP activations, each stored [T, rows, width] contiguous so that a
        variable's weight gradient is ONE reduction over all time steps (sized for 288 GB of HBM:
        ~10 GB at n=40, batch 128, T=32).  Returns (states, tape).

        float_dtype=torch.bfloat16 (bf16 storage, BASELINE config 5): the forward is the inference mode's (bf16 h,
        messages, aggregates, projected messages and hidden activations, bf16 MFMA with fp32 accumulation, fp32 c /
        LayerNorm / gates) and the tape holds those bf16 arrays -- half the bytes; see backward()."""
        T = int(time_steps)
        bf16 = self.float_dtype == torch.bfloat16
        if self.float_dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError("training runs with float_dtype torch.float32 or torch.bfloat16")
        self.check_run(adjacency_matrices, initial_embeddings, T, {})
        device = next(iter(initial_embeddings.values())).device
        f32 = dict(dtype=torch.float32, device=device)
        stored = dict(dtype=self.float_dtype, device=device)   # what the tape keeps of h, messages, activations
        mats, dense = {}, {}
        for v in self.var:
            for u in self.loop[v]:
                if ("fun" in u or "var" not in u) and bf16:
                    raise NotImplementedError("bf16-storage training supports loop entries made of var / msg / mat only")
                if "var" not in u:
                    # graphnn.py:163-165: the matrix itself joins the cell input -- a constant of the step (a placeholder:
                    # tf.gradients stops there); its columns only meet the cell kernel's rows
                    if u["mat"] not in dense:
                        a = adjacency_matrices[u["mat"]]
                        a = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
                        dense[u["mat"]] = a.to(device=device, dtype=torch.float32).contiguous()
                    continue
                if "mat" in u and u["mat"] not in mats:
                    mats[u["mat"]] = DeviceAdjacency.wrap(adjacency_matrices[u["mat"]], device)
        tape = Tape()
        tape.T, tape.mats, tape.dense = T, mats, dense
        tape.folded = {v: self._folded(v, mats) for v in self.var}
        n = {v: initial_embeddings[v].shape[0] for v in self.var}
        tape.H = {v: torch.empty((T + 1, n[v], d), **stored) for v, d in self.var.items()}
        tape.C = {v: torch.empty((T + 1, n[v], d), **f32) for v, d in self.var.items()}
        # cell inputs; for a folded cell: the message y per SOURCE row and Zx = y Kx instead of the aggregate
        tape.X, tape.ZX = {}, {}
        for v in self.var:
            u = tape.folded[v]
            rows_x = n[v] if u is None else n[u["var"]]
            tape.X[v] = torch.empty((T, rows_x, self._RNN_cells[v].dx), **stored)
            if u is not None:
                tape.ZX[v] = torch.empty((T, _pad16(rows_x), 4 * self.var[v]), **stored)
        tape.acts = {}
        for v in self.var:
            to_storage(initial_embeddings[v], self.float_dtype, out=tape.H[v][0])   # (bf16 storage: rounded here, as the
                                                                                    # inference mode does)
            tape.C[v][0].zero_()
            for i, u in enumerate(self.loop[v]):
                if "msg" in u and "var" in u:
                    tape.acts[(v, i)] = None     # allocated below, once the arithmetic (hence the tape's form) is known
        tape.fused = False
        tape.rc = {}

        def alloc_acts():
            for (v, i) in tape.acts:
                u = self.loop[v][i]
                mlp, src = self._msg_MLPs[u["msg"]], u["var"]
                layers = 1 if tape.rc.get((v, i)) else max(mlp.n_square - 1, 1)   # (recomputed: the messages only)
                tape.acts[(v, i)] = torch.empty((layers, T, n[src], self.var[src]), **stored)
        if bf16:
            tape.arith = "bf16"
            tape.pushed = {v: False for v in self.var}
            alloc_acts()
            self._forward_train_bf16(tape, n, T)
            return {v: LSTMStateTuple(c=tape.C[v][T], h=tape.H[v][T]) for v in self.var}, tape
        # forward GEMMs in the split-operand arithmetic selected by self.gemm (fp32-class accuracy).  With f16x2 the
        # cells' backward recomputes z in the same arithmetic (tspgnn_lnlstm_bwd_multi_h2; the tape's projected
        # messages ZX carry the factor 2^s both sides expect); with bf16x3 the backward is fp32 MFMA.
        arith = self._split_arith({v: initial_embeddings[v].shape[0] for v in self.var})
        if arith == "h2":
            # the step's f16x2 packings up front (they are cached for the tasks below), so that the guard can veto them
            # before any kernel multiplies with them
            # (exactly the packings the tasks below take: a folded cell multiplies with Kh and projects with Kx, a pushed
            # one with the PRODUCT [W Kx ; Kh] -- which can leave the range on its own --, the others with the whole kernel)
            for v, cell in self._RNN_cells.items():
                if tape.folded[v] is not None:
                    cell._packed_split("h2", "lstm.kh", cell.dx, cell.dx + cell.d)
                    if "msg" in tape.folded[v]:
                        cell._packed_split("h2", "lstm.kx", 0, cell.dx)
                elif self.push_training and not self.fuse_training_messages and self._pushable(v, mats, tape.folded):
                    cell.pushed_bias_pack(self._msg_MLPs[self.loop[v][0]["msg"]], arith="h2")
                else:
                    cell._packed_split("h2", "lstm", 0, cell.dx + cell.d)
            for mlp in self._msg_MLPs.values():
                if len(mlp._chunks()) == 1:
                    mlp.wb_packed_split("h2", 0, mlp.n_square - 1, mlp.sizes[-1])
            if not self.check_h2_weights():
                arith = self._split_arith({v: initial_embeddings[v].shape[0] for v in self.var})
        tape.arith = arith
        mlp_fn = "tspgnn_mlp_fwd_multi_" + (arith or "f32")
        lstm_fn = "tspgnn_lnlstm_fwd_multi_" + (arith or "f32")
        # pushed cells (f16x2): the tape's X[v] holds the row-sum of the message MLP's LAST HIDDEN activation and the cell
        # runs with K' = [W Kx ; Kh], z starting at degree * (b Kx) (LayerNormBasicLSTMCell.pushed_kernel)
        tape.pushed = {v: bool(arith == "h2" and self.push_training and not self.fuse_training_messages
                               and self._pushable(v, mats, tape.folded)) for v in self.var}
        # recomputed message MLPs (tspgnn_mlp_bwd_rc_h2): the pushed entries whose prefix the kernel covers
        for v in self.var:
            if tape.pushed[v] and self.recompute_messages:
                mlp = self._msg_MLPs[self.loop[v][0]["msg"]]
                tape.rc[(v, 0)] = mlp.recompute_ok(mlp.n_square - 1)
        alloc_acts()

        def message_dest(v, i, t):
            """(out, projection) of loop entry (v, i)'s message MLP at step t: outputs straight into the tape."""
            u = self.loop[v][i]
            mlp = self._msg_MLPs[u["msg"]]
            to_tape = tape.folded[v] is not None or (len(self.loop[v]) == 1 and "mat" not in u)
            out = tape.X[v][t] if to_tape else torch.empty((n[u["var"]], mlp.sizes[-1]), **f32)
            proj = None
            if tape.folded[v] is not None:
                cv = self._RNN_cells[v]
                proj = (cv._packed_split(arith, "lstm.kx", 0, cv.dx) if arith else cv.kx_packed(), tape.ZX[v][t])
            return out, proj

        def messages(t):
            """A: every message MLP of step t in one launch (per width); -> {(v, i): message rows}."""
            msg_out, mlp_tasks = {}, {}
            for v in self.var:
                for i, u in enumerate(self.loop[v]):
                    if "var" not in u:          # an appended matrix: joins the cell input in aggregate()
                        msg_out[(v, i)] = tape.dense[u["mat"]]
                        continue
                    y = tape.H[u["var"]][t]
                    if "fun" in u:              # graphnn.py:149-151 (the backward differentiates it again: _fun_vjp)
                        y = u["fun"](y)
                        if not torch.is_tensor(y) or y.shape != tape.H[u["var"]][t].shape:
                            raise ValueError("loop entry 'fun' must map [rows, d] to a tensor of the same shape")
                        y = y.to(torch.float32).contiguous()
                    if "msg" in u:
                        mlp = self._msg_MLPs[u["msg"]]
                        acts = tape.acts[(v, i)]
                        if tape.pushed[v]:      # all but the last layer; its output is the last saved activation
                            k = mlp.n_square - 1
                            if tape.rc.get((v, i)):
                                out = acts[0, t]
                                task = mlp.prefix_task(y, out, k, arith=arith)
                            else:
                                out = acts[k - 1, t]
                                task = mlp.prefix_task(y, out, k, arith=arith, acts=acts[:, t], acts_stride=acts.stride(0))
                            mlp_tasks.setdefault(mlp.sizes[-1], []).append(task)
                            msg_out[(v, i)] = out
                            continue
                        out, proj = message_dest(v, i, t)
                        task = mlp.task(y, out, acts[:, t], acts.stride(0), proj=proj, arith=arith)
                        if task is None:
                            if arith:
                                raise NotImplementedError("bf16x3 training forward needs single-kernel message MLPs")
                            mlp.forward_saving(y, out, acts[:, t], acts.stride(0))
                            if proj is not None:
                                self._RNN_cells[v].premultiply(out, out=tape.ZX[v][t])
                        else:
                            mlp_tasks.setdefault(mlp.sizes[-1], []).append(task)
                        y = out
                    msg_out[(v, i)] = y
            for d, ts in mlp_tasks.items():
                for k in range(0, len(ts), 4):
                    _lib.call_multi(mlp_fn, ts[k:k + 4], d)
            return msg_out

        def aggregate(t, msg_out):
            """B: adjacency products / vertex-side pre-multiplication of step t."""
            for v in self.var:
                if tape.folded[v] is not None:
                    u = tape.folded[v]
                    if "msg" not in u:   # with a message MLP, Zx was produced by the MLP launch itself
                        tape.X[v][t].copy_(msg_out[(v, 0)])
                        self._RNN_cells[v].premultiply(tape.X[v][t], out=tape.ZX[v][t],
                                                       scale=_lib.lib.tspgnn_h2_weight_scale() if arith == "h2" else None)
                    continue
                single = len(self.loop[v]) == 1
                inputs = []
                for i, u in enumerate(self.loop[v]):
                    y = msg_out[(v, i)]
                    if "var" not in u:
                        if single:
                            tape.X[v][t].copy_(y)
                    elif "mat" in u:
                        y = mats[u["mat"]].matmul(y, transpose=u.get("transpose?", False),
                                                  out=tape.X[v][t] if single else None)
                    elif single and ("msg" not in u or "fun" in u):
                        tape.X[v][t].copy_(y)
                    inputs.append(y)
                if not single:
                    torch.cat(inputs, dim=1, out=tape.X[v][t])

        def cell_task(v, t):
            cell = self._RNN_cells[v]
            st = LSTMStateTuple(c=tape.C[v][t], h=tape.H[v][t])
            out = (tape.H[v][t + 1], tape.C[v][t + 1])
            if tape.folded[v] is not None:
                return cell.gather_task(mats[tape.folded[v]["mat"]], tape.ZX[v][t], st, out, arith=arith)
            if tape.pushed[v]:
                u0 = self.loop[v][0]
                kp, zb = cell.pushed_bias_pack(self._msg_MLPs[u0["msg"]], arith=arith)
                return cell.pushed_task(tape.X[v][t], st, out, kp, zb,
                                        mats[u0["mat"]].row_degrees(bool(u0.get("transpose?", False))), arith=arith)
            return cell.task(tape.X[v][t], st, out, arith=arith)

        # f16x2, opt-in (fuse_training_messages): the message MLPs of step t+1 ride in the cell launch of step t, on the
        # rows of h' it still holds in registers (tspgnn_lnlstm_mlp_fwd_multi_h2 as in the inference plan, here writing
        # the tape: states, hidden activations, messages and projected messages of every step)
        consumers = self._single_consumers() if arith == "h2" and (self.fuse_training_messages or any(tape.rc.values())) \
            else None
        tape.fused = consumers is not None
        if consumers is not None:
            msg_out = messages(0) if T > 0 else {}
            for t in range(T):
                aggregate(t, msg_out)
                tasks, nxt = {}, {}
                for v, d in self.var.items():
                    task = cell_task(v, t)
                    if t < T - 1:
                        cv, ci = consumers[v]
                        mlp = self._msg_MLPs[self.loop[cv][ci]["msg"]]
                        acts = tape.acts[(cv, ci)]
                        if tape.pushed[cv]:     # all but the last layer; the message is the last hidden activation
                            k = mlp.n_square - 1
                            rc = tape.rc.get((cv, ci))
                            out, (pw, po) = acts[0 if rc else k - 1, t + 1], (None, None)
                            saved = None if (rc or k == 1) else acts[:, t + 1]
                        else:
                            out, proj = message_dest(cv, ci, t + 1)
                            pw, po = proj if proj is not None else (None, None)
                            k = mlp.n_square
                            saved = acts[:, t + 1] if k > 1 else None
                        ct = _lib.CellMlpTask(task, _lib.ptr(mlp.wb_packed_split(arith, 0, k - 1, d)), k, mlp.relu_mask(0, k),
                                              _lib.ptr(out), _lib.ptr(pw), _lib.ptr(po), 0, 0,
                                              _lib.ptr(saved), acts.stride(0))
                        nxt[(cv, ci)] = out
                    else:
                        ct = _lib.CellMlpTask(task, None, 0, 0, None, None, None, 0, 0, None, 0)
                    tasks.setdefault(d, []).append(ct)
                for d, ts in tasks.items():
                    for k in range(0, len(ts), 4):
                        _lib.call_multi("tspgnn_lnlstm_mlp_fwd_multi_h2", ts[k:k + 4], d)
                msg_out = nxt
        else:
            for t in range(T):
                aggregate(t, messages(t))
                # ---- C: every cell of the step in one launch (per width)
                lstm_tasks = {}
                for v, d in self.var.items():
                    lstm_tasks.setdefault(d, []).append(cell_task(v, t))
                for d, ts in lstm_tasks.items():
                    for k in range(0, len(ts), 4):
                        _lib.call_multi(lstm_fn, ts[k:k + 4], d)
        states = {v: LSTMStateTuple(c=tape.C[v][T], h=tape.H[v][T]) for v in self.var}
        return states, tape

    def _forward_train_bf16(self, tape, n, T):
        """The T steps of the bf16-storage forward (three launches per step as in _run_bf16: message MLPs -- the vertex
        task with its Kx projection --, adjacency products, cells), every output written into the tape."""
        mats = tape.mats
        bf = dict(dtype=torch.bfloat16, device=self.store.theta.device)
        for v in self.var:
            if self.var[v] not in (32, 64, 128) or self._RNN_cells[v].dx % 32 != 0:
                raise NotImplementedError("bf16 storage needs widths 32/64/128 and cell inputs in multiples of 32")
            if tape.folded[v] is not None and ("msg" not in tape.folded[v]
                                               or self._RNN_cells[v].dx != self._msg_MLPs[tape.folded[v]["msg"]].sizes[-1]
                                               or self._RNN_cells[v].dx != self.var[v]):
                raise NotImplementedError("bf16 storage: a folded cell input needs a message MLP of the cell's width")
            for u in self.loop[v]:
                if "msg" in u:
                    m = self._msg_MLPs[u["msg"]]
                    if m._plan[0] != "square" or m._plan[3] or not (1 <= m.n_square <= 4) or m.input_size != m.sizes[-1]:
                        raise NotImplementedError("bf16 storage needs square message MLPs of at most 4 layers")
        for t in range(T):
            msg_out, plain, with_proj = {}, {}, {}
            for v in self.var:
                for i, u in enumerate(self.loop[v]):
                    y = tape.H[u["var"]][t]
                    if "msg" in u:
                        mlp = self._msg_MLPs[u["msg"]]
                        d = mlp.sizes[-1]
                        acts = tape.acts[(v, i)]
                        to_tape = tape.folded[v] is not None or (len(self.loop[v]) == 1 and "mat" not in u)
                        out = tape.X[v][t] if to_tape else torch.empty((y.shape[0], d), **bf)
                        pw = po = None
                        if tape.folded[v] is not None:
                            cv = self._RNN_cells[v]
                            pw, po = cv._packed_bf16("lstm.kx", 0, cv.dx), tape.ZX[v][t]
                        k = mlp.n_square
                        task = _lib.MlpTaskB(_lib.ptr(y), _lib.ptr(mlp.wb_packed_bf16(0, k - 1, d)), _lib.ptr(out), y.shape[0], k,
                                             mlp.relu_mask(0, k), _lib.ptr(pw), _lib.ptr(po),
                                             _lib.ptr(acts[:, t]) if k > 1 else None, acts.stride(0))
                        (with_proj if pw is not None else plain).setdefault(d, []).append(task)
                        y = out
                    msg_out[(v, i)] = y
            for group in (plain, with_proj):   # (projections in their own launch: see _run_bf16)
                for d, ts in group.items():
                    for k in range(0, len(ts), 4):
                        _lib.call_multi("tspgnn_mlp_fwd_multi_bf16", ts[k:k + 4], d)
            cells = {}
            for v, d in self.var.items():
                cell = self._RNN_cells[v]
                if tape.folded[v] is not None:
                    task = cell.task_bf16(tape.ZX[v][t], tape.H[v][t], tape.C[v][t], tape.H[v][t + 1], tape.C[v][t + 1], n[v],
                                          adj=mats[tape.folded[v]["mat"]])
                else:
                    single = len(self.loop[v]) == 1
                    inputs = []
                    for i, u in enumerate(self.loop[v]):
                        y = msg_out[(v, i)]
                        if "mat" in u:
                            y = mats[u["mat"]].matmul(y, transpose=u.get("transpose?", False),
                                                      out=tape.X[v][t] if single else None)
                        elif single and "msg" not in u:
                            tape.X[v][t].copy_(y)
                        inputs.append(y)
                    if not single:
                        torch.cat(inputs, dim=1, out=tape.X[v][t])
                    task = cell.task_bf16(tape.X[v][t], tape.H[v][t], tape.C[v][t], tape.H[v][t + 1], tape.C[v][t + 1], n[v])
                cells.setdefault(d, []).append(task)
            for d, ts in cells.items():
                for k in range(0, len(ts), 4):
                    _lib.call_multi("tspgnn_lnlstm_fwd_multi_bf16", ts[k:k + 4], d)

    def backward(self, tape, dstates):
        """Back-propagation through time of forward_train.  dstates: {var: (dh, dc)} gradients w.r.t. the
        final states (None = zero).  Parameter gradients are ADDED to the store's flat gradient buffer;
        returns {var: (d h0, d c0)} (gradients w.r.t. the initial embeddings / cell states).

        bf16-storage tape: gradients are fp32 throughout (the fp32-MFMA backward kernels on the widened tape), taken
        of the function the forward evaluated -- stored values as they were rounded, GEMM weights rounded to bf16,
        every rounding passed straight through -- and land on the fp32 master variables."""
        if getattr(tape, "arith", None) == "bf16":
            names = [c.base + "/kernel" for c in self._RNN_cells.values()]
            names += [ln + "/kernel" for m in self._msg_MLPs.values() for ln in m.layer_names]
            with self.store.rounded_to_bf16(names):
                return self._backward(tape, dstates)
        return self._backward(tape, dstates)

    @staticmethod
    def _fun_vjp(fun, h, g_out):
        """g_out (gradient w.r.t. fun(h)) pulled back to h.  A loop entry's 'fun' is the caller's own code (graphnn.py:149-151
        applies it to the states inside the TF graph, and tf.gradients differentiates it there): either it brings its
        vector-Jacobian product along -- ``fun.vjp(h, g_out) -> g_in`` -- or it is made of differentiable torch operations
        and is differentiated the same way TF would, by the framework's autograd on this one call."""
        vjp = getattr(fun, "vjp", None)
        if vjp is not None:
            return vjp(h, g_out).to(torch.float32)
        with torch.enable_grad():
            x = h.detach().to(torch.float32).requires_grad_(True)
            y = fun(x)
            if not y.requires_grad:       # a function that ignores its argument (or detaches): zero gradient
                return torch.zeros_like(x)
            (g_in,) = torch.autograd.grad(y, x, grad_outputs=g_out.to(y.dtype))
        return g_in

    def _backward(self, tape, dstates):
        T, mats = tape.T, tape.mats
        device = self.store.theta.device
        f32 = dict(dtype=torch.float32, device=device)
        n = {v: tape.H[v].shape[1] for v in self.var}
        folded = tape.folded
        bwd_arith = "h2" if getattr(tape, "arith", None) == "h2" else None   # the cells' backward follows the forward
        # bf16-storage tape: the bf16-reading backward kernels take the tape's arrays as they are (widths 64 / 128; the
        # narrow widths widen slices of the tape for the fp32 kernels instead)
        native = getattr(tape, "arith", None) == "bf16" and all(d in (64, 128) for d in self.var.values()) \
            and all(c.dx % 64 == 0 for c in self._RNN_cells.values()) \
            and os.environ.get("TSPGNN_BF16_BACKWARD", "native") == "native"
        tape.native = native
        # Weight gradients are one reduction per variable over a CHUNK of time steps: all T when the gradients w.r.t.
        # the pre-activations of the chunk (4d + the MLP layers' d floats per row and step) fit the budget -- the C2
        # case, ~6 GB -- else the largest chunk that does (a C5 shard: 84 GB for all 64 steps)
        pushed = getattr(tape, "pushed", None) or {v: False for v in self.var}
        rc = getattr(tape, "rc", None) or {}      # entries whose backward recomputes the hidden activations and forms the
                                                  # weight gradients in the same launch (no chunk buffers)
        per_step = sum(n[v] * 4 * d * 4 for v, d in self.var.items())
        per_step += sum(self._msg_MLPs[self.loop[v][i]["msg"]].n_square * n[self.loop[v][i]["var"]]
                        * self.var[self.loop[v][i]["var"]] * 4 * (0 if rc.get((v, i)) else 1)
                        for (v, i) in tape.acts)
        per_step += sum(tape.X[v].shape[1] * 4 * self.var[v] * 4 for v in self.var if folded[v] is not None)   # DZX
        per_step += sum(n[v] * 4 for v in self.var if (getattr(tape, "pushed", None) or {}).get(v))          # degrees
        if getattr(tape, "arith", None) == "bf16" and not native:
            # widened fp32 copies of the chunk's tape slices (h, cell inputs, hidden activations) for the fp32 reductions
            per_step += sum(n[v] * d * 4 + tape.X[v].shape[1] * tape.X[v].shape[2] * 4 for v, d in self.var.items())
            per_step += sum(a.shape[0] * a.shape[2] * a.shape[3] * 4 for a in tape.acts.values())
        budget = self.wgrad_chunk_bytes
        if device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            # never more than half of what is free next to the tape (smaller devices, other HBM sizes); memory the caching
            # allocator holds but has not handed out counts as free
            avail = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
            budget = min(budget, avail // 2)
        CH = max(1, min(T, int(budget // max(per_step, 1)))) if T > 0 else 1
        DZ = {v: torch.empty((CH, n[v], 4 * d), **f32) for v, d in self.var.items()}
        # bf16-storage tape: the f16x2 data gradient of the message MLPs packs 2^s W^T into fp16 pieces, which the bf16 forward
        # never vetted -- pack now and look at the guard's weight word (one 4-byte read per backward pass): beyond half the fp16
        # range the pass uses tspgnn_mlp_bwd_multi_f32.  A pass being CAPTURED cannot look: it follows the eager pass before it
        # (Session.capture_train_step warms up eagerly) and its replays watch the word at the f16x2 guard's lag
        mlp_h2_native = bool(native and self.mlp_backward_h2 and self._mlp_h2_native_ok)
        if native and self.mlp_backward_h2 and device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            store = self.store
            for (v, i), acts in tape.acts.items():
                mlp = self._msg_MLPs[self.loop[v][i]["msg"]]
                if mlp.backward_h2_ok(acts):
                    for l0, nl in mlp._chunks():
                        mlp.wt_packed_h2(l0, l0 + nl - 1, mlp.sizes[-1])
            guard = store.h2_guard()
            bits = int(guard[1].item())
            guard[1:2].zero_()
            store.h2_packs_pending = 0
            mlp_h2_native = self._mlp_h2_native_ok = bits < store.H2_WEIGHT_LIMIT_BITS
        DPRE, RCP = {}, {}
        for (v, i), acts in tape.acts.items():
            u = self.loop[v][i]
            mlp = self._msg_MLPs[u["msg"]]
            if rc.get((v, i)):
                RCP[(v, i)] = mlp.backward_rc_partial(mlp.n_square - 1)   # workgroup partials of {dW, db}, all T steps
                continue
            DPRE[(v, i)] = torch.empty((mlp.n_square - (1 if pushed[v] else 0), CH, n[u["var"]], self.var[u["var"]]), **f32)
        # LayerNorm-gradient partials of all T steps accumulate here (zeroed); one fold per cell after the loop
        ws = {v: _lib.workspace("tspgnn_lnlstm_bwd_workspace_floats", d, device=device).zero_() for v, d in self.var.items()}
        DZX = {v: torch.empty((CH, tape.X[v].shape[1], 4 * self.var[v]), **f32) for v in self.var if folded[v] is not None}

        # pushed cells: gradients w.r.t. the products W Kx [dx, 4d] and b Kx [4d], split into dW, db, dKx after the loop
        push = {}
        for v, d in self.var.items():
            if pushed[v]:
                u0 = self.loop[v][0]
                deg = mats[u0["mat"]].row_degrees(bool(u0.get("transpose?", False)))
                push[v] = dict(mlp=self._msg_MLPs[u0["msg"]], deg=deg, deg_steps=deg.repeat(CH),
                               g_wkx=torch.zeros((self._RNN_cells[v].dx, 4 * d), **f32), g_zb=torch.zeros((1, 4 * d), **f32))

        def weight_gradients(t0, t1):
            """Steps [t0, t1): their dz / dpre sit in slots 0 .. t1-t0-1 of the chunk buffers."""
            steps = t1 - t0
            for v, d in self.var.items():
                cell = self._RNN_cells[v]
                if pushed[v]:
                    cell.pushed_backward_weights(tape.x_steps(v, t0, t1), tape.h_steps(v, t0, t1), DZ[v][:steps].view(-1, 4 * d),
                                                 steps * n[v], push[v]["deg_steps"], push[v]["g_wkx"], push[v]["g_zb"])
                elif folded[v] is not None:
                    rows_src = steps * tape.X[v].shape[1]
                    cell.backward_weights_folded(tape.x_steps(v, t0, t1), DZX[v][:steps].view(-1, 4 * d), rows_src,
                                                 tape.h_steps(v, t0, t1), DZ[v][:steps].view(-1, 4 * d), steps * n[v])
                else:
                    cell.backward_weights(tape.x_steps(v, t0, t1), tape.h_steps(v, t0, t1), DZ[v][:steps].view(-1, 4 * d),
                                          steps * n[v])
            for (v, i), dpre in DPRE.items():
                u = self.loop[v][i]
                mlp = self._msg_MLPs[u["msg"]]
                src, dsrc = u["var"], self.var[u["var"]]
                layers = dpre.shape[0]      # (a pushed entry's last layer has its gradient formed on the receiving side)
                first = tape.h_steps(src, t0, t1)
                if "fun" in u:      # the MLP's input rows are fun(h), step by step as the forward formed them
                    with torch.no_grad():
                        first = torch.cat([u["fun"](tape.h(src, t)).to(torch.float32) for t in range(t0, t1)], dim=0).contiguous()
                inputs = [first] + [tape.acts_steps((v, i), l, t0, t1) for l in range(layers - 1)]
                mlp.backward_weights(inputs, [dpre[l, :steps].reshape(-1, dsrc) for l in range(layers)], steps * n[src],
                                     n_layers=layers)

        dH = {v: (dstates.get(v, (None, None))[0]) for v in self.var}
        dC = {v: (dstates.get(v, (None, None))[1]) for v in self.var}
        # The vertex side's two data-gradient GEMMs ride in the step's two backward launches (TSPGNN_FUSE_DATA_GRADIENTS=0: as
        # launches of their own): a pushed cell's [d(aggregate) | dh] = dz K'^T as a second phase of its backward task,
        # a folded cell's dy = dZx Kx^T as the head of its source MLP's chain
        fuse_data = bwd_arith == "h2" and not native and os.environ.get("TSPGNN_FUSE_DATA_GRADIENTS", "1") != "0"
        fused_data = {v: bool(fuse_data and pushed[v] and self._RNN_cells[v].fuses_pushed_data_gradient()) for v in self.var}
        projected = {}
        for v in self.var:
            u0 = self.loop[v][0]
            projected[v] = bool(
                fuse_data and folded[v] is not None and len(self.loop[v]) == 1 and "msg" in u0 and "fun" not in u0
                and not rc.get((v, 0)) and not pushed[v] and self.mlp_backward_h2 and T > 0
                and self._msg_MLPs[u0["msg"]].backward_task_takes_projection(tape.acts_at((v, 0), 0)[0],
                                                                              4 * self._RNN_cells[v].d))
        for t in range(T - 1, -1, -1):
            k = t % CH      # slot of step t in the chunk buffers (chunks start at multiples of CH)
            ndH = {v: torch.empty((n[v], d), **f32) for v, d in self.var.items()}
            ndC = {v: torch.empty((n[v], d), **f32) for v, d in self.var.items()}
            dX = {v: torch.empty((tape.X[v].shape[1], self._RNN_cells[v].dx), **f32) for v in self.var}
            keep = []       # widened tape slices stay alive until the step's launches are enqueued
            # ---- 1: every cell's backward (recompute z, LayerNorm / gate gradients) in one launch per width
            tasks = {}
            for v, d in self.var.items():
                cell = self._RNN_cells[v]
                h_t, c_t = tape.h(v, t), tape.C[v][t]
                if native:
                    if folded[v] is not None:
                        task = cell.backward_task_bf16(None, h_t, c_t, dH[v], dC[v], DZ[v][k], ndC[v], ws[v],
                                                       adj=mats[folded[v]["mat"]], zx=tape.zx(v, t))
                    else:
                        task = cell.backward_task_bf16(tape.x(v, t), h_t, c_t, dH[v], dC[v], DZ[v][k], ndC[v], ws[v])
                elif folded[v] is not None:
                    zx_t = tape.zx(v, t)
                    keep += [h_t, zx_t]
                    task = cell.gather_backward_task(mats[folded[v]["mat"]], zx_t, h_t, c_t, dH[v], dC[v], DZ[v][k], ndC[v],
                                                     ws[v], dh_in=ndH[v], defer=True, arith=bwd_arith)
                elif pushed[v]:
                    kp, zb = cell.pushed_bias_pack(push[v]["mlp"], arith="h2")
                    # (d == dx == 64: [d(aggregate) | dh] = dz K'^T rides in this launch as a second phase of the task's
                    # workgroups -- one launch less per step)
                    data = (cell.pushed_kernel_t_h2(push[v]["mlp"]), dX[v], ndH[v]) if fused_data[v] else None
                    task = cell.pushed_backward_task(tape.x(v, t), h_t, c_t, dH[v], dC[v], DZ[v][k], ndC[v], ws[v], kp, zb,
                                                     push[v]["deg"], defer=True, data=data)
                else:
                    x_t = tape.x(v, t)
                    keep += [h_t, x_t]
                    task = cell.backward_task(x_t, h_t, c_t, dH[v], dC[v], DZ[v][k], ndC[v], ws[v], defer=True,
                                              arith=bwd_arith)
                tasks.setdefault(d, []).append(task)
            for d, ts in tasks.items():
                for j in range(0, len(ts), 4):
                    _lib.call_multi("tspgnn_lnlstm_bwd_multi_" + ("bf16" if native else (bwd_arith or "f32")), ts[j:j + 4], d)
            # ---- 2: data gradients of the cell GEMMs; these WRITE dh, the message paths below ACCUMULATE into it
            for v in self.var:
                cell = self._RNN_cells[v]
                if native and folded[v] is not None:
                    cell.gather_backward_data_bf16(mats[folded[v]["mat"]], DZ[v][k], ndH[v], DZX[v][k], dX[v])
                elif native:
                    cell.backward_data_bf16(DZ[v][k], dX[v], ndH[v])
                elif folded[v] is not None:   # dX[v] becomes the gradient w.r.t. the message y (source rows)
                    cell.gather_backward_data(mats[folded[v]["mat"]], DZ[v][k], None if cell.d == 64 else ndH[v],
                                              DZX[v][k], None if projected[v] else dX[v])
                    # (d == 64: dh was formed by the cell launch; projected: dy = dZx Kx^T is left to the message MLP's launch)
                elif pushed[v]:             # dX[v] becomes the gradient w.r.t. the aggregated last hidden activation
                    if not fused_data[v]:
                        cell.pushed_backward_data(push[v]["mlp"], DZ[v][k], dX[v], ndH[v])
                else:
                    cell.backward_data(DZ[v][k], dX[v], ndH[v])
            # ---- 3: adjoint adjacency products, then every message MLP's data gradient in one launch
            mlp_tasks, rc_tasks, targets = [], [], []
            for v in self.var:
                off = 0
                for i, u in enumerate(self.loop[v]):
                    w = self._update_width(u)
                    if "var" not in u:      # an appended matrix is a constant of the graph: its columns' gradient ends here
                        off += w
                        continue
                    dy = dX[v] if len(self.loop[v]) == 1 else dX[v][:, off:off + w].contiguous()
                    off += w
                    src = u["var"]
                    gather_uv = None
                    if "fun" in u:
                        # graphnn.py:149-151: y = fun(h) ahead of the message MLP.  The gradient w.r.t. fun's OUTPUT is formed
                        # apart (adjoint product, the MLP's backward on its own, never fused with another writer of dh), then
                        # pulled back through fun (_fun_vjp) and added to dh of the source
                        if "mat" in u:
                            dy = mats[u["mat"]].matmul(dy, transpose=not u.get("transpose?", False))
                        if "msg" in u:
                            mlp = self._msg_MLPs[u["msg"]]
                            (acts_t, acts_stride), dpre = tape.acts_at((v, i), t), DPRE[(v, i)]
                            g_out = torch.empty((n[src], self.var[src]), **f32)
                            mlp.backward_data(dy, acts_t, acts_stride, None, dpre[:, k], dpre.stride(0), g_out, accumulate=False,
                                              h2=False)
                        else:
                            g_out = dy
                        ndH[src].add_(self._fun_vjp(u["fun"], tape.h(src, t), g_out))
                        continue
                    if "mat" in u and folded[v] is None:   # adjoint of mat (x) y is mat^T (x) dy and vice versa
                        adj = mats[u["mat"]]
                        if u.get("transpose?", False) and adj.uv is not None and "msg" in u and src not in targets \
                                and self._msg_MLPs[u["msg"]].backward_task_fuses_gather(dy):
                            gather_uv = adj.uv     # the adjoint of the row-sum is a two-row gather: the MLP launch forms it
                        else:
                            dy = adj.matmul(dy, transpose=not u.get("transpose?", False))
                    if "msg" in u and rc.get((v, i)):
                        # the chain is recomputed from its input rows; data and weight gradients in one launch
                        mlp = self._msg_MLPs[u["msg"]]
                        if src in targets:
                            raise NotImplementedError("recomputed message MLP: a second writer of the source's gradient")
                        h_src = tape.h(src, t)
                        keep.append(h_src)
                        task = mlp.backward_rc_task(mlp.n_square - 1, h_src, tape.acts[(v, i)][0, t], dy, ndH[src], True,
                                                    gather_uv=gather_uv, partial=RCP[(v, i)])
                        rc_tasks.append((mlp, task, dy))
                        targets.append(src)
                        continue
                    if "msg" in u:
                        mlp = self._msg_MLPs[u["msg"]]
                        (acts_t, acts_stride), dpre = tape.acts_at((v, i), t), DPRE[(v, i)]
                        keep.append(acts_t)
                        h2 = bool((bwd_arith == "h2" or mlp_h2_native) and self.mlp_backward_h2 and mlp.backward_h2_ok(acts_t))
                        if projected[v]:   # the chain starts from dZx Kx^T, formed inside the launch
                            cell = self._RNN_cells[v]
                            if src not in targets:
                                task = mlp.backward_task(None, acts_t, acts_stride, None, dpre[:, k], dpre.stride(0), ndH[src],
                                                         True, h2=True, pre=(DZX[v][k], cell.kx_t_packed_h2()))
                                mlp_tasks.append(((self.var[src], True), task, None))
                                targets.append(src)
                                continue
                            _lib.call("tspgnn_linear_f32", _lib.ptr(DZX[v][k]), 4 * cell.d, _lib.ptr(cell.kx_t_packed()), None, 0,
                                      _lib.ptr(dy), cell.dx, 0, DZX[v][k].shape[0], _lib.current_stream())
                        if pushed[v]:   # the chain ends at the last hidden activation (a relu layer: masked by its output)
                            task = mlp.backward_prefix_task(dpre.shape[0], dy, acts_t, acts_stride, acts_t[dpre.shape[0] - 1],
                                                            dpre[:, k], dpre.stride(0), ndH[src], True, gather_uv=gather_uv, h2=h2)
                            if task is None or src in targets:
                                raise NotImplementedError("pushed training needs the message MLP's backward in one launch")
                            mlp_tasks.append(((self.var[src], h2), task, dy))
                            targets.append(src)
                            continue
                        task = mlp.backward_task(dy, acts_t, acts_stride, None, dpre[:, k], dpre.stride(0),
                                                 ndH[src], True, gather_uv=gather_uv, h2=h2)
                        if task is None or src in targets:   # several kernels, or a second writer of ndH[src]
                            mlp.backward_data(dy, acts_t, acts_stride, None, dpre[:, k], dpre.stride(0), ndH[src],
                                              accumulate=True, h2=h2)
                        else:
                            mlp_tasks.append(((self.var[src], h2), task, dy))
                            targets.append(src)
                    else:
                        ndH[src].add_(dy)
            by_d = {}
            for key, task, _ in mlp_tasks:
                by_d.setdefault(key, []).append(task)
            for (d, h2), ts in by_d.items():
                for j in range(0, len(ts), 4):
                    _lib.call_multi("tspgnn_mlp_bwd_multi_" + ("h2" if h2 else "f32"), ts[j:j + 4], d)
            for mlp, task, _ in rc_tasks:
                _lib.call("tspgnn_mlp_bwd_rc_h2", ctypes.cast(ctypes.pointer(task), ctypes.c_void_p), mlp.sizes[-1],
                          _lib.current_stream())
            dH, dC = ndH, ndC
            if k == 0:      # the chunk [t, t + CH) is complete
                weight_gradients(t, min(t + CH, T))
        for (v, i), part in RCP.items():
            mlp = self._msg_MLPs[self.loop[v][i]["msg"]]
            mlp.backward_rc_finish(mlp.n_square - 1, part)
        for v in self.var:
            self._RNN_cells[v].backward_finish(ws[v])      # LayerNorm parameters: the deferred per-step partials
            if pushed[v]:
                self._RNN_cells[v].pushed_backward_finish(push[v]["mlp"], push[v]["g_wkx"], push[v]["g_zb"])
        return {v: (dH[v], dC[v]) for v in self.var}
