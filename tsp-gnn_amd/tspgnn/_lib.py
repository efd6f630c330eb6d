"""ctypes binding of libtspgnn.so (the C ABI declared in include/tspgnn.h).

There is no CPU fallback: if the shared object is missing this module raises at import, and
a launch on a machine without an MI355X surfaces as a ``TspgnnError`` carrying the HIP error.
"""
import ctypes
import os

# torch owns device memory and streams, and its wheel bundles its own HIP runtime
# (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be loaded BEFORE libtspgnn.so so
# that the dynamic loader resolves our NEEDED libamdhip64.so.7 to that same runtime; two HIP
# runtimes in one process do not share devices, streams or allocations.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSPGNN_LIB") or os.path.join(_HERE, "libtspgnn.so")   # (TSPGNN_LIB: A/B builds)
ABI_VERSION = 5

c_int, c_uint, c_float, c_void_p, c_char_p, c_longlong = (ctypes.c_int, ctypes.c_uint, ctypes.c_float,
                                                          ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong)

# name -> argtypes; every function returns int.  Mirrors include/tspgnn.h one to one
# (tests/test_abi.py parses the header and checks this table against it).
SIGNATURES = {
    "tspgnn_gather2_sum_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_csr_rowsum_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_spmm_pair_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_void_p],
    "tspgnn_csr_spmm_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_pack_weights_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_mlp_fwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_uint, c_void_p],
    "tspgnn_lnlstm_fwd_f32": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_int, c_void_p],
    "tspgnn_mlp_fwd_multi_f32": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_fwd_multi_f32": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_pack_weights_x3": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_mlp_fwd_multi_x3": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_fwd_multi_x3": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_mlp_fwd_multi_x3": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_pack_weights_h2": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "tspgnn_pack_mlp_h2": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "tspgnn_mlp_fwd_multi_h2": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_mlp_head_fwd_h2": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tspgnn_lnlstm_fwd_multi_h2": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_mlp_fwd_multi_h2": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_mp_loop_h2": [c_void_p, c_int, c_void_p],
    "tspgnn_mp_resident_h2": [c_void_p, c_int, c_void_p],
    "tspgnn_lnlstm_bwd_multi_f32": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_bwd_multi_h2": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_bwd_multi_bf16": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_linear_bf16w_f32": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_wgrad_bf16x_f32": [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "tspgnn_lnlstm_bwd_finish_f32": [c_void_p, c_void_p, c_int, c_void_p],
    "tspgnn_mlp_bwd_multi_f32": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_mlp_bwd_multi_h2": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_mlp_bwd_rc_h2": [c_void_p, c_int, c_void_p],
    "tspgnn_mlp_bwd_rc_finish_f32": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_gather_fwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_gather_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_gather2_sum_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_csr_rowsum_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_mlp_fwd_multi_bf16": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_fwd_multi_bf16": [c_void_p, c_int, c_int, c_void_p],
    "tspgnn_einit_fwd_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_tile_rows_f32": [c_void_p, c_float, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_rowdot_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_segment_mean_f32": [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tspgnn_bce_metrics_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tspgnn_linear_f32": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "tspgnn_lnlstm_bwd_f32": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_mlp_bwd_f32": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p, c_int,
                           c_int, c_int, c_int, c_uint, c_void_p],
    "tspgnn_wgrad_f32": [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "tspgnn_vote_grad_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "tspgnn_rowdot_bwd_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_wcolsum_f32": [c_void_p, c_void_p, c_longlong, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "tspgnn_einit_bwd_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "tspgnn_adam_clip_step_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float,
                                  c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "tspgnn_convert_f32_to_bf16": [c_void_p, c_void_p, c_longlong, c_void_p],
    "tspgnn_convert_bf16_to_f32": [c_void_p, c_void_p, c_longlong, c_void_p],
    "tspgnn_bucket_pack_f32": [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "tspgnn_bucket_unpack_f32": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
}

HOST_FUNCTIONS = ("tspgnn_host_pack_instance", "tspgnn_host_route_cost", "tspgnn_host_csr_by_vertex",
                  "tspgnn_host_read_graph", "tspgnn_host_count_edges", "tspgnn_host_pack_batch", "tspgnn_host_stage_batch")

# size queries: name -> argtypes; these return long long (floats of workspace)
SIZE_QUERIES = {
    "tspgnn_lnlstm_bwd_workspace_floats": [c_int],
    "tspgnn_wgrad_workspace_floats": [c_longlong, c_int, c_int],
    "tspgnn_mlp_bwd_rc_partial_floats": [c_int, c_int],
    "tspgnn_wcolsum_workspace_floats": [c_longlong, c_int],
    "tspgnn_einit_bwd_workspace_floats": [c_int, c_int],
    "tspgnn_adam_workspace_floats": [],
}


class MlpTask(ctypes.Structure):
    """tspgnn_mlp_task (include/tspgnn.h)."""
    _fields_ = [("X", c_void_p), ("wb", c_void_p), ("Y", c_void_p), ("acts", c_void_p), ("acts_stride", c_longlong),
                ("rows", c_int), ("n_layers", c_int), ("relu_mask", c_uint), ("proj_w", c_void_p), ("proj_out", c_void_p),
                ("range_flag", c_void_p)]


class LstmTask(ctypes.Structure):
    """tspgnn_lstm_task (include/tspgnn.h)."""
    _fields_ = [("x", c_void_p), ("dx", c_int), ("h", c_void_p), ("c", c_void_p), ("K", c_void_p), ("ln", c_void_p),
                ("h_out", c_void_p), ("c_out", c_void_p), ("rows", c_int), ("uv", c_void_p), ("Zx", c_void_p),
                ("zbias", c_void_p), ("zscale", c_void_p), ("range_flag", c_void_p), ("z_centered", c_int)]


class CellMlpTask(ctypes.Structure):
    """tspgnn_cell_mlp_task (include/tspgnn.h): a cell update followed by the MLP that consumes the new h."""
    _fields_ = [("cell", LstmTask), ("mlp_wb", c_void_p), ("mlp_layers", c_int), ("relu_mask", c_uint),
                ("mlp_out", c_void_p), ("proj_w", c_void_p), ("proj_out", c_void_p),
                ("state_in_blocked", c_int), ("state_out_blocked", c_int),
                ("mlp_acts", c_void_p), ("mlp_acts_stride", ctypes.c_longlong)]


class MpLoopArgs(ctypes.Structure):
    """tspgnn_mp_loop_args (include/tspgnn.h): the whole T-step loop as one launch."""
    _fields_ = [("e_h0", c_void_p), ("e_c0", c_void_p), ("e_h", c_void_p), ("e_c", c_void_p), ("uv", c_void_p),
                ("e_K", c_void_p), ("e_ln", c_void_p), ("e_mlp_wb", c_void_p), ("e_mlp_layers", c_int),
                ("e_relu_mask", c_uint), ("msg", c_void_p * 2),
                ("v_h0", c_void_p), ("v_c0", c_void_p), ("v_h", c_void_p), ("v_c", c_void_p),
                ("rowptr", c_void_p), ("eid", c_void_p), ("v_K", c_void_p), ("v_ln", c_void_p),
                ("v_zbias", c_void_p), ("v_zscale", c_void_p), ("v_mlp_wb", c_void_p), ("v_mlp_layers", c_int),
                ("v_relu_mask", c_uint), ("v_proj_w", c_void_p), ("zx", c_void_p * 2), ("vagg", c_void_p * 2),
                ("plan", c_void_p), ("counters", c_void_p), ("n_groups", c_int), ("grid", c_int),
                ("M", c_int), ("N", c_int), ("T", c_int), ("z_centered", c_int),
                ("range_flag", c_void_p), ("status", c_void_p), ("trace", c_void_p)]


class MpResidentArgs(ctypes.Structure):
    """tspgnn_mp_resident_args (include/tspgnn.h): the T-step loop as one launch, edge states through memory."""
    _fields_ = [("e_h0", c_void_p), ("e_c0", c_void_p), ("e_h", c_void_p), ("e_c", c_void_p),
                ("e_hs", c_void_p), ("e_cs", c_void_p), ("uv", c_void_p),
                ("e_K", c_void_p), ("e_ln", c_void_p), ("e_mlp_wb", c_void_p), ("e_mlp_layers", c_int),
                ("e_relu_mask", c_uint), ("msg", c_void_p * 2),
                ("v_h0", c_void_p), ("v_c0", c_void_p), ("v_h", c_void_p), ("v_c", c_void_p),
                ("rowptr", c_void_p), ("eid", c_void_p), ("v_K", c_void_p), ("v_ln", c_void_p),
                ("v_zbias", c_void_p), ("v_zscale", c_void_p), ("v_mlp_wb", c_void_p), ("v_mlp_layers", c_int),
                ("v_relu_mask", c_uint), ("v_proj_w", c_void_p), ("zx", c_void_p * 2), ("vagg", c_void_p * 2),
                ("vh", c_void_p * 2),
                ("plan", c_void_p), ("counters", c_void_p), ("n_groups", c_int), ("grid", c_int),
                ("n_slots", c_int), ("lds_words", c_int), ("n_active", c_int), ("flags", c_int),
                ("M", c_int), ("N", c_int), ("T", c_int), ("z_centered", c_int),
                ("range_flag", c_void_p), ("status", c_void_p), ("trace", c_void_p)]


class MlpTaskB(ctypes.Structure):
    """tspgnn_mlp_task_bf16 (include/tspgnn.h)."""
    _fields_ = [("X", c_void_p), ("wb", c_void_p), ("Y", c_void_p), ("rows", c_int), ("n_layers", c_int),
                ("relu_mask", c_uint), ("proj_w", c_void_p), ("proj_out", c_void_p),
                ("acts", c_void_p), ("acts_stride", ctypes.c_longlong), ("x_blocked", c_int), ("y_interleaved", c_int)]


class LstmTaskB(ctypes.Structure):
    """tspgnn_lstm_task_bf16 (include/tspgnn.h)."""
    _fields_ = [("x", c_void_p), ("dx", c_int), ("h", c_void_p), ("c", c_void_p), ("K", c_void_p), ("ln", c_void_p),
                ("h_out", c_void_p), ("c_out", c_void_p), ("rows", c_int), ("uv", c_void_p), ("Zx", c_void_p),
                ("state_in_blocked", c_int), ("state_out_blocked", c_int)]


class LstmBwdTask(ctypes.Structure):
    """tspgnn_lstm_bwd_task (include/tspgnn.h)."""
    _fields_ = [("x", c_void_p), ("dx", c_int), ("h", c_void_p), ("c", c_void_p), ("K", c_void_p), ("ln", c_void_p),
                ("dh_out", c_void_p), ("dc_out", c_void_p), ("dz", c_void_p), ("dc_in", c_void_p), ("ln_grad", c_void_p),
                ("workspace", c_void_p), ("rows", c_int), ("uv", c_void_p), ("Zx", c_void_p),
                ("KT", c_void_p), ("dxh", c_void_p), ("defer_reduce", c_int), ("zbias", c_void_p), ("zscale", c_void_p),
                ("KTg", c_void_p), ("dxg", c_void_p)]


class MlpBwdTask(ctypes.Structure):
    """tspgnn_mlp_bwd_task (include/tspgnn.h)."""
    _fields_ = [("dY", c_void_p), ("wt", c_void_p), ("acts", c_void_p), ("acts_stride", c_longlong), ("Yout", c_void_p),
                ("dpre", c_void_p), ("dpre_stride", c_longlong), ("dX", c_void_p), ("accumulate_dx", c_int),
                ("rows", c_int), ("n_layers", c_int), ("relu_mask", c_uint), ("uv", c_void_p), ("acts_bf16", c_int),
                ("pre_X", c_void_p), ("pre_wt", c_void_p), ("pre_k", c_int)]


class MlpBwdRcTask(ctypes.Structure):
    """tspgnn_mlp_bwd_rc_task (include/tspgnn.h)."""
    _fields_ = [("X", c_void_p), ("wb", c_void_p), ("wt", c_void_p), ("Yout", c_void_p), ("dY", c_void_p), ("uv", c_void_p),
                ("dX", c_void_p), ("accumulate_dx", c_int), ("rows", c_int), ("n_layers", c_int), ("relu_mask", c_uint),
                ("acts", c_void_p), ("acts_stride", c_longlong), ("dpre", c_void_p), ("dpre_stride", c_longlong),
                ("partial", c_void_p)]


class TspgnnError(RuntimeError):
    """A libtspgnn entry point returned a non-zero status."""

    def __init__(self, fn, status, message):
        super().__init__("%s failed with status %d: %s" % (fn, status, message))
        self.fn, self.status, self.message = fn, status, message


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libtspgnn.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tsp-gnn_amd/csrc`; there is no CPU fallback for the tspgnn hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.tspgnn_version.restype = c_int
    lib.tspgnn_version.argtypes = []
    lib.tspgnn_last_error.restype = c_char_p
    lib.tspgnn_last_error.argtypes = []
    lib.tspgnn_h2_weight_scale.restype = c_float
    lib.tspgnn_h2_weight_scale.argtypes = []
    if lib.tspgnn_version() != ABI_VERSION:
        raise ImportError("libtspgnn.so ABI %d != binding ABI %d" % (lib.tspgnn_version(), ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = c_int
        fn.argtypes = argtypes
    for name, argtypes in SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.restype = c_longlong
        fn.argtypes = argtypes
    # host-side packer entry points (plain C, no device work)
    lib.tspgnn_host_pack_instance.restype = c_longlong
    lib.tspgnn_host_pack_instance.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.tspgnn_host_route_cost.restype = ctypes.c_double
    lib.tspgnn_host_route_cost.argtypes = [c_void_p, c_int, c_void_p, c_int]
    lib.tspgnn_host_csr_by_vertex.restype = c_int
    lib.tspgnn_host_csr_by_vertex.argtypes = [c_void_p, c_longlong, c_int, c_void_p, c_void_p]
    lib.tspgnn_host_count_edges.restype = c_int
    lib.tspgnn_host_count_edges.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.tspgnn_host_pack_batch.restype = c_longlong
    lib.tspgnn_host_pack_batch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           ctypes.c_double, c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p]
    lib.tspgnn_host_stage_batch.restype = c_longlong
    lib.tspgnn_host_stage_batch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                            ctypes.c_double, c_int, ctypes.c_double, c_longlong, c_int, c_void_p, c_void_p]
    lib.tspgnn_host_read_graph.restype = c_int
    lib.tspgnn_host_read_graph.argtypes = [c_char_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    return lib


lib = _load()


# Optional per-launch timing (bench.py): when set to a list, every call appends
# (name, start_event, stop_event) recorded on torch's current stream -- the stream the kernels run on.
TIMELINE = None


def call(name, *args):
    """Invoke an entry point; raise TspgnnError on a non-zero status."""
    if TIMELINE is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        status = getattr(lib, name)(*args)
        e1.record()
        TIMELINE.append((name, e0, e1))
    else:
        status = getattr(lib, name)(*args)
    if status != 0:
        raise TspgnnError(name, status, lib.tspgnn_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(query, *args, device=None):
    """A scratch tensor sized by one of the tspgnn_*_workspace_floats queries."""
    n = int(getattr(lib, query)(*args))
    return torch.empty(max(n, 1), dtype=torch.float32, device=device)


def task_array(tasks):
    """ctypes array of MlpTask / LstmTask (build once, launch many times)."""
    return (type(tasks[0]) * len(tasks))(*tasks)


def call_multi(name, tasks, d):
    """Launch a list of MlpTask / LstmTask structures with one tspgnn_*_multi_f32 call."""
    if isinstance(tasks, list):
        tasks = (type(tasks[0]) * len(tasks))(*tasks)
    call(name, ctypes.cast(tasks, c_void_p), len(tasks), d, current_stream())
