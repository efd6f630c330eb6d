"""``build_network(d)`` and ``Session``: the operator surface of the reference's model.py
(/root/reference/model.py:9-171) on the MI355X-native path.

``build_network`` returns a dict with the reference's keys (model.py:97-104,123,147-157,167):
placeholders ``EV, W, C, time_steps, route_exists, n_vertices, n_edges`` and fetches
``last_states, predictions, TP, FP, TN, FN, acc, loss, train_step`` plus ``gnn``.  Callers do
exactly what train.py:25-42 does:

    model = build_network(64)
    sess = Session(); sess.run(global_variables_initializer())
    loss, acc, pred, TP, FP, TN, FN = sess.run([model['loss'], ...], feed_dict={model['EV']: EV, ...})

``model['EV']`` accepts the reference's dense float64/float32 matrix (converted once to index
form; it must be a 0/1 matrix with two ones per row, which is what create_batch emits) or a
``SparseEV`` that never materialises the dense matrix.
"""
import math
import threading
import zlib

import numpy as np
import torch

from . import _lib
from . import variables as V
from .graphnn import GEMM_ARITH, GraphNN, LSTMStateTuple, to_f32
from .instance_loader import SparseEV
from .mlp import Mlp

LEARNING_RATE = 2e-5          # model.py:13
L2NORM_SCALING = 1e-10        # model.py:14
GLOBAL_NORM_CLIP = 0.65       # model.py:15
STAT_FETCHES = ("loss", "acc", "TP", "FP", "TN", "FN")


class Placeholder(object):
    def __init__(self, name, dtype, ndim):
        self.name, self.dtype, self.ndim = name, dtype, ndim

    def __repr__(self):
        return "<tspgnn placeholder %s>" % self.name


class Fetch(object):
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<tspgnn fetch %s>" % self.name


class _InitOp(Fetch):
    pass


def global_variables_initializer(seed=0):
    """Fetchable like tf.global_variables_initializer(): ``sess.run(global_variables_initializer())``."""
    op = _InitOp("init")
    op.seed = seed
    return op


class DeviceBatch(object):
    """One packed batch resident on the device (see Session.prepare)."""

    def copy_from(self, other):
        """Overwrite this batch's device tensors with ``other``'s (same shapes: same numbers of graphs, vertices and
        edges) -- lets a captured forward graph, which is bound to THIS batch's buffers, serve a stream of batches."""
        mine, theirs = self.tensors(), other.tensors()
        groups = [None if b.adj.loop_plan is None else b.adj.loop_plan[1:] for b in (self, other)]
        if (self.M, self.N, self.B, self.T) != (other.M, other.N, other.B, other.T) or len(mine) != len(theirs) \
                or groups[0] != groups[1] \
                or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(mine, theirs)):
            raise ValueError("copy_from: the batches differ in shape")
        for a, b in zip(mine, theirs):
            a.copy_(b, non_blocking=True)
        return self

    def tensors(self):
        """Every device tensor of the batch (for stream bookkeeping when it was uploaded on another stream)."""
        out = [getattr(self, k, None) for k in ("WC", "labels", "seg")]
        adj = getattr(self, "adj", None)
        if adj is not None:
            out.append(adj.uv)
            for csr in (adj.csr, adj.csr_t):
                out.extend(csr)
            if adj.loop_plan is not None:    # (the one-launch loop's work plan follows the batch's block structure)
                out.append(adj.loop_plan[0])
        return [t for t in out if torch.is_tensor(t)]


class Network(dict):
    """The dict build_network returns, plus the objects Session needs."""


def build_network(d, store=None, float_dtype=torch.float32):
    """model.py:9-170.  ``float_dtype=torch.bfloat16`` (the reference GraphNN's own float_dtype argument,
    graphnn.py:18) stores the embeddings as bf16 with fp32 accumulation (BASELINE config 5); training in that mode is
    mixed precision: bf16 tape, fp32 gradients and fp32 master variables (GraphNN.backward)."""
    d = int(d)
    store = store if store is not None else V.reset_default_store()
    GNN = Network()
    GNN.d = d
    GNN.float_dtype = float_dtype
    GNN.store = store
    # placeholders (model.py:18-29)
    GNN["route_exists"] = Placeholder("route_exists", np.float32, 1)
    GNN["n_vertices"] = Placeholder("n_vertices", np.int32, 1)
    GNN["n_edges"] = Placeholder("edges", np.int32, 1)
    GNN["EV"] = Placeholder("EV", np.float32, 2)
    GNN["W"] = Placeholder("edge_weight", np.float32, 2)
    GNN["C"] = Placeholder("target_cost", np.float32, 2)
    GNN["time_steps"] = Placeholder("time_steps", np.int32, 0)
    # model.py:33-43
    GNN.edge_init_MLP = Mlp(
        layer_sizes=[d / 8, d / 4, d / 2],
        activations=["relu" for _ in range(3)],
        output_size=d,
        name="E_init_MLP",
        name_internal_layers=True,
        kernel_initializer=V.xavier_uniform,
        bias_initializer=V.zeros_init,
        input_size=2,
        store=store,
    )
    # model.py:47
    store.declare("V_init", (1, d), V.normal_init)
    # model.py:57-94
    GNN["gnn"] = GraphNN(
        {"V": d, "E": d},
        {"EV": ("E", "V")},
        {"V_msg_E": ("V", "E"), "E_msg_V": ("E", "V")},
        {
            "V": [{"mat": "EV", "msg": "E_msg_V", "transpose?": True, "var": "E"}],
            "E": [{"mat": "EV", "msg": "V_msg_E", "var": "V"}],
        },
        name="TSP",
        float_dtype=float_dtype,
        store=store,
    )
    # model.py:107-115
    GNN.E_vote_MLP = Mlp(
        layer_sizes=[d for _ in range(3)],
        activations=["relu" for _ in range(3)],
        output_size=1,
        name="E_vote",
        name_internal_layers=True,
        kernel_initializer=V.xavier_uniform,
        bias_initializer=V.zeros_init,
        input_size=d,
        store=store,
    )
    for key in ("last_states", "predictions", "TP", "FP", "TN", "FN", "acc", "loss", "train_step"):
        GNN[key] = Fetch(key)
    build_network.last = GNN
    return GNN


build_network.last = None


class Session(object):
    """Executes fetches of a network built by build_network on one MI355X.

    ``Session(model=None, device='cuda:0')``: without ``model`` it binds to the most recently
    built network (the TF default-graph behaviour train.py:202-206 relies on).

    There is no CPU compute path.  ``device='cpu'`` is accepted for PLUMBING ONLY -- the variable store, feed
    validation and the data-parallel collectives (which is how the gloo tests exercise them without a GPU); every
    fetch that would launch a kernel raises.

    Data parallelism: when ``torch.distributed`` is initialised (or ``process_group`` is given) with
    more than one rank, ``train_step`` all-reduces ONE bucket -- the flat gradient plus the batch statistics --
    and the fetched loss / acc / TP / FP / TN / FN are those of the global batch; replicas are made identical by a
    broadcast of the variables from rank 0 before the first training step."""

    def __init__(self, model=None, device=None, process_group=None):
        self.model = model if model is not None else build_network.last
        if self.model is None:
            raise RuntimeError("Session: call build_network(d) first")
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("tspgnn.Session needs an MI355X (no HIP device visible); there is no CPU path")
            device = "cuda:%d" % torch.cuda.current_device()
        self.device = torch.device(device)
        self.store = self.model.store
        if not self.store.finalized:
            self.store.finalize(self.device)
        self._adj_cache = threading.local()   # per thread: BatchPrefetcher prepares batches on a worker thread
        self._adam = None
        self.process_group = process_group
        self._replicas_synced = False
        self._synced_assignments = -1
        self._ever_synced = False     # data-parallel: the first train_step broadcasts unconditionally (train_step)
        self.last_range_bits = 0     # the f16x2 guard bits (1 overflow, 2 underflow) of the last flagged batch: diagnostics

    @property
    def world_size(self):
        """Ranks of the data-parallel group (1 without an initialised torch.distributed)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.process_group)

    def _require_gpu(self, what):
        if self.device.type != "cuda":
            raise RuntimeError("tspgnn.Session(device=%r) is plumbing only: %s launches HIP kernels and needs an MI355X; "
                               "there is no CPU path" % (str(self.device), what))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    # ------------------------------------------------------------------ feeding
    def _f32(self, a, pinned=False):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        if pinned and self.device.type == "cuda":
            return t.pin_memory().to(self.device, non_blocking=True)
        return t.to(self.device)

    def _adjacency(self, EV, remember=True):
        """Device adjacency of a feed.  A ``DeviceAdjacency`` is used as is; a ``SparseEV`` is uploaded (CSR built)
        and remembered under a fingerprint of its CONTENT -- shape + CRC of the endpoint array, ~0.3 ms at C2 --
        so a caller that re-feeds the same edges (get_cost's probe loop) pays once while one that refills the
        array in place gets the new graph, as TF re-reads every feed; a dense matrix is converted every time."""
        from .graphnn import DeviceAdjacency
        if isinstance(EV, DeviceAdjacency):
            return EV
        if not isinstance(EV, SparseEV):
            try:
                return DeviceAdjacency.from_sparse_ev(SparseEV.fromdense(EV), self.device)
            except ValueError as e:
                raise ValueError("feed for model['EV']: %s" % e)
        if not remember:     # a stream of fresh batches (BatchPrefetcher): nothing to recognise, skip the fingerprint
            return DeviceAdjacency.from_sparse_ev(EV, self.device)
        uv = np.ascontiguousarray(EV.uv)
        key = (tuple(EV.shape), uv.shape, zlib.crc32(uv.view(np.uint8).reshape(-1)))
        cache = self._adj_cache
        if getattr(cache, "key", None) == key:
            return cache.adj
        adj = DeviceAdjacency.from_sparse_ev(EV, self.device)
        cache.key, cache.adj = key, adj
        return adj

    # ------------------------------------------------------------------ forward
    def prepare(self, feed, pinned=False, remember_adjacency=True):
        """Validates a feed_dict and makes the batch resident in HBM (adjacency in index/CSR form,
        (W,C) pairs, labels, segment offsets).  The returned DeviceBatch can be run many times.
        ``pinned``: stage through pinned host memory with non-blocking copies on the current stream
        (used by parallel.BatchPrefetcher on its side stream).  ``remember_adjacency=False`` skips the content fingerprint under
        which the uploaded adjacency is remembered (_adjacency): a stream of fresh batches never repeats one."""
        m = self.model
        adj = self._adjacency(feed[m["EV"]], remember=remember_adjacency)
        M, N = adj.shape
        W = np.asarray(feed[m["W"]], dtype=np.float32).reshape(-1)
        C = np.asarray(feed[m["C"]], dtype=np.float32).reshape(-1)
        if W.shape[0] != M or C.shape[0] != M:
            raise ValueError("W and C must have one row per edge (%d), got %d and %d" % (M, W.shape[0], C.shape[0]))
        n_edges = np.asarray(feed[m["n_edges"]]).astype(np.int64).reshape(-1)
        n_vertices = np.asarray(feed[m["n_vertices"]]).astype(np.int64).reshape(-1)
        B = n_vertices.shape[0]
        labels = self._f32(np.asarray(feed[m["route_exists"]]).reshape(-1), pinned)
        if n_edges.shape[0] != B or labels.shape[0] != B:
            raise ValueError("route_exists, n_vertices and n_edges must have one entry per problem")
        if int(n_edges.sum()) != M:
            raise ValueError("sum(n_edges)=%d does not match the %d rows of EV" % (int(n_edges.sum()), M))
        if int(n_vertices.sum()) != N:
            raise ValueError("sum(n_vertices)=%d does not match the %d columns of EV" % (int(n_vertices.sum()), N))
        b = DeviceBatch()
        b.adj, b.M, b.N, b.B = adj, M, N, B
        b.T = int(feed[m["time_steps"]])
        b.WC = self._f32(np.stack([W, C], axis=1), pinned)
        b.labels = labels
        b.seg = torch.from_numpy(np.concatenate([[0], np.cumsum(n_edges)]).astype(np.int32)).to(self.device)
        return b

    def forward(self, feed, global_stats=False):
        """Runs model.py:33-157 on the device; returns a dict of device tensors.  Rank-local by default, like the
        reference's inference path (no cross-process coupling: ranks may evaluate different numbers of batches, as
        binary_search.get_cost does).  ``global_stats=True`` is a COLLECTIVE call: every rank of a data-parallel
        session must make it, and loss / acc / TP / FP / TN / FN become those of the global batch (8 floats, one
        all-reduce)."""
        b = feed if isinstance(feed, DeviceBatch) else self.prepare(feed)
        out = self.forward_device(b)
        if global_stats and self.world_size > 1:
            self._reduce_stats_only(out["stats"], b.B)
        return out

    def forward_device(self, b):
        self._require_gpu("a forward pass")
        m, d = self.model, self.model.d
        st = _lib.current_stream()
        E0 = m.edge_init_MLP(b.WC)                                            # model.py:43
        V0 = torch.empty((b.N, d), dtype=torch.float32, device=self.device)  # model.py:48-51
        _lib.call("tspgnn_tile_rows_f32", _lib.ptr(self.store.view("V_init")), 1.0 / math.sqrt(float(d)),
                  _lib.ptr(V0), b.N, d, st)
        last = m["gnn"]({"EV": b.adj}, {"V": V0, "E": E0}, b.T)               # model.py:118-122
        Eh = to_f32(last["E"].h)                                              # (bf16 storage: widened once)
        arith = m["gnn"].active_arith()
        vote = (m.E_vote_MLP.forward_split(Eh, arith) if arith else m.E_vote_MLP(Eh)).view(-1)   # model.py:128
        if arith == "h2" and not m["gnn"].check_h2_weights():     # (the vote head's packing vetoed f16x2: bf16x3)
            vote = m.E_vote_MLP.forward_split(Eh, m["gnn"].active_arith()).view(-1)
        logits = torch.empty(b.B, dtype=torch.float32, device=self.device)
        _lib.call("tspgnn_segment_mean_f32", _lib.ptr(vote), _lib.ptr(b.seg), _lib.ptr(logits), b.B, st)
        pred = torch.empty(b.B, dtype=torch.float32, device=self.device)
        stats = torch.empty(6, dtype=torch.float32, device=self.device)
        _lib.call("tspgnn_bce_metrics_f32", _lib.ptr(logits), _lib.ptr(b.labels), _lib.ptr(pred), _lib.ptr(stats),
                  b.B, st)
        return {"last_states": last, "E_vote": vote, "logits": logits, "predictions": pred, "stats": stats,
                "range_guard": self.store.h2_guard()}

    # ---- f16x2 range guard (GraphNN.active_arith): the default arithmetic's fp16 pieces overflow where the reference's
    # fp32 does not.  Weights are vetted when they are packed; an ACTIVATION that leaves the range sets bit 0 of the
    # guard word on the device, which every synchronous entry point (run) reads next to the values it fetches anyway
    # and answers by re-running the batch on bf16x3.  forward() / forward_device() / capture_forward() return device
    # tensors without synchronising: their callers get the word as out["range_guard"] and range_exceeded().
    def range_exceeded(self, clear=True):
        """True if an f16x2 launch since the last call flagged an operand outside the fp16 range (synchronises)."""
        if self.device.type != "cuda":
            return False
        guard = self.store.h2_guard()
        words = guard[:3].tolist()
        if words[2]:                        # tspgnn_mp_loop_h2 gave up waiting: its workgroups were not all resident
            guard[2:3].zero_()
            raise RuntimeError("tspgnn_mp_loop_h2: a wait inside the one-launch loop timed out (status %d); the launch's "
                               "outputs are invalid -- set TSPGNN_LOOP=0 to run the stepwise launches" % words[2])
        bits = words[0] & 3                 # bit 0: an operand beyond fp16's largest value; bit 1: a gate row whose spread
        if bits:                            # is below the absolute error of its operands' fp16 pieces (h2_tile.h)
            self.last_range_bits = bits
            if clear:
                guard[0:1].zero_()
        return bool(bits)

    def capture_forward(self, batch):
        """Captures one forward pass over a resident batch into a HIP graph (hipStreamBeginCapture via
        torch.cuda.CUDAGraph: every libtspgnn launch goes to torch's current stream, so the capture sees
        all of them) and returns ``replay() -> outputs``.  The T-step loop is ~200 launches of 5-90 us
        kernels; replaying the graph removes the per-launch host cost.  The captured graph reads the
        packed weights that were current at capture time: re-capture after the variables change."""
        b = batch if isinstance(batch, DeviceBatch) else self.prepare(batch)
        self.forward_device(b)          # warm-up: builds packed-weight caches (and vets their range) outside the capture
        if self.range_exceeded():       # this batch's activations leave the f16x2 range: capture it on bf16x3
            with self.model["gnn"].forced_off_h2():
                return self.capture_forward(b)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.forward_device(b)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            out = self.forward_device(b)
        version = self.store.version

        def replay():
            if self.store.version != version:
                raise RuntimeError("variables changed since capture_forward(): capture again")
            graph.replay()
            return out
        replay.graph = graph
        replay.batch = b    # the graph reads THIS batch's device buffers on every replay: they live as long as the closure does
        replay.range_exceeded = self.range_exceeded   # (synchronising) did a replay leave the f16x2 range? see run()
        return replay

    # ------------------------------------------------------------------ run
    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        for f in flist:
            if isinstance(f, _InitOp):
                self.store.initialize(seed=f.seed)
                self._reset_optimizer_slots()   # tf.global_variables_initializer also resets <var>/Adam, beta powers
        names = [f.name for f in flist if isinstance(f, Fetch) and not isinstance(f, _InitOp)]
        out = None
        if names:
            if feed_dict is None:
                raise ValueError("these fetches depend on placeholders: feed_dict is required")
            missing = [k for k in ("EV", "W", "C", "time_steps", "route_exists", "n_vertices", "n_edges")
                       if self.model[k] not in feed_dict]
            if missing:
                raise ValueError("You must feed a value for placeholder(s) %s" % ", ".join(missing))
            if "train_step" in names:
                out = self.train_step(feed_dict)   # (looks after the f16x2 range flag itself)
            else:
                # a statistic of the batch is a statistic of the GLOBAL batch in a data-parallel session (collective:
                # every rank fetches it, as run_batch does); predictions / last_states alone stay rank-local
                gs = any(n in STAT_FETCHES for n in names)
                out = self.forward(feed_dict, global_stats=gs)
                if self.range_exceeded():   # an activation left the fp16 range: this batch again on bf16x3 (fp32's range)
                    with self.model["gnn"].forced_off_h2():
                        out = self.forward(feed_dict, global_stats=gs)
        results = []
        stats = None
        for f in flist:
            if isinstance(f, _InitOp) or f.name == "train_step":
                results.append(None)
                continue
            if stats is None:
                stats = out["stats"].cpu().numpy()
            if f.name == "predictions":
                results.append(out["predictions"].cpu().numpy())
            elif f.name == "last_states":
                results.append({v: LSTMStateTuple(c=s.c.cpu().numpy(), h=s.h.to(torch.float32).cpu().numpy())
                                for v, s in out["last_states"].items()})
            else:
                idx = {"loss": 0, "acc": 1, "TP": 2, "FP": 3, "TN": 4, "FN": 5}[f.name]
                results.append(np.float32(stats[idx]))
        return results[0] if single else results

    # ------------------------------------------------------------------ training (model.py:160-167)
    def loss_and_grads(self, feed, keep_tape=False):
        """Forward + backward of ``loss`` (the L2 term is added by the optimiser kernel).  Leaves the
        gradient of the local mean loss in ``store.grad``; returns the forward outputs (``keep_tape``: plus the
        message passing's tape under "tape", for inspection -- it holds every step's states)."""
        self._require_gpu("a training step")
        m, d, st = self.model, self.model.d, _lib.current_stream()
        b = feed if isinstance(feed, DeviceBatch) else self.prepare(feed)
        store = self.store
        store.zero_grad()
        f32 = dict(dtype=torch.float32, device=self.device)
        E0 = m.edge_init_MLP(b.WC)
        V0 = torch.empty((b.N, d), **f32)
        _lib.call("tspgnn_tile_rows_f32", _lib.ptr(store.view("V_init")), 1.0 / math.sqrt(float(d)), _lib.ptr(V0),
                  b.N, d, st)
        last, tape = m["gnn"].forward_train({"EV": b.adj}, {"V": V0, "E": E0}, b.T)
        # vote head: three relu Dense(d) + Dense(1) (model.py:107-115,128), keeping the hidden activations
        mv = m.E_vote_MLP
        n_sq = mv.n_square
        EhT = to_f32(last["E"].h)                # (bf16 storage: widened once; the vote head is fp32)
        Y3 = torch.empty((b.M, d), **f32)
        acts = torch.empty((max(n_sq - 1, 1), b.M, d), **f32)
        mv.forward_saving(EhT, Y3, acts, acts.stride(0))
        head = mv.layer_names[-1]
        vote = torch.empty(b.M, **f32)
        _lib.call("tspgnn_rowdot_f32", _lib.ptr(Y3), _lib.ptr(store.view(head + "/kernel")),
                  _lib.ptr(store.view(head + "/bias")), _lib.ptr(vote), b.M, d, st)
        logits = torch.empty(b.B, **f32)
        _lib.call("tspgnn_segment_mean_f32", _lib.ptr(vote), _lib.ptr(b.seg), _lib.ptr(logits), b.B, st)
        pred = torch.empty(b.B, **f32)
        stats = torch.empty(6, **f32)
        _lib.call("tspgnn_bce_metrics_f32", _lib.ptr(logits), _lib.ptr(b.labels), _lib.ptr(pred), _lib.ptr(stats), b.B,
                  st)
        # ---- backward: loss -> votes -> vote head -> message passing -> initial embeddings
        dvote = torch.empty(b.M, **f32)
        _lib.call("tspgnn_vote_grad_f32", _lib.ptr(logits), _lib.ptr(b.labels), _lib.ptr(b.seg), _lib.ptr(dvote), b.B,
                  st)
        dY3 = torch.empty((b.M, d), **f32)
        _lib.call("tspgnn_rowdot_bwd_f32", _lib.ptr(dvote), _lib.ptr(store.view(head + "/kernel")), _lib.ptr(dY3), b.M,
                  d, st)
        ws = _lib.workspace("tspgnn_wcolsum_workspace_floats", b.M, d, device=self.device)
        _lib.call("tspgnn_wcolsum_f32", _lib.ptr(Y3), _lib.ptr(dvote), b.M, d, 1.0,
                  _lib.ptr(store.grad_view(head + "/kernel")), _lib.ptr(store.grad_view(head + "/bias")), _lib.ptr(ws),
                  st)
        dpre = torch.empty((n_sq, b.M, d), **f32)
        dEh = torch.empty((b.M, d), **f32)
        mv.backward_data(dY3, acts, acts.stride(0), Y3, dpre, dpre.stride(0), dEh, accumulate=False)
        mv.backward_weights([EhT] + [acts[l] for l in range(n_sq - 1)], [dpre[l] for l in range(n_sq)], b.M)
        d0 = m["gnn"].backward(tape, {"E": (dEh, None)})
        dE0, dV0 = d0["E"][0], d0["V"][0]
        mi = m.edge_init_MLP
        ws = _lib.workspace("tspgnn_einit_bwd_workspace_floats", b.M, d, device=self.device)
        _lib.call("tspgnn_einit_bwd_f32", _lib.ptr(b.WC), _lib.ptr(mi.wb()), _lib.ptr(dE0),
                  _lib.ptr(store.grad_span(mi.layer_names[0] + "/kernel", mi.layer_names[-1] + "/bias")), _lib.ptr(ws),
                  b.M, d, st)
        if dV0 is not None:   # None: zero message-passing steps, the loss does not see V_init
            ws = _lib.workspace("tspgnn_wcolsum_workspace_floats", b.N, d, device=self.device)
            _lib.call("tspgnn_wcolsum_f32", _lib.ptr(dV0), None, b.N, d, 1.0 / math.sqrt(float(d)),
                      _lib.ptr(store.grad_view("V_init")), None, _lib.ptr(ws), st)
        out = {"last_states": last, "E_vote": vote, "logits": logits, "predictions": pred, "stats": stats, "batch": b}
        if keep_tape:
            out["tape"] = tape
        return out

    # The data-parallel step.  The loss is a mean over the GLOBAL batch (model.py:157), so rank r's
    # gradient of its local mean counts with weight B_r / B.  Everything that must cross ranks rides in ONE bucket:
    #   bucket = [ B_r * grad_r | B_r, B_r*loss_r, B_r*acc_r, TP_r, FP_r, TN_r, FN_r, guard bits 0, 1, 2 ]  (VariableStore.bucket)
    # one all-reduce(sum) of it (RCCL over xGMI with the 'nccl' backend; 462 KB at d=64), then every rank divides by the
    # reduced B on the device: no host round trip, so the two HIP graphs of capture_train_step run back to back around
    # the collective.  The L2 term, the clip by the GLOBAL norm and Adam follow on the reduced gradient, identically
    # on every rank (model.py:163-167).  Tensor ops only (device-agnostic): the gloo tests run this on CPU.
    def _pack_bucket(self, local_batch, stats, with_grad):
        store = self.store
        if store.grad is None:
            store.zero_grad()
        nb = float(local_batch)
        n = store.theta.numel()
        tail = store.bucket[n:]
        if self.device.type == "cuda":     # one launch (tspgnn_bucket_pack_f32): no at::native node in the captured step
            _lib.call("tspgnn_bucket_pack_f32", _lib.ptr(store.bucket), n, 1 if with_grad else 0, nb, _lib.ptr(stats),
                      store.h2_flag_ptr(), _lib.current_stream())
            return tail
        # device="cpu" (plumbing: the gloo tests): the same arithmetic as tensor ops
        if with_grad:
            store.grad.mul_(nb)
        tail.zero_()
        tail[0:1].fill_(nb)
        if stats is not None:
            tail[1:3].copy_(stats[0:2])
            tail[1:3].mul_(nb)
            tail[3:7].copy_(stats[2:6])
        word = store.h2_guard()[0:1]             # guard bits, one slot each: every rank must skip / repeat the step together
        for k in range(3):
            tail[7 + k:8 + k].copy_((word >> k) & 1)
        return tail

    def _unpack_bucket(self, stats, with_grad):
        store = self.store
        n = store.theta.numel()
        if self.device.type == "cuda":
            _lib.call("tspgnn_bucket_unpack_f32", _lib.ptr(store.bucket), n, 1 if with_grad else 0, _lib.ptr(stats),
                      store.h2_flag_ptr(), _lib.current_stream())
            return
        tail = store.bucket[n:]
        inv = torch.reciprocal(tail[0:1])
        if with_grad:
            store.grad.mul_(inv)
        if stats is not None:
            stats[0:2].copy_(tail[1:3] * inv)
            stats[2:6].copy_(tail[3:7])
        store.h2_guard()[0:1].copy_((tail[7:8] != 0).to(torch.int32) + 2 * (tail[8:9] != 0).to(torch.int32)
                                    + 4 * (tail[9:10] != 0).to(torch.int32))

    def allreduce_grads(self, local_batch, stats=None):
        """One all-reduce of [gradient | batch size, statistics]; afterwards ``store.grad`` holds the gradient of
        the global mean loss and ``stats`` (the 6-vector loss, acc, TP, FP, TN, FN of bce_metrics; optional) the
        statistics of the global batch.  No-op with a single rank."""
        if self.world_size == 1:
            return
        import torch.distributed as dist
        self._pack_bucket(local_batch, stats, True)
        dist.all_reduce(self.store.bucket, group=self.process_group)
        self._unpack_bucket(stats, True)

    def _reduce_stats_only(self, stats, local_batch):
        import torch.distributed as dist
        tail = self._pack_bucket(local_batch, stats, False)
        dist.all_reduce(tail, group=self.process_group)
        self._unpack_bucket(stats, False)

    def allreduce_host_sums(self, values):
        """Sum of a small float64 host vector over the ranks (run_batch's label / prediction means)."""
        v = np.asarray(values, dtype=np.float64)
        if self.world_size == 1:
            return v
        import torch.distributed as dist
        t = torch.from_numpy(v.copy()).to(self.device)
        dist.all_reduce(t, group=self.process_group)
        return t.cpu().numpy()

    def broadcast_variables(self, src=0):
        """Replicas start identical: theta AND the optimiser slots (m, v, step counter) from rank ``src`` to every
        rank.  Every rank issues the same four broadcasts whether or not it already holds optimiser state (a
        checkpoint restored on rank 0 only creates the slots there), so the collective sequence never depends on
        the rank."""
        if self.world_size == 1:
            self._synced_assignments = self.store.assignments
            return
        import torch.distributed as dist
        self._ensure_adam()
        group_src = src if self.process_group is None else dist.get_global_rank(self.process_group, src)
        dist.broadcast(self.store.theta, src=group_src, group=self.process_group)
        for k in ("m", "v", "t"):
            dist.broadcast(self._adam[k], src=group_src, group=self.process_group)
        self._adam["step"] = int(self._adam["t"].item())   # host mirror (save_weights' beta powers)
        self.store.touch()
        self._replicas_synced = True
        self._synced_assignments = self.store.assignments

    def _sync_replicas_once(self):
        """Broadcast before the first training step and again after every outside assignment of the variables
        (initialiser, load_weights / store.load) -- those happen on every rank or on rank 0 only, and either way the
        replicas must leave this call identical.  Collective when it fires: assignments must be made (or not made)
        at the same points of the program on every rank, like the training steps themselves."""
        need = not self._replicas_synced or self._synced_assignments != self.store.assignments
        if self.world_size > 1:
            # the decision is made TOGETHER (one 1-int all-reduce): a restore on rank 0 only bumps rank 0's counter alone,
            # and a rank that broadcast while its peers went on to the gradient all-reduce would hang the job
            import torch.distributed as dist
            flag = torch.tensor([1 if need else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.process_group)
            need = bool(int(flag.item()))
        if need:
            self.broadcast_variables(0)

    def apply_gradients(self):
        """g += 1e-10*theta; clip_by_global_norm(0.65); Adam(lr=2e-5) -- one fused pass over theta."""
        store = self.store
        self._ensure_adam()
        a = self._adam
        a["step"] += 1   # host mirror; the device counter a["t"] is what the kernel uses (graph-replayable)
        b1, b2, eps = 0.9, 0.999, 1e-8     # tf.train.AdamOptimizer defaults
        _lib.call("tspgnn_adam_clip_step_f32", _lib.ptr(store.theta), _lib.ptr(store.grad), _lib.ptr(a["m"]),
                  _lib.ptr(a["v"]), store.theta.numel(), L2NORM_SCALING, GLOBAL_NORM_CLIP, LEARNING_RATE, b1, b2, eps,
                  _lib.ptr(a["gnorm"]), _lib.ptr(a["ws"]), _lib.ptr(a["t"]),
                  store.h2_flag_ptr() if self.device.type == "cuda" else None, _lib.current_stream())
        store.touch()
        return a["gnorm"]

    def _ensure_adam(self):
        store = self.store
        if self._adam is None:
            self._adam = {"m": torch.zeros_like(store.theta), "v": torch.zeros_like(store.theta), "step": 0,
                          "gnorm": torch.zeros(1, dtype=torch.float32, device=self.device),
                          "t": torch.zeros(1, dtype=torch.int32, device=self.device),
                          "ws": _lib.workspace("tspgnn_adam_workspace_floats", device=self.device)}

    def _reset_optimizer_slots(self):
        """Adam's m, v and step counter back to zero, in place (captured graphs keep pointing at the same tensors)."""
        if self._adam is not None:
            for k in ("m", "v", "t", "gnorm"):
                self._adam[k].zero_()
            self._adam["step"] = 0
        self._replicas_synced = False

    def train_step(self, feed):
        """One ``sess.run(train_step)``: forward, backward, (all-reduce), L2 + clip + Adam.

        The guard word is this call's business, whoever the caller is: it is cleared first (a flag left behind by an
        earlier, unchecked forward() must not make the optimiser kernel skip this step -- and every later one), read
        once after the optimiser launch (one blocking 4-byte read per eager step), and a flagged step -- which the
        optimiser kernel skipped on the device -- is repeated: on bf16x3 when an f16x2 launch left its range (bits 0 / 1),
        after a broadcast of rank 0's variables when some rank's variables were assigned since the replicas were last
        made identical (bit 2).  In a data-parallel session the word every rank reads is the all-reduced one, so all ranks
        repeat together -- and the "do we need to re-synchronise?" question rides in the step's ONE all-reduce instead of
        a collective and a host round trip of its own before every step: the price is one wasted forward +
        backward in the step after a checkpoint restore."""
        on_gpu = self.device.type == "cuda"
        dp = self.world_size > 1
        if dp and not self._ever_synced:
            # the session's first training step: every rank is here for the first time, the broadcast needs no agreement
            self.broadcast_variables(0)
            self._ever_synced = True
        need = dp and (not self._replicas_synced or self._synced_assignments != self.store.assignments)
        h2 = on_gpu and self.model["gnn"].active_arith() == "h2"   # (only f16x2 launches raise bits 0 / 1)
        if on_gpu:
            self.store.h2_guard()[0:1].fill_(4 if need else 0)     # always: the optimiser kernel skips on a non-zero word
        out = self._train_step_once(feed)
        bits = self._guard_word() if on_gpu and (h2 or dp) else 0
        if bits & 4:   # skipped everywhere: some replica had been assigned to.  Make them identical, then take the step.
            self._adam["step"] -= 1
            self.broadcast_variables(0)
            out = self._train_step_once(feed)
            bits = self._guard_word() if h2 else 0
        if bits & 3:
            self.last_range_bits = bits & 3
            self._adam["step"] -= 1     # (Adam skipped the update on the device: theta, m, v and t are untouched)
            with self.model["gnn"].forced_off_h2():
                out = self._train_step_once(feed)
        return out

    def _guard_word(self):
        """The guard word (VariableStore.h2_guard()[0]) after a step, cleared if set: one blocking 4-byte read."""
        guard = self.store.h2_guard()
        w = int(guard[0].item())
        if w:
            guard[0:1].zero_()
        return w

    def _train_step_once(self, feed):
        out = self.loss_and_grads(feed)
        self.allreduce_grads(out["batch"].B, out["stats"])
        out["global_norm"] = self.apply_gradients()
        return out

    def capture_train_step(self, batch):
        """HIP-graph replay of the training step on a resident batch: graph A = zero grads + forward +
        backward (+ re-packing of the weights, which change every step), then the gradient all-reduce
        (eager: RCCL), then graph B = L2 + clip + Adam with the step counter on the device.  ~600 kernel
        launches per step stop costing host time.  Returns ``replay() -> outputs``."""
        b = batch if isinstance(batch, DeviceBatch) else self.prepare(batch)
        self._ensure_adam()
        self._sync_replicas_once()
        gnn, store = self.model["gnn"], self.store
        store.h2_guard()[0:1].zero_()         # (a flag left by an earlier unchecked forward() is not this capture's)
        self.loss_and_grads(b)                # warm-up outside the capture (allocator, caches)
        torch.cuda.synchronize()
        # did the warm-up leave the f16x2 range?  Then the step is captured on bf16x3 (and stays there for these
        # variables).  Decided together in a data-parallel session: the ranks' batches differ, their graphs must not.
        bits = int(store.h2_guard()[0].item()) & 3
        if self.world_size > 1:
            import torch.distributed as dist
            agreed = torch.tensor([bits], dtype=torch.int32, device=self.device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MAX, group=self.process_group)
            bits = int(agreed.item())
        if bits and gnn.active_arith() == "h2":
            self.last_range_bits = bits
            store.h2_guard()[0:1].zero_()
            gnn._h2_off_at = store.assignments
            return self.capture_train_step(b)
        side = torch.cuda.Stream(device=self.device)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.store.version += 1               # force every packed-weight copy to be rebuilt INSIDE graph A
        with torch.cuda.graph(ga, stream=side):
            out = self.loss_and_grads(b)
        with torch.cuda.graph(gb, stream=side):
            gnorm = self.apply_gradients()
        self._adam["step"] -= 1               # capture does not execute: undo the host mirror's increment
        out["global_norm"] = gnorm

        # f16x2 range guard under replay: the packings happen inside graph A, so the guard words are looked at LATE --
        # replay i looks at the words as replay i - LAG left them, through a pinned copy and an event recorded then (long
        # complete: no stall).  A FIXED lag, not "whenever the copy happens to have landed": the flag word rides in the
        # all-reduced bucket and the weights are replicated, so every rank of a data-parallel session sees the same words
        # at the same replay index and raises at the same index -- no rank is left alone in the next all-reduce.  Looking
        # late is safe: the weight word trips at half the fp16 range (Adam moves a weight by ~lr per step), and a flagged
        # step -- and every step after it, the flag stays up -- was skipped by the optimiser kernel on the device.
        LAG = 2
        ring = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(LAG + 1)]
        inflight = []                         # [(event, pinned words)] of the last LAG replays, oldest first
        count = [0]

        dead = [None]    # once the guard has fired this closure is finished: graph A holds the f16x2 kernels

        def replay():
            if dead[0] is not None:
                raise RuntimeError(dead[0])
            if len(inflight) >= LAG:
                ev, words = inflight.pop(0)
                ev.synchronize()
                act, weight = int(words[0]) & 3, int(words[1]) >= store.H2_WEIGHT_LIMIT_BITS
                if act or weight:
                    torch.cuda.synchronize()
                    self.last_range_bits = act
                    gnn._h2_off_at = store.assignments
                    gnn._mlp_h2_native_ok = False
                    store.h2_guard().zero_()
                    del inflight[:]
                    self._adam["step"] = int(self._adam["t"].item())   # the device counter did not count the skipped steps
                    if act:
                        why = ("an activation left the fp16 range of the f16x2 split (guard bits %d): that step and the "
                               "replays after it were skipped by the optimiser kernel, the variables are those of the "
                               "last clean step" % act)
                    else:
                        why = ("a weight passed half the fp16 range of the f16x2 packing: the steps so far WERE applied, "
                               "the next ones would not be safe")
                    dead[0] = ("f16x2 range exceeded during replayed training steps: %s.  This replay closure is finished "
                               "(its graph holds the f16x2 kernels): call capture_train_step() again -- it will run on "
                               "bf16x3 (bf16 storage: with the message MLPs' backward on the fp32 matrix instruction)" % why)
                    raise RuntimeError(dead[0])
            ga.replay()
            self.allreduce_grads(b.B, out["stats"])   # device-side only: no host sync between the two graphs
            gb.replay()
            if gnn.training_packs_h2():
                words = ring[count[0] % (LAG + 1)]
                count[0] += 1
                words.copy_(store.h2_guard(), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                inflight.append((ev, words))
            self._adam["step"] += 1
            self.store.touch()
            return out
        replay.graphs = (ga, gb)
        return replay
