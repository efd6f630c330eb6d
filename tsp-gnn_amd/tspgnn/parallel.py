"""Data-parallel helpers for the instance-sharded path.

* ``shard_instances``: split a global batch over ranks.  EV is block diagonal by instance
  (/root/reference/instance_loader.py:56-66), so any partition is valid; shards are balanced by EDGE count
  (the cost of a message-passing step is proportional to M = sum n(n-1)/2, not to the number of graphs) and the
  reference's +/- pair -- the same graph with target cost (1-dev) and (1+dev) at consecutive positions
  (instance_loader.py:21-23,73) -- stays on one rank so labels keep alternating 0/1 inside every shard.
* ``BatchPrefetcher``: packs the next batches on background threads (native packer) and uploads them on side
  streams while the GPU works on the current one (double buffering; batches come out in order); a failure of the worker is re-raised in the
  consumer, and the uploaded tensors are registered with the consumer's stream (record_stream) so that the caching
  allocator does not recycle them under kernels that still read them.
"""
import threading
import weakref

import numpy as np
import torch


def shard_instances(instances, world_size, pair=True):
    """-> list (len world_size) of lists of instance indices.  Greedy longest-processing-time assignment of
    units (pairs of consecutive instances when ``pair``) by edge count; deterministic."""
    n = len(instances)
    step = 2 if pair else 1
    units = []
    for i in range(0, n, step):
        idx = list(range(i, min(i + step, n)))
        cost = sum(int(np.count_nonzero(instances[j][0])) for j in idx)
        units.append((cost, idx))
    order = sorted(range(len(units)), key=lambda k: (-units[k][0], k))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for k in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        shards[r].append(k)
        loads[r] += units[k][0]
    out = []
    for r in range(world_size):
        idx = []
        for k in sorted(shards[r]):     # keep the original order inside a shard (labels alternate 0,1,0,1...)
            idx.extend(units[k][1])
        out.append(idx)
    return out


class _PrefetchState(object):
    """Everything the worker threads of a BatchPrefetcher touch.  The threads hold THIS object, never the prefetcher: an
    abandoned prefetcher is then collected like any other object, and its finaliser (weakref.finalize -> close()) releases
    the workers -- a bound method of the prefetcher as the thread target kept it alive for the life of the process, its
    uploaded batches with it."""

    def __init__(self, sess, batch_iter, time_steps, depth=2, pinned=False, workers=1, pack=None):
        """``pinned``: stage uploads through pinned host memory (non-blocking copies).  Off by default: measured on
        this stack, pinning fresh buffers for every batch costs ~20 ms per C2 batch, while the worker thread's
        blocking copies from pageable memory (the GIL is released) keep up: 2.5 ms per batch end to end.  (Round 4: ONE
        reusable pinned buffer and one asynchronous copy per batch was built and is slower still -- 3.4 vs 1.3 ms per C2
        batch for the prefetcher alone: the CPU's writes INTO pinned memory are the slow part on this stack; not kept.
        ``workers`` > 1 packs faster -- 0.97 vs 1.31 ms per batch -- but its concurrent uploads disturb the forward:
        leave it at 1 unless packing is what you wait for.)"""
        self.sess, self.it, self.T, self.pinned, self.pack = sess, iter(batch_iter), time_steps, bool(pinned), pack
        self.workers = max(1, int(workers))
        self.depth = max(int(depth), self.workers)
        self._lock = threading.Condition()
        self._ready = {}            # sequence number -> (batch, event)
        self._next_in = 0           # next sequence number a worker takes from the iterator
        self._next_out = 0          # next sequence number the consumer hands out
        self._exhausted_at = None   # sequence number at which the iterator ended
        self._error = None
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(self.workers)]
        for t in self._threads:
            t.start()

    def _feed(self, t):
        m = self.sess.model
        EV, W, C, route_exists, n_vertices, n_edges = t
        return {m["EV"]: EV, m["W"]: W, m["C"]: C, m["time_steps"]: self.T, m["route_exists"]: route_exists,
                m["n_vertices"]: n_vertices, m["n_edges"]: n_edges}

    def _work(self):
        stream = torch.cuda.Stream(device=self.sess.device) if self.sess.device.type == "cuda" else None
        seq = None
        try:
            while True:
                seq = None
                with self._lock:
                    # at most ``depth`` batches taken but not yet handed out; the iterator itself is advanced by one
                    # thread at a time (generators are not re-entrant)
                    while self._error is None and self._exhausted_at is None and self._next_in - self._next_out >= self.depth:
                        self._lock.wait()
                    if self._error is not None or self._exhausted_at is not None:
                        return
                    try:
                        item = next(self.it)
                    except StopIteration:
                        self._exhausted_at = self._next_in
                        self._lock.notify_all()
                        return
                    seq = self._next_in
                    self._next_in += 1
                t = self.pack(item) if self.pack is not None else item
                if stream is not None:
                    with torch.cuda.stream(stream):
                        b = self.sess.prepare(self._feed(t), pinned=self.pinned, remember_adjacency=False)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                else:
                    b, ev = self.sess.prepare(self._feed(t), remember_adjacency=False), None
                with self._lock:
                    if self._exhausted_at is None or seq < self._exhausted_at:   # (not after close())
                        self._ready[seq] = (b, ev)
                    self._lock.notify_all()
        except BaseException as exc:   # handed to the consumer: a dead worker must not look like an empty dataset
            with self._lock:
                # ... IN ORDER: the batches ahead of the failed one (other workers may still be packing them) are handed
                # out first, so what the consumer saw before the exception does not depend on the workers' timing
                at = self._next_in if seq is None else seq
                if self._error is None or at < self._error[0]:
                    self._error = (at, exc)
                self._lock.notify_all()

    def close(self):
        """Stop early: the workers end after the batch they are packing, batches already uploaded are dropped (their GPU
        memory returns to the allocator)."""
        with self._lock:
            if self._exhausted_at is None or self._exhausted_at > self._next_out:
                self._exhausted_at = self._next_out
            self._ready.clear()
            self._lock.notify_all()

    def next_batch(self):
        with self._lock:
            while True:
                if self._next_out in self._ready:
                    b, ev = self._ready.pop(self._next_out)
                    self._next_out += 1
                    self._lock.notify_all()
                    break
                if self._error is not None and self._error[0] <= self._next_out:
                    err, self._error = self._error[1], None
                    self._exhausted_at = self._next_out      # (the remaining workers stop)
                    self._ready.clear()                      # batches packed behind the failed one: their GPU memory goes back
                    self._lock.notify_all()
                    raise RuntimeError("BatchPrefetcher worker failed") from err
                if self._exhausted_at is not None and self._next_out >= self._exhausted_at:
                    raise StopIteration
                self._lock.wait()
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)   # the upload must land before the main stream reads it
            # the batch was allocated on the upload stream: tell the caching allocator that the consumer's stream
            # uses it too, or its memory is handed to the next upload while this batch's kernels are still running
            for t in b.tensors():
                t.record_stream(cur)
        return b


class BatchPrefetcher(object):
    """Iterates device-resident batches: ``for dev_batch in BatchPrefetcher(sess, batch_iter, time_steps)``.

    ``batch_iter`` yields create_batch 6-tuples (host) -- or, with ``pack=``, whatever ``pack(item)`` turns into one (e.g.
    lists of instances with ``pack=lambda inst: InstanceLoader.create_batch(inst, dev)``: the packing then runs on the
    workers, in parallel, instead of inside the iterator).  While the caller runs step i on the main stream, ``workers``
    threads pack the next batches (CSR build included; the native packer releases the GIL) and enqueue their uploads on
    side streams; batches come out in the iterator's order."""

    def __init__(self, sess, batch_iter, time_steps, depth=2, pinned=False, workers=1, pack=None):
        """See _PrefetchState.__init__ for the arguments (``pinned``, ``workers``, ``depth``, ``pack``)."""
        self._state = _PrefetchState(sess, batch_iter, time_steps, depth=depth, pinned=pinned, workers=workers, pack=pack)
        self.workers, self.depth = self._state.workers, self._state.depth
        # runs when this handle is collected (or at interpreter exit): the workers hold only the state object
        self._finalizer = weakref.finalize(self, self._state.close)

    def close(self):
        """Stop early: the workers end after the batch they are packing, batches already uploaded are dropped (their GPU
        memory returns to the allocator).  Also runs on leaving a ``with`` block and when the prefetcher is garbage
        collected -- an abandoned iterator does not leave workers blocked on a full queue."""
        self._state.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __iter__(self):
        return self

    def __next__(self):
        return self._state.next_batch()


def stage_layout(M, N, B, plan_ints):
    """Byte offsets (256-byte aligned) of the arrays of one staged batch and the total size: uv, eid, rowptr, wc, labels,
    seg, n_edges (tspgnn_host_stage_batch's order), then the one-launch loop's work plan."""
    sizes = [8 * M, 8 * M, 4 * (N + 1), 8 * M, 4 * B, 4 * (B + 1), 4 * B, 4 * plan_ints]
    off, pos = [], 0
    for nbytes in sizes:
        off.append(pos)
        pos += (nbytes + 255) // 256 * 256
    return off, sizes, max(pos, 256)


def stage_instances(instances, dev, target_cost, M, N, stage_ptr, offsets):
    """tspgnn_host_stage_batch over a list of (Ma, Mw, route): everything Session.prepare would upload, written into the
    buffer at ``stage_ptr`` (the GIL is released inside the call).  Raises like create_batch on malformed instances."""
    import ctypes
    from . import _lib
    from .instance_loader import _KIND
    B = len(instances)
    f64, i64 = np.dtype(np.float64), np.dtype(np.int64)
    mas, mws, routes = [], [], []
    pa, pw, pr, kinds, ns, rls = ([0] * B for _ in range(6))
    for k, (Ma, Mw, route) in enumerate(instances):
        # (the usual case -- C-contiguous int / float64 arrays, the route a list or an int64 vector -- costs a few attribute
        # reads per instance: this loop runs under the interpreter lock, next to the consumer's launches)
        if not (type(Ma) is np.ndarray and Ma.flags.c_contiguous and Ma.dtype in _KIND):
            Ma = np.ascontiguousarray(Ma)
            if Ma.dtype not in _KIND:
                Ma = Ma.astype(np.int64)
        if not (type(Mw) is np.ndarray and Mw.flags.c_contiguous and Mw.dtype == f64):
            Mw = np.ascontiguousarray(Mw, dtype=np.float64)
        if not (type(route) is np.ndarray and route.ndim == 1 and route.flags.c_contiguous and route.dtype == i64):
            route = np.ascontiguousarray(route, dtype=np.int64).reshape(-1)
        if Ma.ndim != 2 or Ma.shape[0] != Ma.shape[1] or Mw.shape != Ma.shape:
            raise ValueError("stage_instances: instance %d: adjacency %s / weight matrix %s" % (k, Ma.shape, Mw.shape))
        mas.append(Ma)
        mws.append(Mw)
        routes.append(route)
        pa[k], pw[k], pr[k] = (Ma.__array_interface__["data"][0], Mw.__array_interface__["data"][0],
                               route.__array_interface__["data"][0])   # (.ctypes builds an object per access)
        kinds[k], ns[k], rls[k] = _KIND[Ma.dtype], Ma.shape[0], route.shape[0]
    ptrs = lambda vals: (ctypes.c_void_p * B)(*vals)
    ints = lambda vals: (ctypes.c_int * B)(*vals)
    off = (ctypes.c_longlong * 7)(*offsets[:7])
    keep = (mas, mws, routes)   # (alive across the call)
    got = _lib.lib.tspgnn_host_stage_batch(ptrs(pa), ints(kinds), ptrs(pw), ints(ns),
                                           ptrs(pr), ints(rls), B, float(dev),
                                           0 if target_cost is None else 1, 0.0 if target_cost is None else float(target_cost),
                                           int(M), int(N), stage_ptr, off)
    del keep
    if got == -2:
        raise IndexError("stage_instances: a route names a vertex outside its graph")
    if got == -3:
        raise ValueError("stage_instances: the batch does not have the %d edges / %d vertices the stager was built for" % (M, N))
    if got != M:
        raise ValueError("stage_instances: malformed instances (status %d)" % got)


class BatchStager(object):
    """Serving path for batches of ONE shape (the same instance sizes in the same order, e.g. BASELINE's 128 x n = 40):

        stager = BatchStager(sess, template_instances, time_steps)
        replay = sess.capture_forward(stager.batch)          # the graph reads the stager's device buffer
        for _ in stager.feed(iterable_of_instance_lists):    # each turn: the next batch is in place
            out = replay()

    Per batch the worker thread makes one native call (tspgnn_host_stage_batch: endpoints, CSR, float32 (weight, cost)
    pairs, labels, segment offsets straight into a PINNED slot, outside the interpreter lock) and enqueues ONE
    asynchronous host-to-device copy on its side stream; the consumer enqueues ONE device-to-device copy into the buffer
    the captured graph is bound to.  BatchPrefetcher + DeviceBatch.copy_from do the same job for arbitrary shapes with eight
    numpy conversions, eight pageable uploads and eight device copies per batch under the interpreter lock -- measured
    (round 4) at 9-17 % behind the resident-batch rate at C2, the one packer thread nearly as slow as the forward itself."""

    def __init__(self, sess, template_instances, time_steps, dev=0.02, target_cost=None, slots=3):
        from .graphnn import DeviceAdjacency, choose_loop_plan, loop_enabled
        from .model import DeviceBatch
        if sess.device.type != "cuda":
            raise RuntimeError("BatchStager needs a GPU session (pinned staging, asynchronous copies)")
        self.sess, self.T, self.dev, self.target_cost = sess, int(time_steps), dev, target_cost
        self.sizes = [int(np.asarray(Ma).shape[0]) for Ma, _, _ in template_instances]
        n_edges = [int(np.count_nonzero(Ma)) for Ma, _, _ in template_instances]
        self.n_edges = np.asarray(n_edges, dtype=np.int32)
        B, M, N = len(self.sizes), int(sum(n_edges)), int(sum(self.sizes))
        plan, self.plan_meta = None, None
        if loop_enabled() and M > 0:
            grid = torch.cuda.get_device_properties(sess.device).multi_processor_count
            grid -= grid % 8
            built = choose_loop_plan(np.concatenate([[0], np.cumsum(n_edges)]), np.concatenate([[0], np.cumsum(self.sizes)]), grid)
            if built is not None:
                plan, self.plan_meta = built
        self.offsets, self.nbytes, total = stage_layout(M, N, B, 0 if plan is None else plan.size)
        self.M, self.N, self.B = M, N, B
        self.slots = max(2, int(slots))
        self.pinned = [torch.empty(total, dtype=torch.uint8).pin_memory() for _ in range(self.slots)]
        if plan is not None:   # the plan follows the block structure, which is the stager's contract: written once per slot
            for p in self.pinned:
                p[self.offsets[7]:self.offsets[7] + 4 * plan.size].view(torch.int32).copy_(torch.from_numpy(plan))
        self.dev_stage = [torch.empty(total, dtype=torch.uint8, device=sess.device) for _ in range(self.slots)]
        self.dev_buf = torch.empty(total, dtype=torch.uint8, device=sess.device)

        def view(k, dtype, shape):
            o, nb = self.offsets[k], self.nbytes[k]
            return self.dev_buf[o:o + nb].view(dtype).view(*shape)
        uv = view(0, torch.int32, (M, 2))
        csr = (torch.arange(0, 2 * M + 1, 2, dtype=torch.int32, device=sess.device), uv.view(-1), None)
        csr_t = (view(2, torch.int32, (N + 1,)), view(1, torch.int32, (2 * M,)), None)
        adj = DeviceAdjacency((M, N), sess.device, csr, csr_t, uv=uv)
        if plan is not None:
            adj.loop_plan = (view(7, torch.int32, (plan.size,)),) + tuple(self.plan_meta)
        b = DeviceBatch()
        b.adj, b.M, b.N, b.B, b.T = adj, M, N, B, self.T
        b.WC, b.labels, b.seg = view(3, torch.float32, (M, 2)), view(4, torch.float32, (B,)), view(5, torch.int32, (B + 1,))
        self.batch = b
        self.load(template_instances)       # the buffer holds a valid batch from the start (capture_forward runs it)

    def load(self, instances):
        """Synchronously place one batch in the device buffer (set-up, tests)."""
        self._stage(instances, 0)
        self.dev_buf.copy_(self.pinned[0], non_blocking=False)
        torch.cuda.synchronize()
        return self.batch

    def _stage(self, instances, slot):
        if len(instances) != self.B or any(len(inst[0]) != n for inst, n in zip(instances, self.sizes)):
            raise ValueError("BatchStager: the batch's instance sizes differ from the template's")
        p = self.pinned[slot]
        stage_instances(instances, self.dev, self.target_cost, self.M, self.N, p.data_ptr(), self.offsets)
        o = self.offsets[6]
        if not np.array_equal(p[o:o + 4 * self.B].view(torch.int32).numpy(), self.n_edges):
            raise ValueError("BatchStager: the batch's edge counts differ from the template's (the work plan and the "
                             "segment layout are fixed per stager)")

    def feed(self, batches):
        """Iterate: every ``next`` leaves the following batch of ``batches`` (lists of (Ma, Mw, route)) in ``self.batch``.
        The worker stages and uploads up to ``slots - 1`` batches ahead."""
        import queue
        q = queue.Queue(maxsize=self.slots - 1)
        free = [None] * self.slots          # per slot: event after which its device stage may be overwritten
        released = [threading.Event() for _ in range(self.slots)]   # ... set once the consumer has recorded that event
        for r in released:
            r.set()
        stop = threading.Event()
        side = torch.cuda.Stream(device=self.sess.device)

        def work():
            try:
                for i, inst in enumerate(batches):
                    if stop.is_set():
                        return
                    slot = i % self.slots
                    while not released[slot].wait(timeout=0.05):   # the consumer has enqueued its copy out of this slot ...
                        if stop.is_set():
                            return
                    released[slot].clear()
                    if free[slot] is not None:
                        free[slot].synchronize()                   # ... and that copy has run
                    self._stage(inst, slot)
                    with torch.cuda.stream(side):
                        self.dev_stage[slot].copy_(self.pinned[slot], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    q.put((slot, ev, None))
                q.put((None, None, None))
            except BaseException as exc:   # noqa: BLE001 -- handed to the consumer
                q.put((None, None, exc))
        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                slot, ev, exc = q.get()
                if exc is not None:
                    raise RuntimeError("BatchStager worker failed") from exc
                if slot is None:
                    return
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                self.dev_buf.copy_(self.dev_stage[slot], non_blocking=True)
                done = torch.cuda.Event()
                done.record(cur)
                free[slot] = done
                released[slot].set()
                yield self.batch
        finally:
            stop.set()
            while t.is_alive():            # unblock a worker waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)
