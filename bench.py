#!/usr/bin/env python
"""Benchmark of the TSP-GNN message-passing hot path on MI355X (contract: task prompt + SURVEY.md §8d).

One *step* of this bench = one forward pass of the hot path over one synthetic batch of the
workload (default C2: 128 complete Euclidean graphs of n=40, d=64, T=32 message-passing steps,
fp32), inputs already resident in HBM.  ``value`` = message-passing steps per second of the whole
job (= n_gpus * K * T / max-over-ranks wall time); one message-passing step aggregates 4M
incidences (E<-V gather of M rows + V<-E row-sum over 2M incidences) and updates both LSTMs.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

The forward pass needs no collective (EV is block diagonal: instances never exchange messages), so
N>1 is weak scaling over independent shards of the global batch: "sharded by instance".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BF16_MFMA_PEAK_TF = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* peak = fp32 vector peak

WORKLOADS = {
    # name: (graph sizes, d, T, storage)
    "c1": ([20] * 32, 64, 8, "f32"),      # BASELINE.json configs[0]
    "c2": ([40] * 128, 64, 32, "f32"),    # BASELINE.json configs[1]: the configuration the metric is quoted on
    # BASELINE.json configs[3]: ragged n in {20..80}, batch 512 (N=25 362, M=695 849; the SpMM operands exceed the
    # 256 MB Infinity Cache).  Not the metric's configuration; no cpu_baseline (the dense EV would be 70 GB).
    "c4": (list(np.random.RandomState(0).randint(20, 81, size=512)), 64, 32, "f32"),
    # BASELINE.json configs[4], ONE GPU's shard of it: n=200, 32 of the 256 graphs, embed=128, T=64, bf16 embeddings
    # with fp32 accumulation (M=636 800 edges; 340 MB of SpMM operands per step).  Not the metric's configuration.
    "c5": ([200] * 32, 128, 64, "bf16"),
    # development: C2's batch in the bf16-storage mode (what that mode costs / saves at d = 64)
    "c2b": ([40] * 128, 64, 32, "bf16"),
}


def spmm_bytes(N, M, d, eb=4, ib=4):
    """Algorithmic (compulsory) bytes of the two aggregation kernels, SURVEY.md §8d M4."""
    gather = N * d * eb + 2 * M * ib + M * d * eb
    rowsum = M * d * eb + (2 * M + N + 1) * ib + N * d * eb
    return gather, rowsum


def dense_flops_per_step(N, M, d, folded=True):
    """MFMA work per message-passing step.  Reference op count: two 4-layer d x d MLPs and two [2d,4d]
    LSTM GEMMs on M+N rows.  Executed (folded=True): the x-half of the edge cell's GEMM runs on the N
    vertex rows ((EV y)Kx = EV(y Kx)) instead of the M edge rows, and the edge message MLP's last (linear)
    layer is pushed through the row-sum into the vertex cell's kernel (3 layers on the M edge rows)."""
    if not folded:
        return (M + N) * 4 * 2 * d * d + (M + N) * 2 * 2 * d * 4 * d
    mlp = M * 3 * 2 * d * d + N * 4 * 2 * d * d
    return mlp + M * 2 * d * 4 * d + N * 2 * 2 * d * 4 * d + N * 2 * d * 4 * d


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03")   # newest first: the rocprofv3 summaries bench.py quotes (profiles/<round>_<workload>_...)


def csrc_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources the shipped library is built from (tsp-gnn_amd/csrc/*.hip,
    *.h, sorted by name).  tools/profile_r05_all.sh writes it into every rocprofv3 summary it commits (`# csrc_sha16:`
    line): a summary whose fingerprint differs from the tree's was taken on OTHER kernels and is not quoted as this
    build's figure (the committed profile could go stale unnoticed)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "tsp-gnn_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profile_kernel_stats(workload):
    """-> (relative path, {kernel name: (calls, avg_us)}) from the newest committed rocprofv3 --kernel-trace --stats
    summary of THIS workload's replayed forward (profiles/rNN_<workload>_forward_kernel_stats.txt, written by
    profiles/summarize_rocpd.py from `rocprofv3 --kernel-trace --stats -- python tools/forward_graph.py <workload> 20`).
    rocprofv3 cannot run inside bench.py; the file travels with the repo, so every figure quoted from it can be
    recomputed from that one named file."""
    for rnd in PROFILE_ROUNDS:
        rel = "profiles/%s_%s_forward_kernel_stats.txt" % (rnd, workload)
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        out = {}
        with open(path) as f:
            for line in f:
                if line.startswith("# csrc_sha16:"):
                    out["__csrc_sha16__"] = (0, line.split(":", 1)[1].strip())
                if line.startswith("#") or line.startswith("kernel "):
                    continue
                parts = line.rstrip().rsplit(None, 9)   # name | grid wg lds vgpr calls avg min max pct
                if len(parts) != 10:
                    continue
                name, calls, avg = parts[0].strip(), int(parts[5]), float(parts[6])
                if name not in out or calls > out[name][0]:
                    out[name] = (calls, avg)
        return rel, out
    return None, {}


def profile_lookup(stats, *needles):
    """The entry with the most calls whose kernel name contains every needle."""
    best = None
    for name, (calls, avg) in stats.items():
        if name.startswith("__"):
            continue
        if all(n in name for n in needles) and (best is None or calls > best[1]):
            best = (name, calls, avg)
    return best


def cell_launch_bytes(N, M, d, eb=4):
    """Compulsory bytes of ONE cell + message launch of the fused forward: the edge task reads h, c and writes
    h', c' and the next step's messages ([M, d] each), gathers the projected messages Zx[N, 4d] and the endpoints; the
    vertex task reads the aggregate x, h, c, writes h', c' ([N, d] each) and the next Zx.  Weights (112 + 258 KB per
    workgroup, L2 hits) are not counted."""
    edge = 5 * M * d * eb + N * 4 * d * 4 + 2 * M * 4
    vertex = 5 * N * d * eb + N * 4 * d * 4
    return {"edge_states_and_messages": 5 * M * d * eb, "Zx_read": N * 4 * d * 4, "endpoints": 2 * M * 4,
            "vertex_side": vertex, "total": edge + vertex}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="what one timed step is: a forward pass (default, the metric's definition) or a full "
                         "training step (forward + backward + gradient all-reduce + L2/clip/Adam)")
    ap.add_argument("--train-steps", type=int, default=3, help="extra (untimed-by-the-driver) training steps reported "
                                                               "under 'train' when --mode forward")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying "
                                                            "the captured HIP graph of the forward pass")
    ap.add_argument("--serve-batches", type=int, default=64,
                    help="fresh-batch ('serve') leg: this many batches of NEW instances go host instances -> native packer "
                         "-> BatchPrefetcher (worker thread, side-stream upload) -> DeviceBatch.copy_from -> replayed graph; "
                         "0 skips it.  Reported under 'serve', never in 'value'.")
    ap.add_argument("--serve-workers", type=int, default=1,
                    help="packer threads of the serve leg's BatchPrefetcher (measured: one keeps up at C2 / C4 and a second one's "
                         "uploads disturb the forward -- 1.51-1.55 vs 1.56-1.89 ms per C2 batch, 11.3-11.6 vs 16.5-18.2 ms at C4)")
    ap.add_argument("--plumbing", action="store_true",
                    help="no GPU: the launcher / rendezvous / collective / JSON-contract path only (Session(device='cpu'), gloo; "
                         "the timed step is the data-parallel bucket's pack -> all-reduce -> unpack).  What the CPU test suite "
                         "runs at --gpus 8; never a performance number.")
    ap.add_argument("--serve-prefetcher", action="store_true", help="serve leg: only the general BatchPrefetcher path "
                                                                    "(skip parallel.BatchStager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget (bounded sample)")
    args = ap.parse_args()

    import torch
    import tspgnn
    from tspgnn import _lib

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become one.  One rank per GPU through torch.distributed.run on
        # the loopback address (same command line the driver uses); the ranks' single JSON line is rank 0's.
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                  str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.plumbing:
        return plumbing_main(args, rank, world)
    # one rank per GPU; (ranks wrap around the visible devices only so that the N>1 path can be exercised with the
    # gloo backend on a single-GPU box: TSPGNN_DIST_BACKEND=gloo, see tests/test_gpu_bench_contract.py)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("TSPGNN_DIST_BACKEND", "nccl")      # "nccl" = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    sizes, d, T, storage = WORKLOADS[args.workload]
    bf16 = storage == "bf16"
    t_pack0 = time.perf_counter()
    batch = tspgnn.synthetic_batch(sizes, seed=1234 + rank)
    t_pack = time.perf_counter() - t_pack0
    EV, W, C, route_exists, n_vertices, n_edges = batch
    M, N = EV.shape
    model = tspgnn.build_network(d, float_dtype=torch.bfloat16 if bf16 else torch.float32)
    sess = tspgnn.Session(model, device=device)
    sess.run(tspgnn.global_variables_initializer(seed=0))
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    dev_batch = sess.prepare(feed)                                     # inputs resident in HBM
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.no_graph

    def make_train_fn():
        if not use_graph:
            return sess.train_step
        replay_t = sess.capture_train_step(dev_batch)  # two HIP graphs around the (eager, RCCL) gradient all-reduce
        sess.run(tspgnn.global_variables_initializer(seed=0))
        return lambda _b: replay_t()

    if args.mode == "train":
        step_fn = make_train_fn()
    elif use_graph:
        replay = sess.capture_forward(dev_batch)       # hipGraph of the whole T-step forward pass
        step_fn = lambda _b: replay()
        for _ in range(20):                            # untimed: graph upload + GPU clock ramp (~40 ms), so that a short
            replay()                                   # --steps/--warmup run measures the same steady state as a long one
        torch.cuda.synchronize()
    else:
        step_fn = sess.forward_device

    def timed(fn, n_warm, n_steps):
        o = None
        for _ in range(n_warm):
            o = fn(dev_batch)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            o = fn(dev_batch)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, o

    elapsed, out = timed(step_fn, args.warmup, args.steps)
    loss = float(out["stats"][0].item())
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")
    # the device-side guard words of the timed region: a one-launch loop whose wait expired raises here, and a
    # batch that left the f16x2 range would have been timed on an arithmetic the product then replaces -- say so
    guard_bits = sess.last_range_bits if sess.range_exceeded() else 0
    # ---- the fresh-batch path (">= 6x at 8 GPUs hinges on the host packer and launch overhead"): every
    # batch is NEW host instances -> tspgnn_host_pack_batch (native) -> upload on a side stream behind the previous
    # batch -> copied into the buffers of the captured graph -> replay.  Every rank packs its own shard concurrently, so at
    # N > 1 this is where host contention between the ranks' packers would show; the resident-batch `value` cannot see it.
    serve = None
    if args.mode == "forward" and use_graph and args.serve_batches > 0:
        try:
            rng = np.random.RandomState(99 + rank)
            uniq = sorted(set(int(n) for n in sizes))
            pool = {n: [tspgnn.random_instance(n, rng) for _ in range(min(len(sizes), 64) if len(uniq) > 1 else 3 * len(sizes))]
                    for n in uniq}

            def fresh_instances(nb):
                for i in range(nb):
                    yield [pool[int(n)][(i * 37 + j) % len(pool[int(n)])] for j, n in enumerate(sizes)]

            def pack(inst):   # runs on the prefetcher's worker threads (native packer: the GIL is released inside)
                return tspgnn.InstanceLoader.create_batch(inst, dev=0.02)

            def prefetched(nb):
                return tspgnn.BatchPrefetcher(sess, fresh_instances(nb), T, workers=args.serve_workers, pack=pack)

            # one shape for every batch (the benchmark's own): the staged path -- one native call into a pinned slot,
            # one upload, one device copy (parallel.BatchStager); ragged pools of other shapes keep the prefetcher
            stager = None
            if not args.serve_prefetcher:
                try:
                    stager = tspgnn.BatchStager(sess, next(fresh_instances(1)), T)
                    if (stager.batch.adj.loop_plan is None) != (dev_batch.adj.loop_plan is None):
                        stager = None
                except Exception:   # noqa: BLE001 -- falls back to the prefetcher, reported below
                    stager = None
            if stager is not None:
                replay_s = sess.capture_forward(stager.batch)
                for _ in stager.feed(fresh_instances(3)):
                    replay_s()
                barrier()
                t_s0 = time.perf_counter()
                keep = []
                for _ in stager.feed(fresh_instances(args.serve_batches)):
                    keep.append(replay_s()["predictions"].clone())
                barrier()
                dt_stage = time.perf_counter() - t_s0
                t_p0 = time.perf_counter()
                for inst in fresh_instances(4):
                    stager._stage(inst, 0)
                stage_ms = 1e3 * (time.perf_counter() - t_p0) / 4
                if world > 1:
                    tmax = torch.tensor([dt_stage, stage_ms], dtype=torch.float64, device=device)
                    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                    dt_stage, stage_ms = float(tmax[0].item()), float(tmax[1].item())
                staged = {"what": "fresh instances every batch: host instances -> ONE native staging call into a pinned slot "
                                  "(worker thread) -> one side-stream upload -> one device copy into the captured graph's "
                                  "buffer -> replay (parallel.BatchStager); max over ranks",
                          "batches": args.serve_batches, "ms_per_batch": round(1e3 * dt_stage / args.serve_batches, 4),
                          "value": round(world * args.serve_batches * T / dt_stage, 2), "unit": "mp-steps/s",
                          "host_stage_ms_per_batch_one_thread": round(stage_ms, 3),
                          "finite": bool(all(torch.isfinite(k).all().item() for k in keep)),
                          "range_guard_bits": sess.last_range_bits if sess.range_exceeded() else 0}   # (raises on a loop status)
            else:
                staged = None

            t_p0 = time.perf_counter()
            for inst in fresh_instances(4):
                pack(inst)
            pack_ms = 1e3 * (time.perf_counter() - t_p0) / 4
            for bb in prefetched(3):      # warm-up: allocator, worker threads
                dev_batch.copy_from(bb)
                replay()
            barrier()
            t_s0 = time.perf_counter()
            keep = []
            for bb in prefetched(args.serve_batches):
                dev_batch.copy_from(bb)
                keep.append(replay()["predictions"].clone())
            barrier()
            dt_serve = time.perf_counter() - t_s0
            if world > 1:
                tmax = torch.tensor([dt_serve, pack_ms], dtype=torch.float64, device=device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt_serve, pack_ms = float(tmax[0].item()), float(tmax[1].item())
            serve = {"what": "fresh instances every batch: host instances -> native packer (%d worker threads) -> side-stream "
                             "upload -> DeviceBatch.copy_from -> replayed forward graph; one prefetcher per rank; "
                             "max over ranks" % args.serve_workers, "batches": args.serve_batches,
                     "ms_per_batch": round(1e3 * dt_serve / args.serve_batches, 4),
                     "value": round(world * args.serve_batches * T / dt_serve, 2), "unit": "mp-steps/s",
                     "host_pack_ms_per_batch_one_thread": round(pack_ms, 3), "n_gpus": world,
                     "finite": bool(all(torch.isfinite(k).all().item() for k in keep)),
                     "range_guard_bits": sess.last_range_bits if sess.range_exceeded() else 0}
            if staged is not None:   # the staged path is the serving path of a fixed-shape workload: it leads the block
                staged["n_gpus"] = world
                staged["vs_resident"] = round(staged["ms_per_batch"] / (1e3 * elapsed / args.steps), 4)
                staged["prefetcher"] = serve
                serve = staged
            dev_batch.copy_from(sess.prepare(feed))   # the benchmark batch back in the graph's buffers
            replay()
            torch.cuda.synchronize()
        except Exception as exc:   # noqa: BLE001 -- reported, not swallowed
            serve = {"error": "%s: %s" % (type(exc).__name__, exc)}

    train = None
    # N > 1: the line's `value` is the TRAINING step's whole-job rate -- the only step of this path with
    # a collective (one RCCL all-reduce of the gradient bucket per step) -- timed over exactly --steps steps
    # after --warmup, barrier-bracketed, max over ranks; the collective-free forward moves to `forward_value`.
    n_train, w_train = (args.steps, max(1, args.warmup)) if world > 1 else (args.train_steps, 1)
    if args.mode == "forward" and n_train > 0:
        # the training step (backward + RCCL all-reduce of the 462 KB gradient bucket + fused optimiser), reported
        # next to the forward number; a failure here (e.g. the collective) must not lose the forward measurement
        try:
            dt_train, tout = timed(make_train_fn(), w_train, n_train)
            allreduce_us = None
            if world > 1:      # the step's one collective alone: the bucket, back to back, HIP events on this stream
                sess.store.zero_grad()
                for _ in range(5):
                    dist.all_reduce(sess.store.bucket)
                barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    dist.all_reduce(sess.store.bucket)
                e1.record()
                torch.cuda.synchronize()
                tm = torch.tensor([e0.elapsed_time(e1) * 1e3 / 20], dtype=torch.float64, device=device)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                allreduce_us = round(float(tm.item()), 2)
            train = {"ms_per_batch": round(1e3 * dt_train / n_train, 3),
                     "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                     "backend": dist.get_backend() if world > 1 else None,
                     "allreduce_us": allreduce_us,
                     "allreduce_bytes": 4 * (sess.store.theta.numel() + sess.store.BUCKET_TAIL) if world > 1 else None,
                     "ms_per_step": round(1e3 * dt_train / n_train, 3),
                     "value": round(world * n_train * T / dt_train, 2), "unit": "mp-steps/s",
                     "mp_steps_per_s": round(world * n_train * T / dt_train, 2),
                     "n_gpus": world, "scaling": "weak", "global_batch": len(sizes) * world,
                     "collective": "one all-reduce (RCCL) of the %d-byte bucket [gradient | batch size, statistics, "
                                   "range flag] per step" % (4 * (sess.store.theta.numel() + sess.store.BUCKET_TAIL))
                                   if world > 1 else None,
                     "steps": n_train, "loss": float(tout["stats"][0].item()),
                     "global_norm": float(tout["global_norm"].item()),
                     "what": "forward + backward through T steps + gradient all-reduce (world %d) + L2/clip/Adam: the step "
                             "north_star scales over GPUs (whole-job mp-steps/s = n_gpus * steps * T / max-over-ranks time; "
                             "the forward-only `value` has no collective)" % world}
        except Exception as exc:   # noqa: BLE001 -- reported, not swallowed
            train = {"error": "%s: %s" % (type(exc).__name__, exc)}
        sess.run(tspgnn.global_variables_initializer(seed=0))   # restore the benchmark weights

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        mp_steps_per_s = world * args.steps * T / elapsed
        gather_b, rowsum_b = spmm_bytes(N, M, d, eb=2 if bf16 else 4)

        # ---- the same forward with the other GEMM arithmetics, next to the headline: their time, and how far their
        # predictions are from the headline's on this batch (all three are fp32-class; the parity tests hold them to
        # the same 1e-5 bar against the float64 oracle)
        gemm = None
        gnn = model["gnn"]
        GEMM_DOC = {
            "f16x2": "f16x2: every fp32 operand split into two fp16 pieces (x = hi + lo to 2^-24 relative), three "
                     "v_mfma_f32_16x16x32_f16 terms per product accumulated in fp32 (dropped term <= 2^-24 relative)",
            "bf16x3": "bf16x3: every fp32 operand split exactly into three bf16 pieces, six v_mfma_f32_16x16x32_bf16 "
                      "terms per product accumulated in fp32 (dropped terms <= 2^-24 relative)",
            "f32": "f32: v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate)",
        }
        if args.mode == "forward" and not bf16:
            headline = gnn.gemm
            pred0 = sess.forward_device(dev_batch)["predictions"].clone()
            alts = {}
            try:
                for alt in ("f16x2", "bf16x3", "f32"):
                    if alt == headline:
                        continue
                    gnn.gemm = alt
                    out_alt = sess.forward_device(dev_batch)
                    diff = float(((out_alt["predictions"] - pred0).abs() / pred0.abs().clamp_min(1e-30)).max())
                    fn_alt = sess.capture_forward(dev_batch) if use_graph else (lambda: sess.forward_device(dev_batch))
                    n_alt = max(3, min(10, args.steps))
                    fn_alt()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n_alt):
                        fn_alt()
                    torch.cuda.synchronize()
                    ms_alt = 1e3 * (time.perf_counter() - t0) / n_alt
                    alts[alt] = {"what": GEMM_DOC[alt], "ms_per_step": round(ms_alt, 4),
                                 "mp_steps_per_s": round(T / (ms_alt * 1e-3), 2),
                                 "max_rel_diff_predictions_vs_headline": diff}
            finally:
                gnn.gemm = headline
            gemm = {"headline": GEMM_DOC[headline], "alternatives": alts}

        # ---- per-kernel durations, live, HIP events on the launch stream (one instrumented pass)
        _lib.TIMELINE = []
        sess.forward_device(dev_batch)
        torch.cuda.synchronize()
        per = {}
        for name, e0, e1 in _lib.TIMELINE:
            per.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)   # us
        _lib.TIMELINE = None
        kernels_us = {k: {"n": len(v), "avg_us": float(np.mean(v)), "total_us": float(np.sum(v))}
                      for k, v in per.items()}

        # ---- SpMM-only (SURVEY.md §8d M1 i): the two aggregation kernels back to back, >= 200 warm iterations
        adj = dev_batch.adj
        sdt = torch.bfloat16 if bf16 else torch.float32
        sfx = "bf16" if bf16 else "f32"
        X = torch.randn((N, d), device=device).to(sdt)
        Z = torch.randn((M, d), device=device).to(sdt)
        Y = torch.empty((M, d), device=device, dtype=sdt)
        Vout = torch.empty((N, d), device=device, dtype=sdt)
        rowptr, eid, _ = adj.csr_t
        st = _lib.current_stream()

        def gather():
            _lib.call("tspgnn_gather2_sum_" + sfx, _lib.ptr(adj.uv), _lib.ptr(X), _lib.ptr(Y), M, N, d, st)

        def rowsum():
            _lib.call("tspgnn_csr_rowsum_" + sfx, _lib.ptr(rowptr), _lib.ptr(eid), _lib.ptr(Z), _lib.ptr(Vout), N, M, d, st)

        def time_loop(fns, iters=300):
            for _ in range(20):
                for f in fns:
                    f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                for f in fns:
                    f()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters   # us per iteration

        def pair():   # both directions of a step's aggregation in one launch (tspgnn_spmm_pair_f32)
            _lib.call("tspgnn_spmm_pair_f32", _lib.ptr(adj.uv), _lib.ptr(X), _lib.ptr(Y), _lib.ptr(rowptr), _lib.ptr(eid),
                      _lib.ptr(Z), _lib.ptr(Vout), M, N, d, st)

        # ---- the V<-E row-sum WHERE IT RUNS: the replayed forward with and without its row-sum launches (the skipped
        # aggregates are whatever the buffers held: timing only), alternating, difference / T.  Includes the launch
        # boundary the kernel adds to the pass -- what the pass actually pays for the aggregation.
        rowsum_in_fwd_us = None
        if use_graph and args.mode == "forward":
            from tspgnn.graphnn import DeviceAdjacency
            full = sess.capture_forward(dev_batch)
            orig_matmul = DeviceAdjacency.matmul
            skipped = [0]

            def no_rowsum(self_, y, transpose=False, out=None):
                if transpose and out is not None:
                    skipped[0] += 1
                    return out
                return orig_matmul(self_, y, transpose=transpose, out=out)
            DeviceAdjacency.matmul = no_rowsum
            try:
                without = sess.capture_forward(dev_batch)
            finally:
                DeviceAdjacency.matmul = orig_matmul
            if skipped[0]:
                def t_replays(fn, n=20):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / n
                for fn in (full, without):
                    t_replays(fn, 5)
                d_full, d_wo = [], []
                for _ in range(5):
                    d_full.append(t_replays(full))
                    d_wo.append(t_replays(without))
                rowsum_in_fwd_us = (float(np.median(d_full)) - float(np.median(d_wo))) * 1e6 / T
            sess.forward_device(dev_batch)   # (leave valid aggregates / outputs behind)

        t_gather, t_rowsum, t_two = time_loop([gather]), time_loop([rowsum]), time_loop([gather, rowsum])
        t_pair = t_two if bf16 else time_loop([pair])   # (the bf16-storage kernels have no one-launch pair)
        pair_gbs = (gather_b + rowsum_b) / (t_pair * 1e-6) / 1e9
        # HBM-side bytes per launch (pair) from the committed PMC passes (rocprofv3 --pmc cannot run inside bench.py):
        # profiles/r02_spmm_pmc_traffic.json, keyed by workload
        traffic, traffic_src, prof_fwd = None, None, {}
        tname = None
        for rnd in PROFILE_ROUNDS + ("r02",):     # newest round that measured THIS workload
            cand = "%s_spmm_pmc_traffic.json" % rnd
            if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                with open(os.path.join(ROOT, "profiles", cand)) as f:
                    ent = json.load(f).get(args.workload)
                if ent:
                    tname = cand
                    traffic, traffic_src = ent.get("pair_traffic_bytes"), "profiles/%s: " % tname + ent.get("how", "")
                    prof_fwd = ent.get("in_forward", {})
                    break
        # the SAME kernels where they execute: inside the forward pass (the instrumented eager pass above -- HIP events on
        # the launch stream around every launch, operands produced by the preceding launch, not replayed from cache)
        in_forward = {}
        for kname, bytes_ in (("tspgnn_csr_rowsum_" + sfx, rowsum_b), ("tspgnn_gather2_sum_" + sfx, gather_b)):
            if kname in kernels_us:
                us = kernels_us[kname]["avg_us"]
                in_forward[kname] = {"avg_us": round(us, 2), "launches": kernels_us[kname]["n"],
                                     "GBs": round(bytes_ / us / 1e3, 1), "frac": round(bytes_ / us / 1e3 / HBM_PEAK_GBS, 4),
                                     "how": "HIP events around each launch in the eager forward (includes ~2 us of "
                                            "launch gap on a ~7 us kernel)"}
                # the same launches as rocprofv3's kernel trace saw them (kernel begin -> end, committed profile)
                tag = "csr_rowsum" if "rowsum" in kname else "gather2_sum"
                for grid_key, pf in sorted(prof_fwd.get(tag, {}).items()):
                    pus = pf.get("avg_us_under_profiler")
                    if pus:
                        in_forward[kname]["rocprof"] = {
                            "avg_us": pus, "dispatches": pf.get("dispatches"), "GBs": round(bytes_ / pus / 1e3, 1),
                            "frac": round(bytes_ / pus / 1e3 / HBM_PEAK_GBS, 4), "traffic_bytes": pf.get("traffic_bytes"),
                            "source": "profiles/%s in_forward (rocprofv3 --kernel-trace --pmc, tools/forward_only.py)" % tname}
        rk = "tspgnn_csr_rowsum_" + sfx
        micro = {
            "kernel": ("MICRO-LOOP, not the forward: tspgnn_gather2_sum_bf16 + tspgnn_csr_rowsum_bf16 back to back" if bf16
                       else "MICRO-LOOP, not the forward: tspgnn_spmm_pair_f32 (E<-V gather + V<-E CSR row-sum of one step "
                            "in one launch) back to back on a cache-resident operand"),
            "achieved": round(pair_gbs, 1), "frac": round(pair_gbs / HBM_PEAK_GBS, 4), "unit": "GB/s", "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_us": {"gather2_sum": round(t_gather, 2), "csr_rowsum": round(t_rowsum, 2), "pair": round(t_pair, 2),
                       "two_launches": round(t_two, 2)},
            "per_kernel_GBs": {"gather2_sum": round(gather_b / t_gather / 1e3, 1),
                               "csr_rowsum": round(rowsum_b / t_rowsum / 1e3, 1)},
            "spmm_steps_per_s": round(1e6 / t_pair, 1), "incidences_per_s": round(4 * M * 1e6 / t_pair, 1),
            "how": "HIP events on the launch stream, 300 launches of the same operands",
        }
        # ---- the headline roofline: the V<-E row-sum launch's OWN duration where the product runs it.
        #   avg_us / frac: the kernel's average duration in the replayed forward as rocprofv3's kernel trace recorded it
        #     (begin -> end of the dispatch), read from the committed summary of THIS workload -- recomputable from that one
        #     named file: frac = algorithmic bytes / avg_us / 8 TB/s;
        #   live_events_us: HIP events on the launch stream around each launch of the (eager) forward, measured now on this
        #     box -- includes ~2 us of launch gap on a ~7 us kernel, so it reads higher than the kernel's own duration;
        #   marginal_cost_us: (replayed forward - replayed forward without its row-sum launches) / T -- what the pass PAYS
        #     for the aggregation (a difference of two replays, NOT a kernel time: the launch that follows starts cold).
        prof_path, prof_stats = profile_kernel_stats(args.workload)
        # the committed summary counts as THIS build's only if it carries the fingerprint of the kernel sources in the tree
        prof_sha = prof_stats.get("__csrc_sha16__", (0, None))[1]
        prof_fresh = prof_sha is not None and prof_sha == csrc_fingerprint()
        rs_prof_any = profile_lookup(prof_stats, "csr_rowsum_bf16" if bf16 else "csr_rowsum_kernel")
        rs_prof = rs_prof_any if prof_fresh else None
        live_us = kernels_us[rk]["avg_us"] if rk in kernels_us else None
        if rs_prof is None and rowsum_in_fwd_us and rowsum_in_fwd_us > 0:
            # no summary of this build: the live figure -- what the replayed pass pays per row-sum launch, measured now
            where_us = rowsum_in_fwd_us
            where_how = ("LIVE: (replayed forward - replayed forward without its %d row-sum launches) / T, medians of 5 x 20 "
                         "alternating HIP-graph replays on this box (the committed rocprofv3 summary %s was taken on other "
                         "kernel sources and is reported under `rocprof_stale` only)" % (T, prof_path))
            where_src = "live"
        elif rs_prof:
            where_us = rs_prof[2]
            where_how = ("rocprofv3 --kernel-trace --stats of the replayed forward (tools/forward_graph.py %s 20): average "
                         "duration of %d dispatches of `%s`" % (args.workload, rs_prof[1], rs_prof[0]))
            where_src = prof_path
        elif live_us is not None:
            where_us, where_how, where_src = live_us, "HIP events around each launch in the eager forward (no committed " \
                "rocprofv3 summary for this workload)", "live"
        else:
            where_us, where_how, where_src = t_rowsum, "micro-loop (no row-sum launch in this forward)", "live"
        rs_traffic = None
        for _gk, pf in sorted(prof_fwd.get("csr_rowsum", {}).items()):
            rs_traffic = pf.get("traffic_bytes", rs_traffic)
        roofline = {
            "kernel": "%s IN THE TIMED FORWARD: the V<-E aggregation launch of a message-passing step, reading the messages "
                      "the cell launch has just written (the E<-V direction has no launch of its own: Zx[u] + Zx[v] is the "
                      "edge cell's operand gather inside the fused launch, see roofline_cell)" % rk,
            "bound": "hbm", "achieved": round(rowsum_b / where_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(rowsum_b / where_us / 1e3 / HBM_PEAK_GBS, 4), "avg_us": round(where_us, 2), "how": where_how,
            "source": where_src,
            "profile_matches_build": prof_fresh,
            "rocprof_stale": None if prof_fresh or not rs_prof_any else
                             {"avg_us": rs_prof_any[2], "source": prof_path, "csrc_sha16": prof_sha, "tree_sha16": csrc_fingerprint()},
            "live_events_us": round(live_us, 2) if live_us is not None else None,
            "live_events_frac": round(rowsum_b / live_us / 1e3 / HBM_PEAK_GBS, 4) if live_us else None,
            "marginal_cost_us": round(rowsum_in_fwd_us, 2) if rowsum_in_fwd_us and rowsum_in_fwd_us > 0 else None,
            "marginal_cost_how": "replayed HIP graph of the timed forward with and without its %d row-sum launches, medians of "
                                 "5 x 20 alternating replays, difference / T; a marginal cost of the pass, not a kernel "
                                 "duration" % T,
            "traffic": rs_traffic,
            "traffic_over_algorithmic": round(rs_traffic / rowsum_b, 3) if rs_traffic else None,
            "traffic_source": ("profiles/%s in_forward.csr_rowsum (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, "
                               "separate passes, tools/forward_only.py; FETCH_SIZE doubled per the gfx950 correction)" % tname)
                              if rs_traffic else None,
            "algorithmic_bytes_per_launch": {"gather2_sum": gather_b, "csr_rowsum": rowsum_b},
            "in_forward": in_forward,
            "micro_loop": micro,
            "note": "north_star's target kernel where the product executes it.  `micro_loop` is the same kernel family timed "
                    "back to back on one operand (SURVEY 8d M1 i) and is labelled as such.",
        }
        # ---- the launch that decides the headline: cell + next step's messages (79 % of the pass).  HBM view: compulsory
        # bytes over the launch's own duration (same committed rocprofv3 file), PMC bytes next to it.
        roofline_cell = None
        cell_prof = profile_lookup(prof_stats, "lnlstm_mlp_fwd_h2_kernel") if not bf16 else \
            profile_lookup(prof_stats, "lnlstm_fwd_bf16_kernel")
        cname_live = "tspgnn_lnlstm_mlp_fwd_multi_h2" if not bf16 else "tspgnn_lnlstm_fwd_multi_bf16"
        if not prof_fresh and cname_live in kernels_us:
            # (stale summary: the launch's duration by HIP events in the eager forward, measured now -- reads ~1-2 us high)
            cell_prof = (cname_live + " (live HIP events; committed summary is of other sources)", kernels_us[cname_live]["n"],
                         kernels_us[cname_live]["avg_us"])
        if cell_prof:
            cb = cell_launch_bytes(N, M, d, eb=2 if bf16 else 4)
            if bf16:   # bf16 storage: h and messages bf16, c fp32; the message MLP is a launch of its own
                cb = {"edge_states": M * d * (2 + 4 + 2 + 4), "Zx_read": N * 4 * d * 2, "endpoints": 2 * M * 4,
                      "vertex_side": N * d * (2 + 2 + 4 + 2 + 4)}
                cb["total"] = sum(cb.values())
            cell_traffic = None
            for key in ("cell_launch", "cell_launch_bf16"):
                for _gk, pf in sorted(prof_fwd.get(key, {}).items()):
                    cell_traffic = pf.get("traffic_bytes", cell_traffic)
            cname = "tspgnn_lnlstm_mlp_fwd_multi_h2" if not bf16 else "tspgnn_lnlstm_fwd_multi_bf16"
            roofline_cell = {
                "kernel": cell_prof[0], "bound": "hbm first, then VALU + MFMA issue (DESIGN 4)",
                "compulsory_bytes_per_launch": cb, "avg_us": cell_prof[2], "dispatches": cell_prof[1],
                "achieved": round(cb["total"] / cell_prof[2] / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(cb["total"] / cell_prof[2] / 1e3 / HBM_PEAK_GBS, 4),
                "traffic": cell_traffic,
                "traffic_over_compulsory": round(cell_traffic / cb["total"], 3) if cell_traffic else None,
                "source": prof_path if prof_fresh else "live", "traffic_source": ("profiles/%s in_forward" % tname) if cell_traffic else None,
                "live_events_us": round(kernels_us[cname]["avg_us"], 2) if cname in kernels_us else None,
                "share_of_forward": round(cell_prof[1] * cell_prof[2] /
                                          max(1e-9, sum(c * a for k_, (c, a) in prof_stats.items() if not k_.startswith("__"))), 4)
                                    if prof_fresh else None,
            }
        dense_names = ("tspgnn_mlp_fwd_f32", "tspgnn_mlp_fwd_multi_f32", "tspgnn_lnlstm_fwd_f32",
                       "tspgnn_lnlstm_fwd_multi_f32", "tspgnn_lnlstm_gather_fwd_f32", "tspgnn_linear_f32",
                       "tspgnn_mlp_fwd_multi_x3", "tspgnn_lnlstm_fwd_multi_x3", "tspgnn_lnlstm_mlp_fwd_multi_x3",
                       "tspgnn_mlp_fwd_multi_h2", "tspgnn_lnlstm_fwd_multi_h2", "tspgnn_lnlstm_mlp_fwd_multi_h2",
                       "tspgnn_mlp_fwd_multi_bf16", "tspgnn_lnlstm_fwd_multi_bf16")
        dense_us = sum(v["total_us"] for k, v in kernels_us.items() if k in dense_names)
        h2 = any(k.endswith("_h2") and "pack" not in k for k in kernels_us)
        x3 = h2 or any(k.endswith("_x3") and "pack" not in k for k in kernels_us)
        # split operands: every fp32 product costs three fp16 (f16x2) or six bf16 (bf16x3) MFMA terms -> the matrix-pipe
        # ceiling in fp32-equivalent flops is the 16-bit dense peak over 3 or 6
        dense_peak = BF16_MFMA_PEAK_TF / (3.0 if h2 else 6.0) if x3 else (BF16_MFMA_PEAK_TF if bf16 else FP32_MFMA_PEAK_TF)
        # E_vote's 3 hidden layers also run through mlp_fwd: count their flops too
        dense_flops = T * dense_flops_per_step(N, M, d, folded=True) + M * 3 * 2 * d * d
        roofline_dense = {
            "kernel": ("lnlstm_mlp_fwd_multi_h2 (+ mlp_fwd_multi_h2): v_mfma_f32_16x16x32_f16 on 2-way fp16 splits, 3 terms "
                       "per fp32 product; peak = fp16 dense peak (2.5 PFLOP/s) / 3") if h2 else
                      ("lnlstm_mlp_fwd_multi_x3 (+ mlp_fwd_multi_x3, fp32 vote MLP): v_mfma_f32_16x16x32_bf16 on exact "
                       "3-way bf16 splits, 6 terms per fp32 product; peak = bf16 dense peak / 6") if x3 else
                      "mlp_fwd_multi_bf16 + lnlstm_fwd_multi_bf16 (v_mfma_f32_16x16x32_bf16 on bf16-rounded operands)" if bf16 else
                      "mlp_fwd_multi + lnlstm_fwd_multi + linear (fp32 MFMA v_mfma_f32_16x16x4_f32)",
            "bound": "mfma", "achieved": round(dense_flops / (dense_us * 1e-6) / 1e12, 2) if dense_us else None,
            "peak": round(dense_peak, 1), "unit": "TFLOP/s (fp32-equivalent)" if x3 else "TFLOP/s",
            "frac": round(dense_flops / (dense_us * 1e-6) / 1e12 / dense_peak, 4) if dense_us else None,
            "executed_gflop_per_mp_step": round(dense_flops_per_step(N, M, d, True) / 1e9, 3),
            "reference_gflop_per_mp_step": round(dense_flops_per_step(N, M, d, False) / 1e9, 3),
            "note": "executed flops (the adjacency product is folded through the edge cell's GEMM); the reference "
                    "graph does 10.3 GFLOP of dense work per step at C2",
        }

        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1 and float(M) * N * 4 < 16e9:   # rank 0, N=1 only; dense EV must fit
            cpu_baseline = run_cpu_baseline(d, batch, T, args.cpu_seconds, M)

        # which step the headline fields describe: the forward pass at N = 1 (BASELINE's metric as the reference runs it),
        # the training step -- with its all-reduce -- at N > 1
        train_ok = world > 1 and isinstance(train, dict) and "error" not in train
        head_value = train["value"] if train_ok else round(mp_steps_per_s, 2)
        head_ms = train["ms_per_step"] if train_ok else round(ms_per_step, 4)
        result = {
            "metric": "message-passing steps/sec (edges aggregated/sec) at n=40, batch=128, T=32"
                      + ("" if args.workload == "c2" else " [measured on workload %s, not the metric's configuration]" % args.workload),
            "value": head_value, "unit": "mp-steps/s",
            "value_is": ("training step: forward + backward + ONE all-reduce (%s) of the gradient bucket over %d ranks + "
                         "L2 / clip / Adam" % ((train or {}).get("backend"), world)) if train_ok
                        else ("forward pass" if world == 1 else "forward pass (NO collective): the training leg failed, see `train`"),
            "forward_value": round(mp_steps_per_s, 2), "forward_ms_per_step": round(ms_per_step, 4),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 storage, f32 accumulate" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": "%s: %d complete Euclidean graphs n=%s, d=%d, T=%d, %s, forward pass "
                                   "(E_init -> T x {msg MLPs, SpMM pair, LN-LSTMs} -> vote -> loss)%s"
                                   % (args.workload, len(sizes), ("%d" % sizes[0]) if min(sizes) == max(sizes)
                                      else "%d..%d" % (min(sizes), max(sizes)), d, T,
                                      "bf16 embeddings / fp32 accumulate" if bf16 else "fp32",
                                      "" if args.mode == "forward" else " + backward + all-reduce + Adam"),
                       "per_gpu_batch": len(sizes), "global_batch": len(sizes) * world, "N": N, "M": M,
                       "parallelism": "shard-by-instance x%d, no data-path collective" % world},
            "edges_per_s": round(mp_steps_per_s * M, 1),
            "incidences_per_s": round(mp_steps_per_s * 4 * M, 1),
            "roofline": roofline,
            "roofline_cell": roofline_cell,
            "roofline_dense": roofline_dense,
            "cpu_baseline": cpu_baseline,
            "gemm": gemm,
            "mode": args.mode, "hip_graph": bool(use_graph),
            # N > 1: the forward-only `value` has no collective and scales by construction; the curve north_star asks for
            # is the TRAINING step's (one RCCL all-reduce per step) -- lifted to the top level so that it cannot be missed
            "rccl_ranks": (train or {}).get("rccl_ranks") if world > 1 else None,
            "allreduce_us": (train or {}).get("allreduce_us") if world > 1 else None,
            "train_value": (train or {}).get("value"),
            "train_ms_per_step": (train or {}).get("ms_per_step"),
            "scaling_note": ("`value` = whole-job mp-steps/s of the TRAINING step (%d ranks x %d steps x T / max-over-ranks "
                             "time), whose gradient bucket crosses all ranks once per step; `forward_value` = forward passes of "
                             "%d independent shards (no data-path collective: weak scaling by construction)"
                             % (world, args.steps, world)) if world > 1 else None,
            "range_guard_bits": guard_bits,
            "train": train,
            "serve": serve,
            "kernels_us": {k: {"n": v["n"], "avg_us": round(v["avg_us"], 2)} for k, v in kernels_us.items()},
            "host_pack_s": round(t_pack, 4),
            "loss": loss,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def plumbing_main(args, rank, world):
    """`bench.py --plumbing --gpus N`: everything of the N-rank benchmark that is not a kernel -- launcher, rendezvous on
    127.0.0.1 (gloo), per-rank shards of unequal size, the data-parallel bucket (pack -> ONE all-reduce -> unpack, the
    same Session code the GPU path runs), barrier-bracketed timing with the max over ranks, rank 0 printing ONE JSON line
    with the driver's keys.  Session(device='cpu') launches no kernel; the numbers time the collective on the host and are
    labelled as such."""
    import torch
    import torch.distributed as dist
    import tspgnn
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo")
    d, T = 32, 2
    sizes = [5 + (rank + i) % 4 for i in range(2 + rank % 3)]            # unequal shards: B_r / B weights matter
    batch = tspgnn.synthetic_batch(sizes, seed=1234 + rank)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model, device="cpu")
    sess.run(tspgnn.global_variables_initializer(seed=0))
    store = model.store
    store.zero_grad()
    store.grad.fill_(float(rank + 1))
    stats = torch.tensor([0.5 + rank, 1.0, 1.0, 0.0, float(len(sizes)) - 1.0, 0.0])

    def step():
        store.grad.fill_(float(rank + 1))
        st = stats.clone()
        sess.allreduce_grads(len(sizes), st)
        return st

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    barrier()
    dt = time.perf_counter() - t0
    total_b = float(len(sizes))
    if world > 1:
        tm = torch.tensor([dt, total_b], dtype=torch.float64)
        mx = tm.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tm, op=dist.ReduceOp.SUM)
        dt, total_b = float(mx[0]), float(tm[1])
    # the reduced gradient of a constant-(r+1) gradient weighted by B_r / B: checked against the closed form on every rank
    want = 1.0
    if world > 1:
        per = torch.zeros(world, 2, dtype=torch.float64)
        per[rank, 0], per[rank, 1] = float(len(sizes)), float(len(sizes) * (rank + 1))
        dist.all_reduce(per)
        want = float(per[:, 1].sum() / per[:, 0].sum())
    ok = bool(abs(float(store.grad[0]) - want) < 1e-5 * max(1.0, want))
    result = None
    if rank == 0:
        result = {
            "metric": "message-passing steps/sec (edges aggregated/sec) at n=40, batch=128, T=32 [PLUMBING RUN: no GPU, "
                      "collective path only]",
            "value": round(world * args.steps * T / dt, 2), "unit": "mp-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "plumbing: %d ranks, shards of 2-4 tiny graphs, d=%d, no kernels" % (world, d),
                       "global_batch": int(total_b), "parallelism": "shard-by-instance x%d, one all-reduce of the bucket" % world},
            "value_is": "the data-parallel bucket's ONE all-reduce per step (the collective the N > 1 headline -- the training "
                        "step -- contains); no kernels", "forward_value": None,
            "plumbing": True, "roofline": None, "cpu_baseline": None,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
            "allreduce_us": round(1e6 * dt / args.steps, 1), "allreduce_bytes": 4 * store.bucket.numel(),
            "reduced_gradient_ok": ok, "reduced_loss": float(st[0]),
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("plumbing: the reduced gradient is %r, expected %r" % (float(store.grad[0]), want))
    return result


def physical_cores():
    """Physical cores of this host (the CPU baseline runs on physical cores, not SMT threads): distinct
    (physical id, core id) pairs of /proc/cpuinfo -- psutil's count came back as the logical 128 on the 64-core EPYC of
    the GPU boxes --, else psutil, else os.cpu_count()."""
    try:
        pairs, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    pairs.add((phys, line.split(":", 1)[1].strip()))
        if pairs:
            physical_cores.sockets = len({p for p, _ in pairs})
            return min(len(pairs), os.cpu_count() or len(pairs))
    except OSError:
        pass
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        return os.cpu_count()


def run_cpu_baseline(d, batch, T, budget_s, M):
    """The oracle's op-for-op dense fp32 restatement of the TF CPU graph (dense EV matmul both
    directions), timed on this host's cores on a BOUNDED sample: as many message-passing steps of
    the full batch as fit the budget.  Checker code used as a reported baseline."""
    import torch
    from oracle import torch_oracle as TO
    cores = physical_cores()
    EV, W, C, route_exists, n_vertices, n_edges = batch
    ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
    t1, threads = TO.time_dense_forward(d, ob, 1, cores, warm=0, iters=1)       # also warms the allocator
    n = int(max(1, min(T, budget_s / max(t1, 1e-3))))
    tn, threads = TO.time_dense_forward(d, ob, n, cores, warm=0, iters=1)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    steps_per_s = n / tn
    return {"value": round(steps_per_s, 4), "unit": "mp-steps/s", "cores": int(threads), "kind": "port",
            "sockets": getattr(physical_cores, "sockets", None),   # (the GPU boxes: 2 x 64-core EPYC 9575F = 128 physical cores)
            "sample": "%d of %d message-passing steps of the full batch, dense EV[%d,%d] fp32 torch-CPU restatement "
                      "of the TF graph (oracle/torch_oracle.py), %.1f s" % (n, T, M, int(np.sum(n_vertices)), tn),
            "edges_per_s": round(steps_per_s * M, 1), "cpu": cpu_model}


if __name__ == "__main__":
    main()
