"""Host-side mirror of the reference interface: constructor checks, variable layout, plumbing
that must work (or fail loudly) without a GPU."""
import numpy as np
import pytest
import torch

import tspgnn
from oracle import params as P
from tspgnn import variables as V


def test_build_network_keys_and_variables():
    m = tspgnn.build_network(64)
    for k in ("gnn", "route_exists", "n_vertices", "n_edges", "EV", "W", "C", "time_steps", "last_states",
              "predictions", "TP", "FP", "TN", "FN", "acc", "loss", "train_step"):      # model.py:97-167
        assert k in m
    assert m.store.names() == P.param_names(64)
    assert m.store.size == 115529
    m128 = tspgnn.build_network(128)
    assert m128.store.size == 457617


def test_store_layout_is_pack_contiguous_and_aligned():
    m = tspgnn.build_network(32)
    st = m.store.finalize("cpu")
    st.initialize(seed=1)
    for name in st.names():
        assert st.offset(name) % 4 == 0
    mlp = m["gnn"]._msg_MLPs["E_msg_V"]
    wb = mlp.wb()
    assert wb.numel() == 4 * (32 * 32 + 32)
    assert torch.equal(wb[:32 * 32].view(32, 32), st.view("TSP/E_msg_V_MLP_layer_1/kernel"))
    cell = m["gnn"]._RNN_cells["E"]
    assert cell.ln().numel() == 10 * 32 and torch.equal(cell.ln()[:32], torch.ones(32))
    # initialisers: zero biases for E_init / E_vote, xavier (non-zero) biases for the message MLPs
    assert torch.count_nonzero(st.view("E_vote_MLP_layer_1/bias")) == 0
    assert torch.count_nonzero(st.view("TSP/V_msg_E_MLP_layer_1/bias")) > 0
    lim = np.sqrt(6.0 / (64 + 128))
    assert st.view("TSP/V_cell/layer_norm_basic_lstm_cell/kernel").abs().max() <= lim
    sd = st.state_dict()
    st.load({k: v * 2 for k, v in sd.items()})
    assert np.allclose(st.state_dict()["V_init"], 2 * sd["V_init"])


def test_check_model_errors_match_reference():
    store = V.VariableStore()
    with pytest.raises(Warning):        # graphnn.py:76
        tspgnn.GraphNN({"V": 64, "U": 64}, {}, {}, {"V": []}, store=store)
    with pytest.raises(Exception, match="Updating variable"):   # graphnn.py:82
        tspgnn.GraphNN({"V": 64}, {}, {}, {"V": [], "X": []}, store=V.VariableStore())
    with pytest.raises(Exception, match="Matrix M definition depends on undeclared"):   # graphnn.py:88
        tspgnn.GraphNN({"V": 64}, {"M": ("Q", "V")}, {}, {"V": []}, store=V.VariableStore())
    with pytest.raises(Exception, match="maps to undeclared"):   # graphnn.py:100
        tspgnn.GraphNN({"V": 64}, {}, {"c": ("V", "Z")}, {"V": []}, store=V.VariableStore())


def test_check_run_errors():
    m = tspgnn.build_network(32)
    m.store.finalize("cpu")
    gnn = m["gnn"]
    ev = tspgnn.SparseEV(np.array([[0, 1], [1, 2], [0, 2]]), 3)
    V0, E0 = torch.zeros(3, 32), torch.zeros(3, 32)
    with pytest.raises(ValueError, match="dimensionality 32"):
        gnn({"EV": ev}, {"V": torch.zeros(3, 16), "E": E0}, 1)
    with pytest.raises(ValueError, match="same number of nodes"):
        gnn({"EV": ev}, {"V": torch.zeros(4, 32), "E": E0}, 1)
    with pytest.raises(ValueError, match="same shape"):
        gnn({"EV": ev}, {"V": V0, "E": E0}, 1, LSTM_initial_states={"V": torch.zeros(2, 32)})


def test_unsupported_shapes_raise_not_fallback():
    with pytest.raises(NotImplementedError):
        tspgnn.Mlp([48, 48], output_size=48, activations=["relu", "relu"], name="odd", input_size=48,
                   store=V.VariableStore())
    with pytest.raises(NotImplementedError):
        tspgnn.Mlp([64], output_size=64, activations=["tanh"], name="t", input_size=64, store=V.VariableStore())


def test_session_requires_gpu_or_fails_loudly():
    tspgnn.build_network(32)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            tspgnn.Session()


def test_product_does_not_import_oracle():
    import os, re
    from conftest import ROOT
    pkg = os.path.join(ROOT, "tsp-gnn_amd", "tspgnn")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_shard_instances_balances_edges_and_keeps_pairs():
    rng = np.random.RandomState(0)
    sizes = rng.randint(20, 81, size=64)
    base = [tspgnn.random_instance(int(n), rng) for n in sizes]
    instances = [inst for inst in base for _ in (0, 1)]          # the loader yields every instance twice
    shards = tspgnn.shard_instances(instances, 8)
    assert sorted(i for s in shards for i in s) == list(range(128))
    for s in shards:                                             # +/- pairs stay together, in order
        assert all(s[k] % 2 == 0 and s[k + 1] == s[k] + 1 for k in range(0, len(s), 2))
    loads = [sum(np.count_nonzero(instances[i][0]) for i in s) for s in shards]
    assert max(loads) <= 1.10 * (sum(loads) / 8.0)               # within 10 % of the mean edge load
    by_count = [len(instances) // 8] * 8                         # what naive equal-count sharding would give
    naive = [sum(np.count_nonzero(instances[i][0]) for i in range(r * 16, r * 16 + 16)) for r in range(8)]
    assert max(loads) <= max(naive)


def test_batch_prefetcher_keeps_the_order_with_several_workers():
    """parallel.BatchPrefetcher with ``workers`` packer threads and ``pack=``: batches come out in the iterator's order
    whatever order the workers finish in, and a failing item surfaces in the consumer after the batches before it
    (device="cpu": plumbing only, nothing is launched)."""
    import time
    import numpy as np
    import pytest
    import tspgnn
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model, device="cpu")
    rng = np.random.RandomState(0)
    insts = [[tspgnn.random_instance(int(rng.randint(4, 9)), rng) for _ in range(2)] for _ in range(9)]
    delays = rng.rand(len(insts)) * 0.01

    def pack(inst):
        time.sleep(delays[[i is inst for i in insts].index(True)])
        return tspgnn.InstanceLoader.create_batch(inst, dev=0.02)
    got = [b.M for b in tspgnn.BatchPrefetcher(sess, iter(insts), 2, workers=3, pack=pack)]
    assert got == [tspgnn.InstanceLoader.create_batch(i)[0].shape[0] for i in insts]

    def bad(inst):
        if inst is insts[4]:
            raise ValueError("boom")
        return tspgnn.InstanceLoader.create_batch(inst)
    seen = []
    with pytest.raises(RuntimeError, match="worker failed"):
        for b in tspgnn.BatchPrefetcher(sess, iter(insts), 2, workers=2, pack=bad):
            seen.append(b.M)
    assert seen == got[:4]


def test_abandoned_batch_prefetcher_is_collected_and_releases_its_workers():
    """a consumer that drops a BatchPrefetcher half-way (no close(), no `with`) must not leave its worker
    threads blocked on the full queue for the life of the process -- the threads hold the shared state, not the
    prefetcher, so the prefetcher is collected and its finaliser closes the state."""
    import gc
    import time
    import weakref
    import tspgnn
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model, device="cpu")
    rng = np.random.RandomState(1)
    insts = [[tspgnn.random_instance(5, rng) for _ in range(2)] for _ in range(40)]
    pf = tspgnn.BatchPrefetcher(sess, iter(insts), 2, workers=2, depth=2,
                                pack=lambda inst: tspgnn.InstanceLoader.create_batch(inst, dev=0.02))
    next(pf)                                  # the workers now sit on a full queue
    threads = list(pf._state._threads)
    ref = weakref.ref(pf)
    del pf
    gc.collect()
    assert ref() is None, "the prefetcher is still referenced (by its own workers?)"
    deadline = time.time() + 10.0
    while any(t.is_alive() for t in threads) and time.time() < deadline:
        time.sleep(0.02)
    assert not any(t.is_alive() for t in threads), "worker threads outlived the abandoned prefetcher"
