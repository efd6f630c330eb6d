#!/usr/bin/env python
"""Drop-in demonstration: the reference's run_batch (/root/reference/train.py:17-63; tspgnn/train.py) running on
the MI355X path, over synthetic Euclidean instances (no Concorde, no dataset on disk).

    python examples/train_synthetic.py -d 64 -timesteps 32 -batchsize 8 -epochs 2 --batches 8
    python examples/train_synthetic.py --instances /tmp/tsp/instances --checkpoints /tmp/tsp/ckpt   # .graph files + resume

The flags keep train.py's names (train.py:107-119).  Every instance appears twice in a batch with target
cost (1-dev) and (1+dev) times its tour cost and labels 0/1, exactly like InstanceLoader.get_batches
(instance_loader.py:16-27,82-87).  The "tour" is the planted Hamiltonian cycle, not the optimum, so the
accuracy printed here says nothing about TSP -- the point is the call sequence and the loss going down.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
from tspgnn import (InstanceLoader, Session, build_network, global_variables_initializer, load_weights,  # noqa: E402
                    random_instance, save_weights, write_graph)
from tspgnn.train import run_batch, summarize_epoch  # noqa: E402


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('-d', default=64, type=int)
    p.add_argument('-timesteps', default=32, type=int)
    p.add_argument('-dev', default=0.02, type=float)
    p.add_argument('-epochs', default=2, type=int)
    p.add_argument('-batchsize', default=8, type=int)
    p.add_argument('-seed', default=42, type=int)
    p.add_argument('--batches', default=8, type=int, help='batches per epoch')
    p.add_argument('--instances', default=None, help='directory of .graph files (written with synthetic instances if '
                                                     'missing); batches then come from InstanceLoader.get_batches like train.py:178')
    p.add_argument('--checkpoints', default=None, help='directory for TensorFlow-format checkpoints: resumes from the '
                                                       'newest epoch=N found there and saves after every epoch (train.py:205)')
    a = p.parse_args()
    rng = np.random.RandomState(a.seed)
    loader = None
    if a.instances is not None:
        if not os.path.isdir(a.instances):
            os.makedirs(a.instances)
            for i in range(a.batchsize * a.batches):
                Ma, Mw, route = random_instance(int(rng.randint(20, 41)), rng)
                write_graph(Ma, Mw, os.path.join(a.instances, '{}.graph'.format(i)), route=route)
        loader = InstanceLoader(a.instances)
    GNN = build_network(a.d)
    with Session() as sess:
        sess.run(global_variables_initializer(seed=a.seed))
        first = 0
        if a.checkpoints is not None and os.path.isdir(a.checkpoints):
            saved = sorted(int(x.split('=')[1]) for x in os.listdir(a.checkpoints) if x.startswith('epoch='))
            if saved:
                first = load_weights(sess, '{}/epoch={}'.format(a.checkpoints, saved[-1])) + 1
        for epoch in range(first, first + a.epochs):
            stats = []
            if loader is not None:
                loader.reset()
                batches = loader.get_batches(a.batchsize, a.dev)
            else:
                def synthetic():
                    for _ in range(a.batches):
                        base = [random_instance(int(rng.randint(20, 41)), rng) for _ in range(a.batchsize)]
                        yield InstanceLoader.create_batch([inst for inst in base for _ in (0, 1)], dev=a.dev)   # each twice
                batches = synthetic()
            for b, batch in enumerate(batches):
                stats.append(run_batch(sess, GNN, batch, b, epoch, a.timesteps, train=True)[:4])
            summarize_epoch(epoch, *zip(*stats), train=True)
            if a.checkpoints is not None:
                save_weights(sess, '{}/epoch={}'.format(a.checkpoints, epoch))
