#!/usr/bin/env python
"""Drop-in demonstration: the reference's run_batch (/root/reference/train.py:17-63; tspgnn/train.py) running on
the MI355X path, over synthetic Euclidean instances (no Concorde, no dataset on disk).

    python examples/train_synthetic.py -d 64 -timesteps 32 -batchsize 8 -epochs 2 --batches 8

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
from tspgnn import InstanceLoader, Session, build_network, global_variables_initializer, random_instance  # noqa: E402
from tspgnn.train import run_batch, summarize_epoch  # noqa: E402,F401


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('-d', default=64, type=int)
    p.add_argument('-timesteps', default=32, type=int)
    p.add_argument('-dev', default=0.02, type=float)
    p.add_argument('-epochs', default=2, type=int)
    p.add_argument('-batchsize', default=8, type=int)
    p.add_argument('-seed', default=42, type=int)
    p.add_argument('--batches', default=8, type=int, help='batches per epoch')
    a = p.parse_args()
    rng = np.random.RandomState(a.seed)
    GNN = build_network(a.d)
    with Session() as sess:
        sess.run(global_variables_initializer(seed=a.seed))
        for epoch in range(a.epochs):
            losses = []
            for b in range(a.batches):
                base = [random_instance(int(rng.randint(20, 41)), rng) for _ in range(a.batchsize)]
                instances = [inst for inst in base for _ in (0, 1)]          # each instance twice
                batch = InstanceLoader.create_batch(instances, dev=a.dev)
                losses.append(run_batch(sess, GNN, batch, b, epoch, a.timesteps, train=True)[0])
            print('Train Epoch {} Average\t|\tLoss={:.4f}'.format(epoch, float(np.mean(losses))), flush=True)
