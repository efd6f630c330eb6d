#!/usr/bin/env python
"""Drop-in demonstration: the reference's run_batch (/root/reference/train.py:17-63) running unchanged on
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


def run_batch(sess, model, batch, batch_i, epoch_i, time_steps, train=False, verbose=True):
    # body identical in structure to train.py:17-63
    EV, W, C, route_exists, n_vertices, n_edges = batch
    feed_dict = {model['EV']: EV, model['W']: W, model['C']: C, model['time_steps']: time_steps,
                 model['route_exists']: route_exists, model['n_vertices']: n_vertices, model['n_edges']: n_edges}
    if train:
        outputs = [model['train_step'], model['loss'], model['acc'], model['predictions'], model['TP'], model['FP'],
                   model['TN'], model['FN']]
    else:
        outputs = [model['loss'], model['acc'], model['predictions'], model['TP'], model['FP'], model['TN'], model['FN']]
    loss, acc, predictions, TP, FP, TN, FN = sess.run(outputs, feed_dict=feed_dict)[-7:]
    if verbose:
        print('{} Epoch {} Batch {}\t|\t(n,m,batch size)=({},{},{})\t|\t(Loss,Acc)=({:.4f},{:.4f})\t|\tAvg. (Sat,Prediction)=({:.4f},{:.4f})'
              .format('Train' if train else 'Test', epoch_i, batch_i, np.sum(n_vertices), np.sum(n_edges),
                      n_vertices.shape[0], loss, acc, np.mean(route_exists), np.mean(np.round(predictions))), flush=True)
    return loss, acc, np.mean(route_exists), np.mean(predictions), TP, FP, TN, FN


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
