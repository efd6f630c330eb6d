"""The measurement routines of the reference's experiment scripts on the MI355X path, without the plotting:
``get_predictions`` (/root/reference/experiments/acceptance_curve.py:19-37), ``get_accuracy``
(experiments/test_varying_sizes.py:20-38, test_varying_dev.py) and the two sweeps built on them.  Each keeps the
reference's feed / fetch; the sweeps call them exactly like the scripts' loops (acceptance_curve.py:87-99,
test_varying_sizes.py:82-114)."""
from itertools import islice

import numpy as np


def _feed(model, batch, time_steps):
    EV, W, C, route_exists, n_vertices, n_edges = batch[0], batch[1], batch[2], batch[-3], batch[-2], batch[-1]
    return {model['EV']: EV, model['W']: W, model['C']: C, model['time_steps']: time_steps,
            model['route_exists']: route_exists, model['n_vertices']: n_vertices, model['n_edges']: n_edges}


def get_predictions(sess, model, batch, time_steps):
    """``batch``: the 6-tuple of create_batch (a 7-tuple with an edges_mask in fourth place is accepted, as in
    acceptance_curve.py:21)."""
    return sess.run(model['predictions'], feed_dict=_feed(model, batch, time_steps))


def get_accuracy(sess, model, batch, time_steps):
    return np.mean(sess.run(model['acc'], feed_dict=_feed(model, batch, time_steps)))


def acceptance_curve(sess, model, loader, time_steps, deviations, batch_size=16, max_batches=64):
    """Mean prediction as a function of the target-cost deviation (acceptance_curve.py:87-99): for every deviation
    the loader is rewound and up to ``max_batches`` batches (every instance twice: (1-dev) and (1+dev) times its
    tour cost) are scored.  Returns an array of len(deviations) means."""
    out = np.zeros(len(deviations))
    for i, dev in enumerate(deviations):
        loader.reset()
        preds = [get_predictions(sess, model, b, time_steps) for b in islice(loader.get_batches(batch_size, dev), max_batches)]
        out[i] = np.mean(np.concatenate(preds)) if preds else np.nan
    return out


def accuracy_by_size(sess, model, loaders, time_steps, dev, batch_size=16, max_batches=64):
    """{n: accuracy} over per-size instance loaders (test_varying_sizes.py:82-114)."""
    result = {}
    for n, loader in loaders.items():
        loader.reset()
        accs = [get_accuracy(sess, model, b, time_steps) for b in islice(loader.get_batches(batch_size, dev), max_batches)]
        result[n] = float(np.mean(accs)) if accs else float('nan')
    return result
