"""Binary search on the target cost with the decision network -- the inference caller of the hot path
in the reference (/root/reference/experiments/binary_search.py:13-78, ``get_cost``).

``get_cost(sess, model, instance, time_steps)`` keeps the reference's signature, return tuple and loop
(one ``sess.run(model['predictions'])`` per probe, same bracket updates and stopping rule).
``parallel=k`` (k > 1) is the MI355X-shaped variant: every round packs k copies of the instance with k
target costs spread over the bracket into ONE batch -- the graphs are independent blocks of EV -- and
narrows the bracket by a factor k+1 per forward pass instead of 2.
"""
import numpy as np

from .instance_loader import InstanceLoader


def cost_bounds(Mw, n):
    """Bracket of the per-vertex tour cost the search starts from, as the reference forms it
    (experiments/binary_search.py:21-33): each triangle of the weight matrix is laid out as a full n x n array
    (the other triangle zero-filled -- the zeros take part in the ranking, so for non-negative weights the lower
    end of the bracket is 0), its n smallest and n largest entries are summed, and the looser of the two
    triangles wins on either side.  Returned per vertex (divided by n)."""
    lows, highs = [], []
    for triangle in (np.triu(Mw), np.tril(Mw)):
        ranked = np.sort(triangle, axis=None)
        lows.append(ranked[:n].sum())
        highs.append(ranked[ranked.size - n:].sum())
    return min(lows) / n, max(highs) / n


def get_cost(sess, model, instance, time_steps, threshold=0.5, stopping_delta=0.01, parallel=1):
    Ma, Mw, route = instance
    n = Ma.shape[0]
    m = len(np.nonzero(Ma)[0])
    wmin, wmax = cost_bounds(Mw, n)
    wpred = (wmin + wmax) / 2
    # the true closing edge, unlike create_batch's quirk (binary_search.py:40 vs instance_loader.py:70)
    route_cost = sum(Mw[min(i, j), max(i, j)] for (i, j) in zip(route, route[1:] + route[:1])) / n
    k = max(1, int(parallel))
    EV, W, _, route_exists, n_vertices, n_edges = InstanceLoader.create_batch([(Ma, Mw, route)] * k, target_cost=wpred)
    feed_dict = {model["EV"]: EV, model["W"]: W, model["C"]: None, model["time_steps"]: time_steps,
                 model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    iterations, pred = 0, None
    while wmin < wpred * (1 - stopping_delta) or wpred * (1 + stopping_delta) < wmax:
        if k == 1:
            feed_dict[model["C"]] = np.ones((m, 1)) * wpred
            pred = sess.run(model["predictions"], feed_dict=feed_dict)
            if pred < threshold:
                wmin = wpred
            else:
                wmax = wpred
        else:
            probes = wmin + (wmax - wmin) * (np.arange(1, k + 1) / (k + 1.0))
            feed_dict[model["C"]] = np.repeat(probes, m).reshape(-1, 1)
            preds = sess.run(model["predictions"], feed_dict=feed_dict)
            # the answer is monotone in the target cost for a trained model: keep the bracket around
            # the first probe the network accepts
            accept = np.nonzero(preds >= threshold)[0]
            first = accept[0] if len(accept) else k
            lo = wmin if first == 0 else probes[first - 1]
            hi = wmax if first == k else probes[first]
            wmin, wmax = lo, hi
            pred = preds[min(first, k - 1):min(first, k - 1) + 1]
        wpred = (wmax + wmin) / 2
        iterations += 1
    return wpred, pred, route_cost, iterations
