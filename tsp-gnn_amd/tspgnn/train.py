"""Per-batch driver of the MI355X path with the calling contract of the reference's ``run_batch`` /
``summarize_epoch`` (/root/reference/train.py:17-76): same argument lists, and ``run_batch`` returns the
8-tuple ``(loss, acc, mean label, mean prediction, TP, FP, TN, FN)`` a training script written against the
reference unpacks.  Everything else -- how the feed is assembled, what is logged and how -- is this
package's own.

In a data-parallel session (``Session(..., process_group=...)``) the statistics that come back are the ones
of the GLOBAL batch (``Session`` reduces them across ranks); the label / prediction means of
the tuple are reduced here the same way, so every rank returns identical numbers.
"""
import sys

import numpy as np

_FEED_ORDER = ("EV", "W", "C", "route_exists", "n_vertices", "n_edges")
_STAT_KEYS = ("loss", "acc", "predictions", "TP", "FP", "TN", "FN")


def _feed_for(model, batch, time_steps):
    """feed_dict of one ``InstanceLoader`` batch (a 6-tuple in ``_FEED_ORDER``)."""
    if len(batch) != len(_FEED_ORDER):
        raise ValueError("run_batch: a batch is the 6-tuple (%s), got %d items" % (", ".join(_FEED_ORDER), len(batch)))
    feed = {model[key]: value for key, value in zip(_FEED_ORDER, batch)}
    feed[model["time_steps"]] = time_steps
    return feed


def _batch_means(sess, labels, predictions):
    """(mean label, mean prediction): of this batch, or -- in a data-parallel session -- of the global batch
    (local sums and counts summed over the session's process group)."""
    if getattr(sess, "world_size", 1) <= 1:
        return np.mean(labels), np.mean(predictions)
    sums = sess.allreduce_host_sums(np.array(
        [np.sum(labels, dtype=np.float64), np.sum(predictions, dtype=np.float64), float(len(labels))]))
    count = max(sums[2], 1.0)
    return sums[0] / count, sums[1] / count


def _log(kind, epoch_i, fields):
    sys.stdout.write("[%s] epoch %d  %s\n" % (kind, epoch_i, "  ".join("%s=%s" % kv for kv in fields)))
    sys.stdout.flush()


def run_batch(sess, model, batch, batch_i, epoch_i, time_steps, train=False, verbose=True):
    """One ``sess.run`` over ``batch``; with ``train`` the optimiser step is fetched first (train.py:36-42)."""
    fetches = [model[key] for key in _STAT_KEYS]
    if train:
        fetches.insert(0, model["train_step"])
    values = sess.run(fetches, feed_dict=_feed_for(model, batch, time_steps))
    stats = dict(zip(_STAT_KEYS, values[len(values) - len(_STAT_KEYS):]))
    labels, n_vertices, n_edges = batch[3], batch[4], batch[5]
    mean_label, mean_pred = _batch_means(sess, labels, stats["predictions"])
    if verbose:
        _log("train" if train else "test", epoch_i, (
            ("batch", batch_i),
            ("vertices", int(np.sum(n_vertices))), ("edges", int(np.sum(n_edges))), ("graphs", len(n_vertices)),
            ("loss", "%.4f" % stats["loss"]), ("acc", "%.4f" % stats["acc"]),
            ("label_mean", "%.4f" % mean_label),
            ("decided_yes", "%.4f" % float(np.mean(np.round(stats["predictions"])))),
        ))
    return (stats["loss"], stats["acc"], mean_label, mean_pred,
            stats["TP"], stats["FP"], stats["TN"], stats["FN"])


def summarize_epoch(epoch_i, loss, acc, sat, pred, train=False):
    """Epoch line: the means of the per-batch values ``run_batch`` returned (train.py:66-76)."""
    _log("train" if train else "test", epoch_i, (
        ("batches", len(loss)),
        ("loss", "%.4f" % float(np.mean(loss))), ("acc", "%.4f" % float(np.mean(acc))),
        ("label_mean", "%.4f" % float(np.mean(sat))), ("pred_mean", "%.4f" % float(np.mean(pred))),
    ))
