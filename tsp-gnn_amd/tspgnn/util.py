"""``load_weights`` / ``save_weights`` with the reference's signatures (/root/reference/util.py:5-37).

The reference wraps ``tf.train.Saver``: a checkpoint is the directory ``path`` holding ``model.ckpt.index`` and
``model.ckpt.data-00000-of-00001``, keyed by TensorFlow variable names (``TSP/E_cell/layer_norm_basic_lstm_cell/
kernel`` ...), optimiser slots included (``<var>/Adam``, ``<var>/Adam_1``, ``beta1_power``, ``beta2_power``).
The same files are written and read here (tf_checkpoint.py), so a model trained with the reference loads into
this implementation and the other way round.
"""
import math
import os

import numpy as np
import torch

from . import tf_checkpoint

ADAM_BETA1, ADAM_BETA2 = 0.9, 0.999   # tf.train.AdamOptimizer defaults (model.py:160)


def _epoch_of(path):
    tail = os.path.normpath(path).split(os.sep)[-1]
    return int(tail.replace("epoch=", ""))     # util.py:10 -- raises ValueError for other directory names


def load_weights(sess, path, scope=None):
    """Restore the variables (all of them, or those under ``scope``) and, when the checkpoint holds them, the
    Adam moments and step count.  Returns the epoch encoded in the directory name ``.../epoch=N``."""
    if not os.path.exists(path):
        raise Exception('Path does not exist!')
    print('Restoring saved model ... ')
    epoch = _epoch_of(path)
    ckpt = tf_checkpoint.read_bundle("%s/model.ckpt" % path)
    store = sess.store
    names = [n for n in store.names() if scope is None or n.startswith(scope)]
    missing = [n for n in names if n not in ckpt]
    if missing:
        raise KeyError("checkpoint %s lacks variables: %s" % (path, ", ".join(missing[:5])))
    store.load({n: ckpt[n] for n in names})
    if "beta1_power" in ckpt and all((n + "/Adam") in ckpt and (n + "/Adam_1") in ckpt for n in names):
        sess._ensure_adam()
        for slot, buf in (("/Adam", sess._adam["m"]), ("/Adam_1", sess._adam["v"])):
            host = buf.detach().cpu()
            for n in names:
                off, cnt = store._offsets[n]
                host[off:off + cnt] = torch.from_numpy(np.asarray(ckpt[n + slot], dtype=np.float32).reshape(-1).copy())
            buf.copy_(host)
        # TF keeps beta1^(t+1) after t applied steps (the power is advanced at the end of every apply)
        t = max(0, int(round(math.log(float(ckpt["beta1_power"])) / math.log(ADAM_BETA1))) - 1)
        sess._adam["step"] = t
        sess._adam["t"].fill_(t)
    return epoch


def save_weights(sess, path, scope=None):
    """Write ``path/model.ckpt.{index,data-00000-of-00001}`` (+ the ``checkpoint`` state file)."""
    if not os.path.exists(path):
        os.makedirs(path)
    store = sess.store
    tensors = {n: v for n, v in store.state_dict().items() if scope is None or n.startswith(scope)}
    adam = getattr(sess, "_adam", None)
    if adam is not None and scope is None:
        m, v = adam["m"].detach().cpu().numpy(), adam["v"].detach().cpu().numpy()
        for n in store.names():
            off, cnt = store._offsets[n]
            shape = store.shape(n)
            tensors[n + "/Adam"] = m[off:off + cnt].reshape(shape).copy()
            tensors[n + "/Adam_1"] = v[off:off + cnt].reshape(shape).copy()
        t = int(adam["step"])
        tensors["beta1_power"] = np.float32(ADAM_BETA1 ** (t + 1))
        tensors["beta2_power"] = np.float32(ADAM_BETA2 ** (t + 1))
    tf_checkpoint.write_bundle("%s/model.ckpt" % path, {k: np.asarray(v) for k, v in tensors.items()})
    print('Model saved in path: {path}\n'.format(path=path))
