"""oracle/ -- CPU restatement of the reference's algorithm for the TSP-GNN hot path.

THIS IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker -- never as the
thing measured or shipped.  The product (``tsp-gnn_amd/tspgnn``) never imports this package
and raises when the HIP library is missing.

PARITY UNPINNED.  The arithmetic of the reference's path lives in TensorFlow 1.x
(``tf.contrib.rnn.LayerNormBasicLSTMCell``, ``tf.layers.Dense``, ``tf.matmul`` ...), a
third-party dependency that is neither vendored under /root/reference nor pinned to a
version there (no requirements.txt / lockfile; tf.contrib implies 1.x <= 1.15), and the
reference holds no tests, golden vectors or fixtures for this path.
The oracle therefore restates the published TF-1.x semantics and anchors on what *can* be
executed from the reference in this container: ``instance_loader.InstanceLoader.create_batch``
(numpy only), whose outputs are committed as fixtures under tests/golden/ by
``oracle/gen_golden.py`` and pin the batch layout (edge order, EV pattern, W, C quirk, labels).

Files: ``params.py`` (variable inventory + initialisers), ``torch_oracle.py`` (dense and index
variants, autograd gradients, clip, Adam, cpu_baseline timer), ``np_oracle.py`` (independent
NumPy forward + kernel-level helpers), ``gen_golden.py`` (fixture generator, runs only where
/root/reference exists).
"""
