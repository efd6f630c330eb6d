"""tspgnn -- MI355X-native implementation of the TSP-GNN message-passing hot path behind the
reference's own Python surface (build_network / GraphNN / Mlp / InstanceLoader.create_batch).

Compute is hand-written HIP for gfx950 in libtspgnn.so (C ABI: include/tspgnn.h); this package
is the host-side mirror of the reference interface.  Importing it requires the built library.
"""
from . import _lib  # noqa: F401  (raises ImportError if libtspgnn.so is missing)
from ._lib import TspgnnError
from .graphnn import GraphNN, LSTMStateTuple, DeviceAdjacency, LayerNormBasicLSTMCell
from .instance_loader import InstanceLoader, SparseEV, read_graph, write_graph, synthetic_batch, random_instance
from .binary_search import get_cost
from .mlp import Mlp
from .parallel import BatchPrefetcher, BatchStager, shard_instances
from .model import build_network, Session, global_variables_initializer
from .variables import VariableStore, get_default_store, reset_default_store
from . import tf_checkpoint
from .util import load_weights, save_weights
from .train import run_batch, summarize_epoch
from . import experiments

__all__ = [
    "TspgnnError", "GraphNN", "LSTMStateTuple", "DeviceAdjacency", "LayerNormBasicLSTMCell", "InstanceLoader",
    "SparseEV", "read_graph", "write_graph", "synthetic_batch", "random_instance", "Mlp", "build_network",
    "Session", "global_variables_initializer", "get_cost", "BatchPrefetcher", "BatchStager", "shard_instances", "VariableStore", "get_default_store", "reset_default_store",
    "load_weights", "save_weights", "run_batch", "summarize_epoch",
]
