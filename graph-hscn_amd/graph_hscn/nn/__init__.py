"""Operator surface mirroring the torch_geometric names the reference imports
(model/hscn.py:6-14, config/config.py:8, train/train_clustering.py:6)."""
from .conv import GATConv, GCNConv, GraphConv, HeteroConv, Linear
from .norm import BatchNorm1d, LayerNorm
from .pool import dense_mincut_pool, gcn_norm, global_mean_pool, mincut_pool_sparse, to_dense_adj

__all__ = [
    "GATConv", "GCNConv", "GraphConv", "HeteroConv", "Linear", "BatchNorm1d", "LayerNorm",
    "dense_mincut_pool", "gcn_norm", "global_mean_pool", "mincut_pool_sparse", "to_dense_adj",
]
