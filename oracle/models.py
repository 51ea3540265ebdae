"""Restatement of the reference's two models on top of oracle.pyg_ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Follows /root/reference/graph_hscn/model/hscn.py:
  SCN   :19-64   (GraphConv stack -> Linear -> to_dense_adj -> dense_mincut_pool)
  HSCN  :67-114  (L x HeteroConv{lv: GAT, ll: GCN, vv: GCN} -> ReLU -> mean pool -> 2 Linear)
and the stage-A driver /root/reference/graph_hscn/train/train_clustering.py:36-69;
  MPNN  /root/reference/graph_hscn/model/mpnn.py:13-62 (the GCN baseline of BASELINE config 1).
``state_dict`` keys follow PyG naming so weights can
be copied 1:1 into the product modules.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import pyg_ops as P

ACT = {  # reference: config/config.py:13-18
    "elu": F.elu,
    "relu": F.relu,
    "tanh": torch.tanh,
    "identity": lambda t: t,
}


class _MP(nn.Module):
    """PyG ``Sequential('x, edge_index, edge_weight', [(GraphConv, ...), act, ...])``;
    children are registered as ``module_{i}`` (only GraphConv children own params)."""

    def __init__(self, convs: List[P.GraphConv], act: Callable):
        super().__init__()
        self.act = act
        self._n = len(convs)
        for i, c in enumerate(convs):
            setattr(self, f"module_{2 * i}", c)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        for i in range(self._n):
            x = getattr(self, f"module_{2 * i}")(x, edge_index, edge_weight)
            x = self.act(x)
        return x


class SCN(nn.Module):
    """model/hscn.py:19-64.  ``mlp_units`` must be empty-compatible with the
    reference's (buggy, :50-53) loop: each hidden Linear is ``out_channels ->
    units`` and the next still consumes ``out_channels``; only ``mlp_units=[]``
    (the only value the reference passes, main.py:101-105) is shape-safe in
    general, so other values are restated literally."""

    def __init__(self, mp_units: list, mp_act: str, num_features: int, num_clusters: int,
                 mlp_units: list = [], mlp_act: str = "identity"):
        super().__init__()
        convs = [P.GraphConv(num_features, mp_units[0])]
        for i in range(len(mp_units) - 1):
            convs.append(P.GraphConv(mp_units[i], mp_units[i + 1]))
        self.mp = _MP(convs, ACT[mp_act.lower()])
        out_channels = mp_units[-1]
        layers: List[nn.Module] = []
        self._mlp_act = ACT[mlp_act.lower()]
        for units in mlp_units:
            layers.append(P.PygLinear(out_channels, units))
            layers.append(nn.Identity())  # activation slot (keeps Sequential indices)
        layers.append(P.PygLinear(out_channels, num_clusters))
        self.mlp = nn.Sequential(*layers)

    def _run_mlp(self, x: Tensor) -> Tensor:
        for m in self.mlp:
            x = self._mlp_act(x) if isinstance(m, nn.Identity) else m(x)
        return x

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], store: Optional[Callable] = None):
        """``store``: reduced-precision storage emulation (see ``HSCN.forward``): applied to the input features
        and to the message-passing stack's output (the hidden activation the backward re-reads)."""
        if store is not None:
            x = store(x)
        x = self.mp(x, edge_index, edge_weight)
        if store is not None:
            x = store(x)
        s = self._run_mlp(x)
        adj = P.to_dense_adj(edge_index)
        _, _, mc_loss, o_loss = P.dense_mincut_pool(x, adj, s)
        return torch.softmax(s, dim=-1), mc_loss, o_loss, adj


def build_conv_relation(conv_type: str, in_src: int, in_dst: int, hidden: int) -> nn.Module:
    """model/hscn.py:117-125 with the lazy ``-1`` dims made explicit."""
    t = conv_type.lower()
    if t == "gat":
        return P.GATConv((in_src, in_dst), hidden, add_self_loops=False)
    if t == "gcn":
        return P.GCNConv(in_src, hidden, add_self_loops=False)
    raise ValueError(f"conv type {conv_type!r} is not usable on this path")


LL = ("local", "to", "local")
VV = ("virtual", "to", "virtual")
LV = ("local", "to", "virtual")


class HSCN(nn.Module):
    """model/hscn.py:67-114."""

    def __init__(self, lv_conv: str, ll_conv: str, vv_conv: str, activation: Callable,
                 num_features: int, hidden_channels: int, num_classes: int, num_layers: int):
        super().__init__()
        self.activation = activation
        self.convs = nn.ModuleList()
        for layer in range(num_layers):
            fin = num_features if layer == 0 else hidden_channels
            self.convs.append(P.HeteroConv({
                LV: build_conv_relation(lv_conv, fin, fin, hidden_channels),
                LL: build_conv_relation(ll_conv, fin, fin, hidden_channels),
                VV: build_conv_relation(vv_conv, fin, fin, hidden_channels),
            }, aggr="sum"))
        self.lin_1 = P.PygLinear(hidden_channels, hidden_channels)
        self.lin_2 = P.PygLinear(hidden_channels, num_classes)

    def forward(self, x_dict: dict, edge_index_dict: dict, batch_local: Tensor, num_graphs: Optional[int] = None,
                store: Optional[Callable] = None, keep: Optional[dict] = None) -> Tensor:
        """``store``: emulation of a reduced-precision STORAGE type for node features and inter-layer activations
        (BASELINE.json configs[4]; the reference itself has no such mode): applied to the input features and to
        every layer's output of both node types -- the points at which the HIP kernels round -- with float32
        arithmetic in between (``half_storage`` below).  ``keep``: a dict that receives the final ``x_dict``."""
        if store is not None:
            x_dict = {k: store(v) for k, v in x_dict.items()}
        for conv in self.convs:
            x_dict = conv(x_dict, edge_index_dict)
            x_dict = {k: v.relu() for k, v in x_dict.items()}
            if store is not None:
                x_dict = {k: store(v) for k, v in x_dict.items()}
        if keep is not None:
            keep.update(x_dict)
        x = P.global_mean_pool(x_dict["local"], batch_local, num_graphs)
        x = self.activation(self.lin_1(x))
        return self.lin_2(x)


class MPNN(nn.Module):
    """model/mpnn.py:13-62 with ``conv = GCNConv`` (config/config.py:19-23, configs/GCN/peptides_func_GCN.yaml:6):
    GCNConv(add_self_loops=True) stack, ``F.relu`` then the configured activation then dropout after every
    hidden layer, ``scatter_mean`` over the batch vector at the end.  Normalisation layers as the reference builds
    them (mpnn.py:34-44): BOTH lists under ``use_layer_norm`` (``use_batch_norm`` alone is an AttributeError at
    mpnn.py:54), plain ``torch.nn.BatchNorm1d`` / ``torch.nn.LayerNorm``."""

    def __init__(self, activation: Callable, num_features: int, hidden_channels: int, num_classes: int,
                 num_layers: int, dropout: float = 0.0, use_batch_norm: bool = False, use_layer_norm: bool = False):
        super().__init__()
        self.use_batch_norm = use_batch_norm
        if use_layer_norm:
            self.bns = nn.ModuleList(nn.BatchNorm1d(hidden_channels) for _ in range(num_layers - 1))
        self.use_layer_norm = use_layer_norm
        if use_layer_norm:
            self.lns = nn.ModuleList(nn.LayerNorm(hidden_channels) for _ in range(num_layers - 1))
        self.num_layers = num_layers
        self.conv_layers = nn.ModuleList()
        self.conv_layers.append(P.GCNConv(num_features, hidden_channels))
        for _ in range(num_layers - 2):
            self.conv_layers.append(P.GCNConv(hidden_channels, hidden_channels))
        self.conv_layers.append(P.GCNConv(hidden_channels, num_classes))
        self.activation = activation
        self.dropout = dropout

    def forward(self, x: Tensor, edge_index: Tensor, batch: Tensor, num_graphs: Optional[int] = None,
                masks: Optional[List[Tensor]] = None) -> Tensor:
        """``masks`` (one 0/1 tensor per hidden layer) replaces F.dropout's own draw so that a product run
        with a known mask can be compared element for element."""
        for i in range(self.num_layers - 1):
            x = F.relu(self.conv_layers[i](x, edge_index))
            if self.use_batch_norm:
                x = self.bns[i](x)
            if self.use_layer_norm:
                x = self.lns[i](x)
            x = self.activation(x)
            if masks is not None:
                x = x * masks[i] * (1.0 / (1.0 - self.dropout))
            else:
                x = F.dropout(x, p=self.dropout, training=self.training)
        x = self.conv_layers[-1](x, edge_index)
        return P.global_mean_pool(x, batch, num_graphs)


def criterion(loss_fn: str, pred: Tensor, true: Tensor):
    """loss.py:6-19 (quirk kept: the L1 branch scores with sigmoid)."""
    if loss_fn == "cross_entropy":
        if pred.ndim > 1 and true.ndim == 1:
            lp = F.log_softmax(pred, dim=-1)
            return F.nll_loss(lp, true), lp
        true = true.float()
        return F.binary_cross_entropy_with_logits(pred, true, reduction="mean"), torch.sigmoid(pred)
    return F.l1_loss(pred, true), torch.sigmoid(pred)


def half_storage(t: Tensor) -> Tensor:
    """Round to IEEE half and widen again; the gradient passes straight through (the HIP backward treats the
    stored value as the activation)."""
    return t + (t.half().float() - t).detach()


def scn_step_single_graph(model: SCN, x: Tensor, edge_index: Tensor):
    """One body of the stage-A loop, train_clustering.py:37-48: gcn_norm with
    self loops on the raw graph, forward, ``mc + o``."""
    ei, ew = P.gcn_norm(edge_index, None, x.size(0), add_self_loops=True)
    s, mc, o, adj = model(x.float(), ei, ew)
    return s, mc, o, adj, ei, ew


def assign_clusters(soft: Tensor) -> np.ndarray:
    """train_clustering.py:68 -- ``clust.max(1)[1].cpu().numpy()`` (first maximal
    index wins on ties)."""
    return soft.max(1)[1].cpu().numpy()


def train_clustering_loop(model: SCN, graphs, cluster_epochs: int, optimizer: torch.optim.Optimizer):
    """The stage-A driver, /root/reference/graph_hscn/train/train_clustering.py:34-69, restated: per epoch, per
    graph: gcn_norm(add_self_loops=True) (:37-42) -> forward (:45-47) -> ``loss = mc_loss + o_loss`` (:48) ->
    backward, ONE optimizer step per graph (:49-50); then the assignment pass (:57-69): same normalisation and
    forward per graph, ``clust.max(1)[1]`` (:68).  ``graphs``: objects with ``.x`` and ``.edge_index``; the raw
    graph is re-normalised on every visit, as iterating a PyG ``InMemoryDataset`` hands out a fresh copy each time
    (the in-place overwrite at :37 persists only for ``list`` datasets).  Like the reference, the
    assignment pass runs with autograd on (:57-69 has no ``no_grad``); its graphs are simply dropped.
    Returns (list of int64 id arrays, list of the soft assignments of the final pass)."""
    for _ in range(cluster_epochs):
        for g in graphs:
            optimizer.zero_grad()
            _, mc, o, _, _, _ = scn_step_single_graph(model, g.x, g.edge_index)
            loss = mc + o
            loss.backward()
            optimizer.step()
    ids, soft = [], []
    for g in graphs:
        s, *_ = scn_step_single_graph(model, g.x, g.edge_index)
        soft.append(s.detach())
        ids.append(assign_clusters(s))
    return ids, soft
