"""The MPNN baseline with the reference's signature (model/mpnn.py:13-78; BASELINE config 1,
configs/GCN/peptides_func_GCN.yaml): ``num_layers`` convolutions ``F -> H -> ... -> C``, each hidden
one followed by ReLU, the configured activation and dropout, then a per-graph mean of the last
convolution's node outputs.  Everything numerical runs in the HIP library: GCNConv (transform +
normalised CSR gather-reduce with the ReLU in its epilogue), the activation, the counter-based
dropout and the segment mean.

Normalisation layers (model/mpnn.py:34-44,53-56) are built and applied exactly as the reference does, quirk
included: BOTH module lists -- ``bns`` (BatchNorm1d) and ``lns`` (LayerNorm) -- are created under ``use_layer_norm``
(mpnn.py:35 tests the wrong flag), so ``use_batch_norm=True`` alone reads a ``self.bns`` that does not exist and the
forward raises AttributeError, as the reference's does.  The layers compute through csrc/norm.hip.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn
from torch import Tensor

from ..config.config import ACT_DICT, CONV_DICT, MPNNConfig
from ..nn import functional as Fh
from ..nn.norm import BatchNorm1d, LayerNorm
from ..nn.pool import global_mean_pool


class MPNN(nn.Module):
    def __init__(self, conv: type, activation: Callable, num_features: int, hidden_channels: int,
                 num_classes: int, num_layers: int, dropout: float = 0.0, use_batch_norm: bool = False,
                 use_layer_norm: bool = False) -> None:
        super().__init__()
        self.num_layers = num_layers
        self.conv_layers = nn.ModuleList()                                  # mpnn.py:27-32
        self.conv_layers.append(conv(num_features, hidden_channels))
        for _ in range(num_layers - 2):
            self.conv_layers.append(conv(hidden_channels, hidden_channels))
        self.conv_layers.append(conv(hidden_channels, num_classes))
        self.use_batch_norm = use_batch_norm
        if use_layer_norm:                                                  # mpnn.py:35-38 (sic: not use_batch_norm)
            self.bns = nn.ModuleList(BatchNorm1d(hidden_channels) for _ in range(num_layers - 1))
        self.use_layer_norm = use_layer_norm
        if use_layer_norm:                                                  # mpnn.py:41-44
            self.lns = nn.ModuleList(LayerNorm(hidden_channels) for _ in range(num_layers - 1))
        self.activation = activation
        self.dropout = dropout
        self.dropout_seed: Optional[int] = None     # tests pin the mask; None = torch.initial_seed() + call counter

    def forward(self, batch) -> Tensor:
        x, edge_index, batch_vec = batch.x, batch.edge_index, batch.batch   # mpnn.py:50
        act_name = getattr(self.activation, "hscn_name", None)
        for i in range(self.num_layers - 1):
            x = self.conv_layers[i](x, edge_index, act="relu")              # F.relu(conv(x)) in the epilogue
            if self.use_batch_norm:
                x = self.bns[i](x)                                          # mpnn.py:53-54
            if self.use_layer_norm:
                x = self.lns[i](x)                                          # mpnn.py:55-56
            normed = self.use_batch_norm or self.use_layer_norm             # (a normalised x is no longer >= 0)
            if normed or act_name not in ("relu", "identity", "elu"):       # relu / elu are the identity on x >= 0
                x = self.activation(x)
            seed = None if self.dropout_seed is None else self.dropout_seed + i
            x = Fh.dropout(x, p=self.dropout, training=self.training, seed=seed)
        x = self.conv_layers[-1](x, edge_index)
        size = getattr(batch, "num_graphs", None)
        return global_mean_pool(x, batch_vec, size)                         # scatter_mean(x, batch, dim=0)


def build_mpnn(model_cfg: MPNNConfig, num_features: int, num_classes: int) -> MPNN:  # mpnn.py:65-78
    return MPNN(CONV_DICT[model_cfg.conv_type.lower()], ACT_DICT[model_cfg.activation.lower()], num_features,
                model_cfg.hidden_channels, num_classes, model_cfg.num_layers, model_cfg.dropout,
                model_cfg.use_batch_norm, model_cfg.use_layer_norm)
