"""``torch.nn.BatchNorm1d`` / ``torch.nn.LayerNorm`` with the same parameter names, buffers and defaults, computing
through the HIP library (csrc/norm.hip): what the reference's MPNN builds at model/mpnn.py:34-44 and applies at
:53-56.  Subclasses of the torch modules, so ``state_dict`` keys, ``train()`` / ``eval()`` and ``isinstance`` checks
are the reference's."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from . import functional as Fh


class LayerNorm(nn.LayerNorm):
    def __init__(self, normalized_shape: int, eps: float = 1e-5):
        super().__init__(int(normalized_shape), eps=eps, elementwise_affine=True)

    def forward(self, x: Tensor) -> Tensor:
        if x.dim() != 2 or x.size(1) != self.normalized_shape[0]:
            raise ValueError("graph_hscn.nn.LayerNorm normalises the last dimension of [N, H] activations")
        return Fh.LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class BatchNorm1d(nn.BatchNorm1d):
    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__(int(num_features), eps=eps, momentum=momentum, affine=True, track_running_stats=True)

    def forward(self, x: Tensor) -> Tensor:
        if x.dim() != 2 or x.size(1) != self.num_features:
            raise ValueError("graph_hscn.nn.BatchNorm1d takes [N, H] activations")
        if self.training:
            if x.size(0) < 2:
                raise ValueError("Expected more than 1 value per channel when training")   # torch's own refusal
            self.num_batches_tracked.add_(1)
        return Fh.BatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                    self.momentum, self.eps)
