"""Message-passing modules with the torch_geometric signatures, parameter names
and initialisers the reference relies on (model/hscn.py:6-14,83-96,117-125;
config/config.py:19-23), computing through the HIP kernels.

``state_dict`` keys match PyG 2.2/2.3 so checkpoints move
between the reference and this package unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from .._hip import ACT
from ..structure import Relation, relation_of, self_loop_relation_of
from . import functional as Fh


def _glorot(t: Tensor) -> Tensor:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-a, a)


class Linear(nn.Module):
    """torch_geometric.nn.Linear: ``y = x W^T + b``; 'glorot' or kaiming-uniform
    (a=sqrt 5) weights; ``in_channels=-1`` is materialised on first use."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True,
                 weight_initializer: Optional[str] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer = weight_initializer
        if in_channels > 0:
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        else:
            self.weight = nn.parameter.UninitializedParameter()
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.in_channels <= 0:
            return
        if self.weight_initializer == "glorot":
            _glorot(self.weight)
        else:
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels)
            nn.init.uniform_(self.bias, -bound, bound)

    def materialize(self, in_channels: int, like: Tensor) -> None:
        if isinstance(self.weight, nn.parameter.UninitializedParameter):
            self.in_channels = int(in_channels)
            self.weight.materialize((self.out_channels, self.in_channels), device=like.device,
                                    dtype=torch.float32)
            self.reset_parameters()

    def forward(self, x: Tensor, act: str = "identity") -> Tensor:
        self.materialize(x.size(-1), x)
        return Fh.linear(x, self.weight, self.bias, act)


def _relation(edge_index: Union[Tensor, Relation], num_src: int, num_dst: int, both: bool = False) -> Relation:
    if isinstance(edge_index, Relation):
        return edge_index
    return relation_of(edge_index, num_src, num_dst, both=both)


class GraphConv(nn.Module):
    """PyG GraphConv(aggr='add'): ``lin_rel(sum_j w_ji x_j) + lin_root(x_i)``
    (used at model/hscn.py:32,40)."""

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "add", bias: bool = True):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("GraphConv on the hot path uses aggr='add'")
        self.lin_rel = Linear(in_channels, out_channels, bias=bias)
        self.lin_root = Linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Union[Tensor, Relation], edge_weight: Optional[Tensor] = None,
                act: str = "identity") -> Tensor:
        # (sum_j w_ji x_j) W_rel + x W_root: only a gradient w.r.t. x walks the source-keyed CSR
        rel = _relation(edge_index, x.size(0), x.size(0), both=torch.is_grad_enabled() and x.requires_grad)
        return Fh.GraphConvFn.apply(x, edge_weight, self.lin_rel.weight, self.lin_rel.bias,
                                    self.lin_root.weight, rel, ACT[act])


class GCNConv(nn.Module):
    """PyG GCNConv, ``normalize=True``, unit edge weights.
    ``add_self_loops=False`` is what build_conv_relation constructs (model/hscn.py:117-125);
    the default ``add_self_loops=True`` is what the MPNN baseline constructs (model/mpnn.py:28-32):
    the same kernels over the edge list with one loop per node appended, whose in-degree is
    gcn_norm's degree."""

    def __init__(self, in_channels: int, out_channels: int, add_self_loops: bool = True,
                 cached: bool = False, bias: bool = True):
        super().__init__()
        self.add_self_loops = bool(add_self_loops)
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x: Tensor, edge_index: Union[Tensor, Relation], act: str = "identity") -> Tensor:
        self.lin.materialize(x.size(-1), x)
        if isinstance(edge_index, Relation):
            rel = edge_index                      # the caller built the structure (loops included if wanted)
        else:
            # A_hat (X W): with gradients on, the backward walks the source-keyed CSR for the weight gradient already
            both = torch.is_grad_enabled()
            if self.add_self_loops:
                rel = self_loop_relation_of(edge_index, x.size(0), both=both)
            else:
                rel = relation_of(edge_index, x.size(0), x.size(0), both=both)
        return Fh.GCNConvFn.apply(x, self.lin.weight, self.bias, rel, ACT[act])


class GATConv(nn.Module):
    """PyG GATConv as built for local->virtual (model/hscn.py:85-87): bipartite
    ``in_channels=(-1,-1)``, heads=1, negative_slope=0.2, dropout=0,
    ``add_self_loops=False``."""

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int, heads: int = 1,
                 negative_slope: float = 0.2, add_self_loops: bool = True, cached: bool = False,
                 bias: bool = True):
        super().__init__()
        if heads != 1 or add_self_loops:
            raise NotImplementedError("the hot path builds GATConv with heads=1, add_self_loops=False")
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.negative_slope = negative_slope
        self.out_channels = out_channels
        self.lin_src = Linear(in_channels[0], out_channels, bias=False, weight_initializer="glorot")
        self.lin_dst = Linear(in_channels[1], out_channels, bias=False, weight_initializer="glorot")
        self.att_src = nn.Parameter(_glorot(torch.empty(1, 1, out_channels)))
        self.att_dst = nn.Parameter(_glorot(torch.empty(1, 1, out_channels)))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x: Union[Tensor, Tuple[Tensor, Tensor]], edge_index: Union[Tensor, Relation],
                act: str = "identity") -> Tensor:
        x_src, x_dst = (x, x) if isinstance(x, Tensor) else x
        self.lin_src.materialize(x_src.size(-1), x_src)
        self.lin_dst.materialize(x_dst.size(-1), x_dst)
        rel = _relation(edge_index, x_src.size(0), x_dst.size(0), both=torch.is_grad_enabled())
        return Fh.GATConvFn.apply(x_src, x_dst, self.lin_src.weight, self.lin_dst.weight, self.att_src,
                                  self.att_dst, self.bias, rel, self.negative_slope, ACT[act])


class HeteroConv(nn.Module):
    """PyG HeteroConv(aggr='sum'): run each relation's conv in
    ``edge_index_dict`` order, sum the outputs that share a target type."""

    def __init__(self, convs: Dict[Tuple[str, str, str], nn.Module], aggr: str = "sum"):
        super().__init__()
        if aggr != "sum":
            raise NotImplementedError("HeteroConv on the hot path uses aggr='sum'")
        self.convs = nn.ModuleDict({"__".join(k): v for k, v in convs.items()})

    def forward(self, x_dict: Dict[str, Tensor], edge_index_dict: Dict[Tuple[str, str, str], Tensor]):
        outs: Dict[str, list] = {}
        for edge_type, edge_index in edge_index_dict.items():
            src, _, dst = edge_type
            key = "__".join(edge_type)
            if key not in self.convs:
                continue
            conv = self.convs[key]
            if src == dst:
                o = conv(x_dict[src], edge_index)
            else:
                o = conv((x_dict[src], x_dict[dst]), edge_index)
            outs.setdefault(dst, []).append(o)
        return {k: (v[0] if len(v) == 1 else torch.stack(v, 0).sum(0)) for k, v in outs.items()}
