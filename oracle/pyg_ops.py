"""Restatement of the torch_geometric operators the Graph-HSCN hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

Third-party dependency restated here: ``torch-geometric`` (unpinned in
/root/reference/requirements-cpu.txt:10; the torch==1.13.1 pin at :8 puts it
in the 2.2 / 2.3 series) and ``torch-scatter`` (:9).  Each function names the
reference call site it serves and follows the published PyG algorithm op for
op in its CPU order: ``index_select`` gather -> scale -> ``index_add_``
(sequential in edge order on CPU), int64 COO ``edge_index = [row; col]`` with
messages flowing row (source j) -> col (target i).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


# --------------------------------------------------------------------------- #
# scatter helpers (torch_scatter.scatter_add / scatter_max on CPU)
# --------------------------------------------------------------------------- #
def scatter_add(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``scatter(src, index, dim=0, dim_size, reduce='sum')``: sequential
    accumulation in index order on CPU."""
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return out.index_add_(0, index, src)


def scatter_max(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``scatter(src, index, dim=0, dim_size, reduce='max')``; empty segments
    hold 0 (torch_scatter fills untouched slots with 0)."""
    out = src.new_full((dim_size,) + tuple(src.shape[1:]), float("-inf"))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)


def maybe_num_nodes(edge_index: Tensor, num_nodes: Optional[int] = None) -> int:
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


# --------------------------------------------------------------------------- #
# A.1  gcn_norm  (reference: train/train_clustering.py:37-42,58-63; inside
#      GCNConv for model/hscn.py:88-93 with add_self_loops=False)
# --------------------------------------------------------------------------- #
def add_remaining_self_loops(
    edge_index: Tensor,
    edge_attr: Optional[Tensor],
    fill_value: float,
    num_nodes: int,
) -> Tuple[Tensor, Optional[Tensor]]:
    """PyG ``utils.add_remaining_self_loops``: existing self loops leave the
    main list, keep their weight and move to the tail block ``arange(N)``;
    missing ones get ``fill_value``."""
    N = num_nodes
    mask = edge_index[0] != edge_index[1]
    loop_index = torch.arange(0, N, dtype=torch.long, device=edge_index.device)
    loop_index = loop_index.unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        loop_attr = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fill_value)
        inv_mask = ~mask
        loop_attr[edge_index[0][inv_mask]] = edge_attr[inv_mask]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    edge_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
    return edge_index, edge_attr


def gcn_norm(
    edge_index: Tensor,
    edge_weight: Optional[Tensor] = None,
    num_nodes: Optional[int] = None,
    improved: bool = False,
    add_self_loops: bool = True,
    dtype: torch.dtype = torch.float32,
) -> Tuple[Tensor, Tensor]:
    fill_value = 2.0 if improved else 1.0
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=dtype, device=edge_index.device)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(
            edge_index, edge_weight, fill_value, num_nodes
        )
    row, col = edge_index[0], edge_index[1]
    deg = scatter_add(edge_weight, col, num_nodes)
    deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt = deg_inv_sqrt.masked_fill(deg_inv_sqrt == float("inf"), 0.0)
    return edge_index, deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]


# --------------------------------------------------------------------------- #
# PyG ``Linear`` initialisers
# --------------------------------------------------------------------------- #
def glorot_(t: Tensor) -> Tensor:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-a, a)


class PygLinear(nn.Linear):
    """PyG ``Linear``: same math as ``nn.Linear``; 'glorot' weight option."""

    def __init__(self, in_f: int, out_f: int, bias: bool = True, weight_initializer: str | None = None):
        super().__init__(in_f, out_f, bias=bias)
        if weight_initializer == "glorot":
            glorot_(self.weight)


# --------------------------------------------------------------------------- #
# A.2  GraphConv  (reference: model/hscn.py:32,40)
# --------------------------------------------------------------------------- #
class GraphConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.lin_rel = PygLinear(in_channels, out_channels, bias=bias)
        self.lin_root = PygLinear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        row, col = edge_index[0], edge_index[1]
        x_j = x.index_select(0, row)
        msg = x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j
        agg = scatter_add(msg, col, x.size(0))
        return self.lin_rel(agg) + self.lin_root(x)


# --------------------------------------------------------------------------- #
# A.5  GCNConv(-1, H, add_self_loops=False, cached=False)
#      (reference: model/hscn.py:88-93 via build_conv_relation :117-125)
# --------------------------------------------------------------------------- #
class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, add_self_loops: bool = True, cached: bool = False):
        super().__init__()
        self.add_self_loops = add_self_loops
        self.lin = PygLinear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        edge_index, edge_weight = gcn_norm(
            edge_index, edge_weight, x.size(0), False, self.add_self_loops, x.dtype
        )
        x = self.lin(x)
        row, col = edge_index[0], edge_index[1]
        msg = edge_weight.view(-1, 1) * x.index_select(0, row)
        out = scatter_add(msg, col, x.size(0))
        return out + self.bias


# --------------------------------------------------------------------------- #
# A.6  GATConv((-1,-1), H, heads=1, add_self_loops=False)  bipartite
#      (reference: model/hscn.py:85-87)
# --------------------------------------------------------------------------- #
def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """PyG ``utils.softmax``: max is detached, denominator gets +1e-16."""
    src_max = scatter_max(src.detach(), index, num_nodes)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = scatter_add(out, index, num_nodes) + 1e-16
    return out / out_sum.index_select(0, index)


class GATConv(nn.Module):
    def __init__(self, in_channels: Tuple[int, int], out_channels: int, negative_slope: float = 0.2,
                 add_self_loops: bool = False, cached: bool = False):
        super().__init__()
        assert not add_self_loops, "the hot path only uses add_self_loops=False"
        self.negative_slope = negative_slope
        self.out_channels = out_channels
        self.lin_src = PygLinear(in_channels[0], out_channels, bias=False, weight_initializer="glorot")
        self.lin_dst = PygLinear(in_channels[1], out_channels, bias=False, weight_initializer="glorot")
        self.att_src = nn.Parameter(glorot_(torch.empty(1, 1, out_channels)))
        self.att_dst = nn.Parameter(glorot_(torch.empty(1, 1, out_channels)))
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x: Tuple[Tensor, Tensor], edge_index: Tensor) -> Tensor:
        x_src, x_dst = x
        H, C = 1, self.out_channels
        h_src = self.lin_src(x_src).view(-1, H, C)
        h_dst = self.lin_dst(x_dst).view(-1, H, C)
        alpha_src = (h_src * self.att_src).sum(dim=-1)      # [N_src, 1]
        alpha_dst = (h_dst * self.att_dst).sum(dim=-1)      # [N_dst, 1]
        row, col = edge_index[0], edge_index[1]
        alpha = alpha_src.index_select(0, row) + alpha_dst.index_select(0, col)
        alpha = F.leaky_relu(alpha, self.negative_slope)
        alpha = segment_softmax(alpha, col, x_dst.size(0))  # [E, 1]
        msg = alpha.unsqueeze(-1) * h_src.index_select(0, row)  # [E,1,C]
        out = scatter_add(msg, col, x_dst.size(0)).view(-1, H * C)
        return out + self.bias


# --------------------------------------------------------------------------- #
# A.8  HeteroConv(aggr="sum")   (reference: model/hscn.py:83-96,109)
# --------------------------------------------------------------------------- #
class HeteroConv(nn.Module):
    def __init__(self, convs: dict, aggr: str = "sum"):
        super().__init__()
        assert aggr == "sum"
        self.convs = nn.ModuleDict({"__".join(k): v for k, v in convs.items()})

    def forward(self, x_dict: dict, edge_index_dict: dict) -> dict:
        out: dict = {}
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
            out.setdefault(dst, []).append(o)
        return {k: (v[0] if len(v) == 1 else torch.stack(v, dim=0).sum(dim=0)) for k, v in out.items()}


# --------------------------------------------------------------------------- #
# A.3  to_dense_adj  (reference: model/hscn.py:61)
# --------------------------------------------------------------------------- #
def to_dense_adj(edge_index: Tensor, max_num_nodes: Optional[int] = None) -> Tensor:
    N = maybe_num_nodes(edge_index, max_num_nodes)
    adj = torch.zeros(N * N, dtype=torch.float32, device=edge_index.device)
    idx = edge_index[0] * N + edge_index[1]
    adj.index_add_(0, idx, torch.ones(idx.numel(), dtype=torch.float32, device=edge_index.device))
    return adj.view(1, N, N)


# --------------------------------------------------------------------------- #
# A.4  dense_mincut_pool  (reference: model/hscn.py:63)
# --------------------------------------------------------------------------- #
def dense_mincut_pool(x: Tensor, adj: Tensor, s: Tensor):
    x = x.unsqueeze(0) if x.dim() == 2 else x
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    s = s.unsqueeze(0) if s.dim() == 2 else s
    k = s.size(-1)
    s = torch.softmax(s, dim=-1)
    out = torch.matmul(s.transpose(1, 2), x)
    out_adj = torch.matmul(torch.matmul(s.transpose(1, 2), adj), s)
    mincut_num = torch.einsum("ijj->i", out_adj)
    d_flat = torch.einsum("ijk->ij", adj)
    d = torch.diag_embed(d_flat)
    mincut_den = torch.einsum("ijj->i", torch.matmul(torch.matmul(s.transpose(1, 2), d), s))
    mincut_loss = torch.mean(-(mincut_num / mincut_den))
    ss = torch.matmul(s.transpose(1, 2), s)
    i_s = torch.eye(k).type_as(ss)
    ortho_loss = torch.norm(
        ss / torch.norm(ss, dim=(-1, -2), keepdim=True) - i_s / torch.norm(i_s),
        dim=(-1, -2),
    )
    ortho_loss = torch.mean(ortho_loss)
    EPS = 1e-15
    ind = torch.arange(k, device=out_adj.device)
    out_adj = out_adj.clone()
    out_adj[:, ind, ind] = 0
    dd = torch.einsum("ijk->ij", out_adj)
    dd = torch.sqrt(dd)[:, None] + EPS
    out_adj = (out_adj / dd) / dd.transpose(1, 2)
    return out, out_adj, mincut_loss, ortho_loss


# --------------------------------------------------------------------------- #
# A.7  global_mean_pool  (reference: model/hscn.py:111; mpnn.py:60 scatter_mean)
# --------------------------------------------------------------------------- #
def global_mean_pool(x: Tensor, batch: Tensor, size: Optional[int] = None) -> Tensor:
    B = int(batch.max()) + 1 if size is None else size
    s = scatter_add(x, batch, B)
    cnt = scatter_add(torch.ones(batch.numel(), dtype=x.dtype), batch, B).clamp_(min=1)
    return s / cnt.view(-1, 1)
