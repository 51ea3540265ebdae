"""SignNet positional-encoding encoder with the reference's class names and signatures
(/root/reference/graph_hscn/encoder/signnet.py:11-381): ``MLP`` / ``GIN`` / ``GINDeepSigns`` / ``MaskedGINDeepSigns``
/ ``SignNetNodeEncoder``.  In the reference it runs ONCE per batch of the dataset, under ``no_grad``, with randomly
initialised weights (train/train.py:29-51) -- pre-processing, SURVEY.md section 8(f)4 -- so this is a mirror of the
interface on top of the library's operators, not a fused kernel:

* ``GINConv``: sum aggregation = the CSR gather-reduce of csrc/spmm.hip with unit weights (``hscn_spmm_csr_weighted``),
  for the 3-D eigenvector tensors ``[K, N, C]`` one pass over node rows of width K * C;
* ``Linear`` / ``BatchNorm1d`` / ``LayerNorm`` / activation / dropout: the library's (csrc/linear.hip, norm.hip, ...);
* the sign-invariant sum ``enc(x) + enc(-x)``, the DeepSet mask and the concatenation are tensor plumbing.

One deviation from the reference, without which nothing here could be constructed: ``MLP`` takes its activation from
``ACT_DICT[activation]``; the reference reads ``ACT_DICT["activation"]`` (signnet.py:49), a KeyError.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor
from torch.autograd import Function

from ..config.config import ACT_DICT
from ..nn import BatchNorm1d, LayerNorm, Linear
from ..nn import functional as Fh
from ..structure import relation_of


class _SumAggFn(Function):
    """out_i = sum over edges (j -> i) of x_j on [N, W] rows (edge order, as torch_scatter's CPU sum)."""

    @staticmethod
    def forward(ctx, x: Tensor, rel):
        ctx.rel = rel
        return Fh.spmm_weighted_raw(rel.csr, None, x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return Fh.spmm_weighted_raw(ctx.rel.csr_t, None, g.contiguous()), None


def _flat2(x: Tensor):
    """[..., C] -> ([rows, C], restore)."""
    if x.dim() == 2:
        return x, lambda y: y
    lead = x.shape[:-1]
    return x.reshape(-1, x.shape[-1]), lambda y: y.reshape(*lead, y.shape[-1])


class GINConv(nn.Module):
    """torch_geometric.nn.GINConv(nn, eps=0, train_eps=False): ``nn((1 + eps) * x_i + sum_{j -> i} x_j)``; ``x`` may be
    [N, C] or [K, N, C] (node dimension -2, as PyG propagates)."""

    def __init__(self, net: nn.Module, eps: float = 0.0):
        super().__init__()
        self.nn = net
        self.eps = float(eps)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        N = x.shape[-2]
        rel = relation_of(edge_index, N, N, both=torch.is_grad_enabled() and x.requires_grad)
        if x.dim() == 3:
            K, _, C = x.shape
            rows = x.permute(1, 0, 2).reshape(N, K * C)
            agg = _SumAggFn.apply(rows, rel).reshape(N, K, C).permute(1, 0, 2)
        else:
            agg = _SumAggFn.apply(x, rel)
        out = agg + x if self.eps == 0.0 else agg + (1.0 + self.eps) * x
        return self.nn(out)


class MLP(nn.Module):  # signnet.py:11-82
    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, num_layers: int, use_bn: bool = False,
                 use_ln: bool = False, dropout: float = 0.5, activation: str = "relu", residual: bool = False) -> None:
        super().__init__()
        self.fcs = nn.ModuleList()
        if use_bn:
            self.bns = nn.ModuleList()
        if use_ln:
            self.lns = nn.ModuleList()
        mk = lambda i, o: Linear(i, o)      # noqa: E731  (torch.nn.Linear's default initialisation, parameter names weight / bias)
        if num_layers == 1:
            self.fcs.append(mk(in_channels, out_channels))
        else:
            self.fcs.append(mk(in_channels, hidden_channels))
            if use_bn:
                self.bns.append(BatchNorm1d(hidden_channels))
            if use_ln:
                self.lns.append(LayerNorm(hidden_channels))
            for _ in range(num_layers - 2):
                self.fcs.append(mk(hidden_channels, hidden_channels))
                if use_bn:
                    self.bns.append(BatchNorm1d(hidden_channels))
                if use_ln:
                    self.lns.append(LayerNorm(hidden_channels))
            self.fcs.append(mk(hidden_channels, out_channels))
        self.activation = ACT_DICT[activation.lower()]      # (reference: ACT_DICT["activation"], a KeyError)
        self.dropout = dropout
        self.use_bn, self.use_ln, self.residual = use_bn, use_ln, residual

    def forward(self, x: Tensor) -> Tensor:
        x_prev = x
        for i, fc in enumerate(list(self.fcs)[:-1]):
            x2, back = _flat2(x)
            x2 = self.activation(fc(x2))
            if self.use_bn:
                if x.dim() not in (2, 3):
                    raise ValueError("Invalid dimension of x")
                x2 = self.bns[i](x2)          # [K, N, C]: BatchNorm1d over the channel axis of (K, C, N) == over rows of [K N, C]
            if self.use_ln:
                x2 = self.lns[i](x2)
            x = back(x2)
            if self.residual and x_prev.shape == x.shape:
                x = x + x_prev
            x = Fh.dropout(x, p=self.dropout, training=self.training)
            x_prev = x
        x2, back = _flat2(x)
        x = back(self.fcs[-1](x2))
        if self.residual and x_prev.shape == x.shape:
            x = x + x_prev
        return x


class GIN(nn.Module):  # signnet.py:85-161
    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, n_layers: int, use_bn: bool = True,
                 dropout: float = 0.5, activation: str = "relu") -> None:
        super().__init__()
        self.layers = nn.ModuleList()
        if use_bn:
            self.bns = nn.ModuleList()
        self.use_bn, self.dropout = use_bn, dropout
        self.layers.append(GINConv(MLP(in_channels, hidden_channels, hidden_channels, 1, use_bn=use_bn, dropout=dropout,
                                       activation=activation)))
        for _ in range(n_layers - 2):
            self.layers.append(GINConv(MLP(hidden_channels, hidden_channels, hidden_channels, 1, use_bn=use_bn,
                                           dropout=dropout, activation=activation)))
            if use_bn:
                self.bns.append(BatchNorm1d(hidden_channels))
        self.layers.append(GINConv(MLP(hidden_channels, hidden_channels, out_channels, 2, use_bn=use_bn, dropout=dropout,
                                       activation=activation)))
        if use_bn:
            self.bns.append(BatchNorm1d(hidden_channels))

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        for i, layer in enumerate(self.layers):
            if i != 0:
                x = Fh.dropout(x, p=self.dropout, training=self.training)
                if self.use_bn:
                    if x.dim() not in (2, 3):
                        raise ValueError("Invalid x dim.")
                    x2, back = _flat2(x)
                    x = back(self.bns[i - 1](x2))
            x = layer(x, edge_index)
        return x


class GINDeepSigns(nn.Module):  # signnet.py:164-218
    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, num_layers: int, k: int, dim_pe: int,
                 rho_num_layers: int, use_bn: bool = False, use_ln: bool = False, dropout: float = 0.5,
                 activation: str = "relu") -> None:
        super().__init__()
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)
        self.rho = MLP(out_channels * k, hidden_channels, dim_pe, rho_num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)

    def forward(self, x: Tensor, edge_index: Tensor, batch_index: Tensor) -> Tensor:
        N = x.shape[0]
        x = x.transpose(0, 1).contiguous()                           # n x k x in -> k x n x in
        x = self.enc(x, edge_index) + self.enc(-x, edge_index)       # sign invariance
        x = x.transpose(0, 1).reshape(N, -1)                         # k x n x out -> n x (k * out)
        return self.rho(x)


class MaskedGINDeepSigns(nn.Module):  # signnet.py:221-293
    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, num_layers: int, dim_pe: int,
                 rho_num_layers: int, use_bn: bool = False, use_ln: bool = False, dropout: float = 0.5,
                 activation: str = "relu") -> None:
        super().__init__()
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)
        self.rho = MLP(out_channels, hidden_channels, dim_pe, rho_num_layers, use_bn=use_bn, dropout=dropout,
                       activation=activation)

    def batched_n_nodes(self, batch_index: Tensor) -> Tensor:
        """Number of nodes of its own graph, for every node (signnet.py:257-267)."""
        n_nodes = torch.bincount(batch_index)
        return n_nodes[batch_index]

    def forward(self, x: Tensor, edge_index: Tensor, batch_index: Tensor) -> Tensor:
        K = x.shape[1]
        x = x.transpose(0, 1).contiguous()                           # N x K x 1 -> K x N x 1
        x = self.enc(x, edge_index) + self.enc(-x, edge_index)       # K x N x out
        x = x.transpose(0, 1)                                        # N x K x out
        # frequencies beyond a graph's node count do not exist (they are the NaN padding): zero them, then sum over K
        mask = torch.arange(K, device=x.device).unsqueeze(0) < self.batched_n_nodes(batch_index).unsqueeze(1)
        x = (x * mask.unsqueeze(-1).to(x.dtype)).sum(dim=1)
        return self.rho(x)


class SignNetNodeEncoder(nn.Module):  # signnet.py:296-381
    def __init__(self, cfg, dim_in: int, dim_emb: int, expand_x: bool = True) -> None:
        super().__init__()
        dim_pe = cfg.dim_pe
        model_type = cfg.model
        if model_type not in ["MLP", "DeepSet"]:
            raise ValueError(f"Unexpected SignNet model {model_type}")
        self.model_type = model_type
        if cfg.post_layers < 1:
            raise ValueError("Num layers in rho model has to be positive.")
        self.pass_as_var = cfg.pass_as_var
        if dim_emb - dim_pe < 1:
            raise ValueError(f"SignNet PE size {dim_pe} is too large for desired embedding size of {dim_emb}.")
        if expand_x:
            self.linear_x = Linear(dim_in, dim_emb - dim_pe)
        self.expand_x = expand_x
        common = dict(in_channels=1, hidden_channels=cfg.phi_hidden_dim, out_channels=cfg.phi_out_dim,
                      num_layers=cfg.layers, dim_pe=dim_pe, rho_num_layers=cfg.post_layers, use_bn=cfg.use_bn,
                      dropout=0.0, activation="relu")
        if model_type == "MLP":
            self.sign_inv_net = GINDeepSigns(k=cfg.eigen_max_freqs, **common)
        else:
            self.sign_inv_net = MaskedGINDeepSigns(**common)

    def forward(self, batch):
        if not (hasattr(batch, "eigvals_sn") and hasattr(batch, "eigvecs_sn")) or batch.eigvecs_sn is None:
            raise ValueError("Precomputed eigen values and vectors are required for SignNetNodeEncoder; "
                             "set config 'posenc_SignNet.enable' to True")
        pos_enc = torch.nan_to_num(batch.eigvecs_sn.unsqueeze(-1).float(), nan=0.0)     # NaN padding -> 0 (signnet.py:361-363)
        pos_enc = self.sign_inv_net(pos_enc, batch.edge_index, batch.batch)
        h = self.linear_x(batch.x.to(torch.float32)) if self.expand_x else batch.x
        batch.x = torch.cat((h, pos_enc), 1)
        if self.pass_as_var:
            batch.pe_SignNet = pos_enc
        return batch
