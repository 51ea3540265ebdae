"""CPU restatement of the SignNet positional-encoding path -- TEST INFRASTRUCTURE (imported by tests/ only).

Follows /root/reference/graph_hscn/encoder/signnet.py:11-381 and transform/posenc.py:14-107 in plain torch / numpy, with
the un-vendored PyG pieces written out from their published semantics (PyG 2.2/2.3):
  * ``GINConv(nn, eps=0)``: ``nn(x_i + sum_{j -> i} x_j)``, propagating along the node dimension -2 (so a [K, N, C]
    input is aggregated per frequency), gather -> ``index_add_`` in edge order;
  * ``get_laplacian(edge_index, normalization)``: self loops removed, unit weights; None: D - A, "sym": I - D^-1/2 A D^-1/2,
    "rw": I - D^-1 A;  ``to_undirected``: both directions, coalesced.
The reference's ``MLP`` cannot be constructed (``ACT_DICT["activation"]``, signnet.py:49); as in the
product the activation is looked up by its NAME.  Parity unpinned, like the rest of oracle/: the reference holds no
vectors for this path."""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

ACT = {"relu": torch.relu, "elu": F.elu, "tanh": torch.tanh, "identity": lambda t: t}


def gin_aggregate(x: Tensor, edge_index: Tensor) -> Tensor:
    src, dst = edge_index
    msg = x.index_select(-2, src)
    return torch.zeros_like(x).index_add_(-2, dst, msg)


class GINConv(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.nn = net

    def forward(self, x, edge_index):
        return self.nn(gin_aggregate(x, edge_index) + x)


class MLP(nn.Module):  # signnet.py:11-82
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, use_bn=False, use_ln=False, dropout=0.5,
                 activation="relu", residual=False):
        super().__init__()
        self.fcs = nn.ModuleList()
        if use_bn:
            self.bns = nn.ModuleList()
        if use_ln:
            self.lns = nn.ModuleList()
        if num_layers == 1:
            self.fcs.append(nn.Linear(in_channels, out_channels))
        else:
            self.fcs.append(nn.Linear(in_channels, hidden_channels))
            if use_bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            if use_ln:
                self.lns.append(nn.LayerNorm(hidden_channels))
            for _ in range(num_layers - 2):
                self.fcs.append(nn.Linear(hidden_channels, hidden_channels))
                if use_bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
                if use_ln:
                    self.lns.append(nn.LayerNorm(hidden_channels))
            self.fcs.append(nn.Linear(hidden_channels, out_channels))
        self.activation = ACT[activation]
        self.dropout, self.use_bn, self.use_ln, self.residual = dropout, use_bn, use_ln, residual

    def forward(self, x):
        x_prev = x
        for i, fc in enumerate(self.fcs[:-1]):
            x = self.activation(fc(x))
            if self.use_bn:
                x = self.bns[i](x) if x.ndim == 2 else self.bns[i](x.transpose(2, 1)).transpose(2, 1)
            if self.use_ln:
                x = self.lns[i](x)
            if self.residual and x_prev.shape == x.shape:
                x = x + x_prev
            x = F.dropout(x, p=self.dropout, training=self.training)
            x_prev = x
        x = self.fcs[-1](x)
        if self.residual and x_prev.shape == x.shape:
            x = x + x_prev
        return x


class GIN(nn.Module):  # signnet.py:85-161
    def __init__(self, in_channels, hidden_channels, out_channels, n_layers, use_bn=True, dropout=0.5, activation="relu"):
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
                self.bns.append(nn.BatchNorm1d(hidden_channels))
        self.layers.append(GINConv(MLP(hidden_channels, hidden_channels, out_channels, 2, use_bn=use_bn, dropout=dropout,
                                       activation=activation)))
        if use_bn:
            self.bns.append(nn.BatchNorm1d(hidden_channels))

    def forward(self, x, edge_index):
        for i, layer in enumerate(self.layers):
            if i != 0:
                x = F.dropout(x, p=self.dropout, training=self.training)
                if self.use_bn:
                    x = self.bns[i - 1](x) if x.ndim == 2 else self.bns[i - 1](x.transpose(2, 1)).transpose(2, 1)
            x = layer(x, edge_index)
        return x


class GINDeepSigns(nn.Module):  # signnet.py:164-218
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, k, dim_pe, rho_num_layers, use_bn=False,
                 dropout=0.5, activation="relu"):
        super().__init__()
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout, activation=activation)
        self.rho = MLP(out_channels * k, hidden_channels, dim_pe, rho_num_layers, use_bn=use_bn, dropout=dropout, activation=activation)

    def forward(self, x, edge_index, batch_index):
        N = x.shape[0]
        x = x.transpose(0, 1)
        x = self.enc(x, edge_index) + self.enc(-x, edge_index)
        return self.rho(x.transpose(0, 1).reshape(N, -1))


class MaskedGINDeepSigns(nn.Module):  # signnet.py:221-293
    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dim_pe, rho_num_layers, use_bn=False,
                 dropout=0.5, activation="relu"):
        super().__init__()
        self.enc = GIN(in_channels, hidden_channels, out_channels, num_layers, use_bn=use_bn, dropout=dropout, activation=activation)
        self.rho = MLP(out_channels, hidden_channels, dim_pe, rho_num_layers, use_bn=use_bn, dropout=dropout, activation=activation)

    def forward(self, x, edge_index, batch_index):
        N, K = x.shape[0], x.shape[1]
        x = x.transpose(0, 1)
        x = self.enc(x, edge_index) + self.enc(-x, edge_index)
        x = x.transpose(0, 1).clone()
        counts = torch.bincount(batch_index)
        per_node = counts[batch_index]                                   # signnet.py:257-267
        mask = torch.arange(K).unsqueeze(0).expand(N, K) < per_node.unsqueeze(1)
        x[~mask] = 0
        return self.rho(x.sum(dim=1))


class SignNetNodeEncoder(nn.Module):  # signnet.py:296-381
    def __init__(self, cfg, dim_in, dim_emb, expand_x=True):
        super().__init__()
        if expand_x:
            self.linear_x = nn.Linear(dim_in, dim_emb - cfg.dim_pe)
        self.expand_x, self.pass_as_var = expand_x, cfg.pass_as_var
        kw = dict(in_channels=1, hidden_channels=cfg.phi_hidden_dim, out_channels=cfg.phi_out_dim, num_layers=cfg.layers,
                  dim_pe=cfg.dim_pe, rho_num_layers=cfg.post_layers, use_bn=cfg.use_bn, dropout=0.0, activation="relu")
        self.sign_inv_net = GINDeepSigns(k=cfg.eigen_max_freqs, **kw) if cfg.model == "MLP" else MaskedGINDeepSigns(**kw)

    def forward(self, x, eigvecs_sn, edge_index, batch):
        pos_enc = eigvecs_sn.unsqueeze(-1).clone()
        pos_enc[torch.isnan(pos_enc)] = 0
        pos_enc = self.sign_inv_net(pos_enc, edge_index, batch)
        h = self.linear_x(x.to(torch.float32)) if self.expand_x else x
        return torch.cat((h, pos_enc), 1), pos_enc


# ---- transform/posenc.py ------------------------------------------------------------------------------------------------
def laplacian_dense(edge_index: Tensor, N: int, normalization: Optional[str], undirected_input: bool) -> np.ndarray:
    ei = edge_index
    if not undirected_input:
        ei = torch.unique(torch.cat([ei, ei.flip(0)], 1), dim=1)
    A = torch.zeros(N, N, dtype=torch.float64)
    keep = ei[0] != ei[1]
    A.index_put_((ei[0][keep], ei[1][keep]), torch.ones(int(keep.sum()), dtype=torch.float64), accumulate=True)
    deg = A.sum(1)
    if normalization is None:
        L = torch.diag(deg) - A
    elif normalization == "sym":
        dis = deg.pow(-0.5)
        dis[torch.isinf(dis)] = 0
        L = torch.eye(N, dtype=torch.float64) - dis[:, None] * A * dis[None, :]
    else:
        di = 1.0 / deg
        di[torch.isinf(di)] = 0
        L = torch.eye(N, dtype=torch.float64) - di[:, None] * A
    return L.numpy().astype(np.float32)


def posenc_stats(edge_index: Tensor, N: int, max_freqs: int, lap_norm: str = "sym", eigvec_norm: str = "L2",
                 undirected_input: bool = True):
    """posenc.py:14-83 for one graph -> (eigvals [N, max_freqs, 1], eigvecs [N, max_freqs]), NaN padded."""
    L = laplacian_dense(edge_index, N, None if lap_norm.lower() == "none" else lap_norm.lower(), undirected_input)
    evals, evects = np.linalg.eigh(L)
    idx = evals.argsort()[:max_freqs]
    evals, evects = evals[idx], np.real(evects[:, idx])
    ev = torch.from_numpy(np.real(evals)).clamp_min(0)
    vec = torch.from_numpy(evects).float()
    if eigvec_norm == "L1":
        den = vec.norm(p=1, dim=0, keepdim=True)
    elif eigvec_norm == "L2":
        den = vec.norm(p=2, dim=0, keepdim=True)
    else:
        den = vec.abs().max(dim=0, keepdim=True).values
    vec = vec / den.clamp_min(1e-12)
    if N < max_freqs:
        vec = F.pad(vec, (0, max_freqs - N), value=float("nan"))
        ev = F.pad(ev, (0, max_freqs - N), value=float("nan"))
    return ev.unsqueeze(0).repeat(N, 1).unsqueeze(2), vec
