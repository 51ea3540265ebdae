"""Pooling / coarsening functions with the torch_geometric signatures the
reference imports (model/hscn.py:6-14; train/train_clustering.py:6)."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .. import _hip
from .._hip import call, ptr, stream
from ..structure import CSR, Relation, adopt_relation, build_csr, relation_of, segments_from_batch
from . import functional as Fh


def global_mean_pool(x: Tensor, batch: Optional[Tensor], size: Optional[int] = None,
                     ptr_: Optional[Tensor] = None) -> Tensor:
    """PyG global_mean_pool (model/hscn.py:111).  ``size=None`` reads
    ``batch.max()+1`` from the device (one sync), like the reference."""
    if batch is None:
        batch = torch.zeros(x.size(0), dtype=torch.int64, device=x.device)
        size = 1
    seg = segments_from_batch(batch, size)
    return Fh.SegmentMeanFn.apply(x, seg, batch.contiguous(), False)


def to_dense_adj(edge_index: Tensor, max_num_nodes: Optional[int] = None) -> Tensor:
    """PyG to_dense_adj without batch / edge_attr (model/hscn.py:61):
    ``adj[0, row, col] += 1`` -> ``[1, N, N]``.  ``N`` defaults to
    ``edge_index.max()+1`` (one sync, as in PyG)."""
    N = int(edge_index.max().item()) + 1 if max_num_nodes is None else int(max_num_nodes)
    adj = torch.zeros(N, N, dtype=torch.float32, device=edge_index.device)
    E = edge_index.size(1)
    call("hscn_to_dense_adj", ptr(edge_index[0].contiguous()), ptr(edge_index[1].contiguous()), E, N,
         ptr(adj), stream())
    return adj.view(1, N, N)


def to_dense_adj_batched(edge_index: Tensor, num_graphs: int, nodes_per_graph: int) -> Tensor:
    """``to_dense_adj(edge_index, batch)`` for a block-diagonal batch of equally sized graphs: ``[B, n, n]``."""
    B, n = int(num_graphs), int(nodes_per_graph)
    adj = torch.zeros(B, n, n, dtype=torch.float32, device=edge_index.device)
    call("hscn_to_dense_adj_batched", ptr(edge_index[0].contiguous()), ptr(edge_index[1].contiguous()),
         edge_index.size(1), B, n, ptr(adj), stream())
    return adj


def to_dense_adj_ragged(edge_index: Tensor, nptr: Tensor, gid: Tensor, num_graphs: int, max_nodes: int,
                        raw: bool = False, as_bytes: bool = False, flag: Optional[Tensor] = None,
                        symmetry: bool = False):
    """``to_dense_adj(edge_index, batch)`` for a block-diagonal batch of graphs of different sizes: ``[B, nmax, nmax]``,
    zero beyond a graph's own nodes.  ``raw=True``: ``edge_index`` is the RAW edge list and the result is what
    ``gcn_norm(add_self_loops=True)`` followed by ``to_dense_adj`` gives -- off-diagonal counts plus the identity
    (train/train_clustering.py:37-42, model/hscn.py:61) -- without building the self-looped list first."""
    B, n = int(num_graphs), int(max_nodes)
    if as_bytes:
        # the counts as bytes, rows padded to a multiple of 32: what the batched dense route streams through the
        # matrix cores (a quarter of the float adjacency's bytes; exact up to 255 parallel edges, flag bit 16 beyond)
        # symmetry=True (as_bytes only): also returns asym int32 [B] -- 1 where a graph's adjacency is NOT symmetric
        # (hscn_dense_adj_asymmetry_u8); the flags live behind the adjacency in the same zero-filled allocation
        lda = (n + 31) // 32 * 32
        nb = B * n * lda
        buf = torch.zeros(nb + (4 * B if symmetry else 0), dtype=torch.uint8, device=edge_index.device)
        adj8 = buf[:nb].view(B, n, lda)
        call("hscn_to_dense_adj_ragged_u8", ptr(edge_index[0].contiguous()), ptr(edge_index[1].contiguous()),
             edge_index.size(1), ptr(nptr), ptr(gid), int(gid.numel()), B, n, 1 if raw else 0, ptr(adj8), ptr(flag), stream())
        if symmetry:
            asym = buf[nb:].view(torch.int32)
            call("hscn_dense_adj_asymmetry_u8", ptr(adj8), B, n, ptr(asym), stream())
            return adj8, asym
        return adj8
    adj = torch.zeros(B, n, n, dtype=torch.float32, device=edge_index.device)
    call("hscn_to_dense_adj_ragged", ptr(edge_index[0].contiguous()), ptr(edge_index[1].contiguous()),
         edge_index.size(1), ptr(nptr), ptr(gid), int(gid.numel()), B, n, 1 if raw else 0, ptr(adj), stream())
    return adj


def gcn_norm_static(edge_index: Tensor, edge_weight: Optional[Tensor] = None, num_nodes: Optional[int] = None,
                    improved: bool = False) -> Tuple[Tensor, Tensor]:
    """``gcn_norm(add_self_loops=True)`` with a STATIC output shape ``[2, E + N]`` (no boolean-mask indexing, no
    data-dependent size: the whole of it can sit inside a hipGraph capture).  The E input edges keep their slots -- an
    input self loop stays in place with weight 0 and hands its weight to the node's loop -- then one loop per node.
    Degrees, normalised weights and every aggregation over this list equal PyG's over its (shorter) list: a
    zero-weight edge adds nothing to a degree and +0 to a sum, at the position the removed edge had.  What differs
    is the list's LENGTH when the input has self loops: ``gcn_norm`` (above) stays the drop-in for callers that look
    at the returned ``edge_index``; stage A's layered route uses this one."""
    if num_nodes is None:
        raise ValueError("gcn_norm_static needs num_nodes (reading edge_index.max() would synchronise)")
    N, E = int(num_nodes), int(edge_index.size(1))
    dev = edge_index.device
    ei = torch.empty(2, E + N, dtype=torch.int64, device=dev)
    w_in = torch.empty(E + N, dtype=torch.float32, device=dev)
    call("hscn_gcn_norm_self_loops", ptr(edge_index[0].contiguous()), ptr(edge_index[1].contiguous()),
         ptr(edge_weight.contiguous()) if edge_weight is not None else None, E, N, 2.0 if improved else 1.0,
         ptr(ei[0]), ptr(ei[1]), ptr(w_in), stream())
    csr = build_csr(ei[1], ei[0], N, N)
    dinv = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
    w = torch.empty(max(E + N, 1), dtype=torch.float32, device=dev)
    call("hscn_gcn_norm_weights", ptr(csr.rowptr), ptr(csr.col), ptr(csr.eid), ptr(w_in), N, ptr(dinv), ptr(w), stream())
    adopt_relation(ei, N, N, csr)      # the GraphConv that receives `ei` walks this very CSR: no second build
    return ei, w[: E + N]


def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor] = None, num_nodes: Optional[int] = None,
             improved: bool = False, add_self_loops: bool = True,
             dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor]:
    """PyG gcn_norm (train/train_clustering.py:37-42).

    The self-loop bookkeeping (existing loops move to the tail block keeping
    their weight, missing ones are appended with weight 1) is index plumbing on
    the device; degrees and the symmetric normalisation run in
    ``hscn_gcn_norm_weights`` with the reference's summation order."""
    if num_nodes is None:
        num_nodes = int(edge_index.max().item()) + 1
    N = int(num_nodes)
    dev = edge_index.device
    fill = 2.0 if improved else 1.0
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=dev)
    if add_self_loops:
        mask = edge_index[0] != edge_index[1]
        loop_attr = torch.full((N,), fill, dtype=edge_weight.dtype, device=dev)
        inv = ~mask
        loop_attr[edge_index[0][inv]] = edge_weight[inv]
        loops = torch.arange(N, dtype=torch.int64, device=dev)
        edge_index = torch.cat([edge_index[:, mask], loops.unsqueeze(0).repeat(2, 1)], 1).contiguous()
        edge_weight = torch.cat([edge_weight[mask], loop_attr], 0).contiguous()
    csr = build_csr(edge_index[1], edge_index[0], N, N)
    dinv = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
    w = torch.empty(max(edge_index.size(1), 1), dtype=torch.float32, device=dev)
    call("hscn_gcn_norm_weights", ptr(csr.rowptr), ptr(csr.col), ptr(csr.eid), ptr(edge_weight.contiguous()),
         N, ptr(dinv), ptr(w), stream())
    return edge_index, w[: edge_index.size(1)]


def mincut_pool_sparse(x: Optional[Tensor], edge_index: Union[Tensor, Relation], s: Tensor,
                       node_ptr: Optional[Tensor] = None):
    """dense_mincut_pool semantics evaluated on the edge list:
    ``A = sum_e E[row_e, col_e]`` is never densified.  ``node_ptr`` (int32
    ``[G+1]``) splits a block-diagonal batch into graphs; losses are the mean over
    graphs, exactly dense_mincut_pool's mean over its batch dimension.

    Returns ``(S, pooled_x [G,K,F], pooled_adj [G,K,K], mincut_loss, ortho_loss)``."""
    n = s.size(0)
    rel = edge_index if isinstance(edge_index, Relation) else relation_of(edge_index, n, n)
    if node_ptr is None:
        node_ptr = torch.tensor([0, n], dtype=torch.int32, device=s.device)
    G = int(node_ptr.numel()) - 1
    S, mc, o, px, padj = Fh.MinCutSparseFn.apply(s, x, rel, node_ptr.to(torch.int32).contiguous(), G)
    return S, px, padj, mc, o


def dense_mincut_pool(x: Tensor, adj: Tensor, s: Tensor, mask: Optional[Tensor] = None):
    """torch_geometric.nn.dense_mincut_pool (reference model/hscn.py:63) on a
    DENSE adjacency ``[B,n,n]`` (2-D inputs are treated as batch 1): the cluster-assignment
    contractions run on the matrix cores with exact-fp32 MFMA (csrc/dense.hip).
    Returns ``(out [B,K,F], out_adj [B,K,K], mincut_loss, ortho_loss)``; the losses carry
    gradients to ``s``."""
    if mask is not None:
        raise NotImplementedError("the hot path never passes a mask (hscn.py:63)")
    x = x.unsqueeze(0) if x.dim() == 2 else x
    adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
    s = s.unsqueeze(0) if s.dim() == 2 else s
    S, mc, o, px, padj = Fh.MinCutDenseFn.apply(s, x, adj)
    return px, padj, mc, o
