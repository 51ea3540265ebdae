"""Operators of the hot path as autograd Functions over the HIP C ABI.

Every forward/backward here is one or more ``hscn_*`` calls (include/hscn.h) on
the current HIP stream; nothing falls back to eager PyTorch arithmetic.  The
semantics are those of the torch_geometric operators the reference calls
(model/hscn.py:6-14).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from .. import _hip
from .._hip import ACT, call, ptr, stream
from ..structure import CSR, Relation


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"hot-path tensors are float32 (got {t.dtype})")
    return t.contiguous()


# --------------------------------------------------------------------------- #
# raw launchers (no autograd)
# --------------------------------------------------------------------------- #
def linear_raw(x: Tensor, W: Tensor, bias: Optional[Tensor] = None, x2: Optional[Tensor] = None,
               W2: Optional[Tensor] = None, att: Optional[Tensor] = None, act: int = 0,
               w_layout: int = 0) -> Tuple[Tensor, Optional[Tensor]]:
    rows, in_f = x.shape
    out_f = W.shape[0] if w_layout == 0 else W.shape[1]
    y = torch.empty(rows, out_f, dtype=torch.float32, device=x.device)
    a = torch.empty(rows, dtype=torch.float32, device=x.device) if att is not None else None
    call("hscn_linear_fwd", ptr(x), ptr(W), ptr(bias), ptr(x2), ptr(W2), ptr(att), ptr(a), ptr(y),
         rows, in_f, out_f, w_layout, act, stream())
    return y, a


def linear_bwd_w_raw(gy: Tensor, x: Optional[Tensor], want_w: bool = True,
                     want_b: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    rows, out_f = gy.shape
    in_f = x.shape[1] if x is not None else 0
    dev = gy.device
    gW = torch.empty(out_f, in_f, dtype=torch.float32, device=dev) if (want_w and x is not None) else None
    gb = torch.empty(out_f, dtype=torch.float32, device=dev) if want_b else None
    nbytes = int(_hip.lib().hscn_linear_bwd_w_workspace_bytes(rows, in_f, out_f))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
    call("hscn_linear_bwd_w", ptr(gy), ptr(x), ptr(gW), ptr(gb), rows, in_f, out_f, 0, ptr(ws), nbytes,
         stream())
    return gW, gb


def act_bwd_raw(gy: Tensor, y: Tensor, act: int) -> Tensor:
    if act == 0:
        return gy
    g = torch.empty_like(gy)
    call("hscn_act_bwd", ptr(gy), ptr(y), ptr(g), gy.numel(), act, stream())
    return g


def spmm_gcn_raw(csr: CSR, dinv_r: Tensor, dinv_c: Tensor, h: Tensor, bias: Optional[Tensor] = None,
                 act: int = 0) -> Tensor:
    out = torch.empty(csr.num_rows, h.shape[1], dtype=torch.float32, device=h.device)
    call("hscn_spmm_csr_gcn", ptr(csr.rowptr), ptr(csr.col), ptr(dinv_r), ptr(dinv_c), ptr(h), ptr(bias),
         ptr(out), csr.num_rows, h.shape[1], 0, act, stream())
    return out


def spmm_weighted_raw(csr: CSR, w: Optional[Tensor], x: Tensor) -> Tensor:
    out = torch.empty(csr.num_rows, x.shape[1], dtype=torch.float32, device=x.device)
    call("hscn_spmm_csr_weighted", ptr(csr.rowptr), ptr(csr.col), ptr(csr.eid) if w is not None else None,
         ptr(w), ptr(x), ptr(out), csr.num_rows, x.shape[1], stream())
    return out


# --------------------------------------------------------------------------- #
# Activation
# --------------------------------------------------------------------------- #
class ActFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, act: int):
        x = _c(x)
        y = torch.empty_like(x)
        call("hscn_act_fwd", ptr(x), ptr(y), x.numel(), act, stream())
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        (y,) = ctx.saved_tensors
        return act_bwd_raw(_c(g), y, ctx.act), None


class Activation:
    """Callable activation computed on the HIP path; ``hscn_name`` lets modules
    fuse it into the producing kernel's epilogue."""

    def __init__(self, name: str):
        self.hscn_name = name

    def __call__(self, x: Tensor) -> Tensor:
        if self.hscn_name == "identity":
            return x
        return ActFn.apply(x, ACT[self.hscn_name])

    def __repr__(self) -> str:
        return f"Activation({self.hscn_name})"


# --------------------------------------------------------------------------- #
# Dropout (MPNN baseline, reference model/mpnn.py:58)
# --------------------------------------------------------------------------- #
class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, p: float, seed: int):
        x = _c(x)
        y = torch.empty_like(x)
        call("hscn_dropout", ptr(x), ptr(y), x.numel(), float(p), int(seed), stream())
        ctx.p, ctx.seed = float(p), int(seed)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        g = _c(g)
        gx = torch.empty_like(g)
        call("hscn_dropout", ptr(g), ptr(gx), g.numel(), ctx.p, ctx.seed, stream())   # same seed: same mask
        return gx, None, None


_dropout_calls = 0


def dropout(x: Tensor, p: float = 0.5, training: bool = True, seed: Optional[int] = None) -> Tensor:
    """``F.dropout`` semantics (inverted scaling; identity when not training or p == 0).  The mask comes
    from the library's counter-based generator, keyed by ``seed`` (default: ``torch.initial_seed()`` mixed
    with a per-call counter, so ``torch.manual_seed`` makes a run repeatable); it is NOT the mask torch's
    own generator would draw.  The default seed is a host value: a captured hipGraph replays one mask."""
    global _dropout_calls
    if p < 0.0 or p > 1.0:
        raise ValueError(f"dropout probability has to be between 0 and 1, but got {p}")
    if not training or p == 0.0:
        return x
    if p == 1.0:
        return x * 0.0
    if seed is None:
        _dropout_calls += 1
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _dropout_calls) & 0xFFFFFFFFFFFFFFFF
    return DropoutFn.apply(x, p, seed)


# --------------------------------------------------------------------------- #
# Linear
# --------------------------------------------------------------------------- #
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, bias: Optional[Tensor], act: int):
        x, W, bias = _c(x), _c(W), _c(bias)
        y, _ = linear_raw(x, W, bias, act=act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        x, W, y = ctx.saved_tensors
        g = act_bwd_raw(_c(g), y, ctx.act)
        gx = linear_raw(g, W, w_layout=1)[0] if ctx.needs_input_grad[0] else None
        gW, gb = linear_bwd_w_raw(g, x, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return gx, gW, gb, None


def linear(x: Tensor, W: Tensor, bias: Optional[Tensor] = None, act: str = "identity") -> Tensor:
    lead = x.shape[:-1]
    y = LinearFn.apply(x.reshape(-1, x.shape[-1]), W, bias, ACT[act])
    return y.view(*lead, W.shape[0])


# --------------------------------------------------------------------------- #
# GCNConv (unit weights, add_self_loops=False)
# --------------------------------------------------------------------------- #
class GCNConvFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, bias: Optional[Tensor], rel: Relation, act: int):
        x, W, bias = _c(x), _c(W), _c(bias)
        h, _ = linear_raw(x, W)
        out = spmm_gcn_raw(rel.csr, rel.dinv, rel.dinv, h, bias, act)
        ctx.rel, ctx.act, ctx.has_bias = rel, act, bias is not None
        ctx.save_for_backward(x, W, out)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        x, W, out = ctx.saved_tensors
        rel: Relation = ctx.rel
        g = act_bwd_raw(_c(g), out, ctx.act)
        gb = linear_bwd_w_raw(g, None, False, True)[1] if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        gx = gW = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gh = spmm_gcn_raw(rel.csr_t, rel.dinv, rel.dinv, g)
            if ctx.needs_input_grad[1]:
                gW = linear_bwd_w_raw(gh, x, True, False)[0]
            if ctx.needs_input_grad[0]:
                gx = linear_raw(gh, W, w_layout=1)[0]
        return gx, gW, gb, None, None


# --------------------------------------------------------------------------- #
# GraphConv (edge-weighted sum aggregation)
# --------------------------------------------------------------------------- #
class GraphConvFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, edge_weight: Optional[Tensor], W_rel: Tensor, b_rel: Optional[Tensor],
                W_root: Tensor, rel: Relation, act: int):
        x, W_rel, b_rel, W_root = _c(x), _c(W_rel), _c(b_rel), _c(W_root)
        edge_weight = _c(edge_weight)
        agg = spmm_weighted_raw(rel.csr, edge_weight, x)
        y, _ = linear_raw(agg, W_rel, b_rel, x2=x, W2=W_root, act=act)
        ctx.rel, ctx.act, ctx.has_bias = rel, act, b_rel is not None
        ctx.save_for_backward(x, edge_weight, W_rel, W_root, agg, y)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        x, ew, W_rel, W_root, agg, y = ctx.saved_tensors
        rel: Relation = ctx.rel
        g = act_bwd_raw(_c(g), y, ctx.act)
        gW_rel, gb = linear_bwd_w_raw(g, agg, ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3])
        gW_root = linear_bwd_w_raw(g, x, True, False)[0] if ctx.needs_input_grad[4] else None
        gx = None
        if ctx.needs_input_grad[0]:
            g_agg = linear_raw(g, W_rel, w_layout=1)[0]
            gx = spmm_weighted_raw(rel.csr_t, ew, g_agg)
            gx += linear_raw(g, W_root, w_layout=1)[0]
        # d/d edge_weight is not provided: gcn_norm weights are constants on this path
        return gx, None, gW_rel, gb, gW_root, None, None


# --------------------------------------------------------------------------- #
# GATConv, heads=1, bipartite, add_self_loops=False
# --------------------------------------------------------------------------- #
class GATConvFn(Function):
    @staticmethod
    def forward(ctx, x_src: Tensor, x_dst: Tensor, W_src: Tensor, W_dst: Tensor, att_src: Tensor,
                att_dst: Tensor, bias: Optional[Tensor], rel: Relation, slope: float, act: int):
        x_src, x_dst, W_src, W_dst, bias = _c(x_src), _c(x_dst), _c(W_src), _c(W_dst), _c(bias)
        att_s, att_d = _c(att_src).view(-1), _c(att_dst).view(-1)
        hs, a_s = linear_raw(x_src, W_src, att=att_s)
        hd, a_d = linear_raw(x_dst, W_dst, att=att_d)
        H = hs.shape[1]
        alpha = torch.empty(max(rel.num_edges, 1), dtype=torch.float32, device=hs.device)
        out = torch.empty(rel.num_dst, H, dtype=torch.float32, device=hs.device)
        call("hscn_gat_segment_fwd", ptr(rel.csr.rowptr), ptr(rel.csr.col), ptr(a_s), ptr(a_d), ptr(hs),
             ptr(bias), ptr(alpha), ptr(out), rel.num_dst, H, float(slope), 0, act, stream())
        ctx.rel, ctx.act, ctx.slope, ctx.has_bias = rel, act, float(slope), bias is not None
        ctx.save_for_backward(x_src, x_dst, W_src, W_dst, att_s, att_d, hs, hd, a_s, a_d, alpha, out)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        x_src, x_dst, W_src, W_dst, att_s, att_d, hs, hd, a_s, a_d, alpha, out = ctx.saved_tensors
        rel: Relation = ctx.rel
        H = hs.shape[1]
        dev = hs.device
        g = act_bwd_raw(_c(g), out, ctx.act)
        gb = linear_bwd_w_raw(g, None, False, True)[1] if (ctx.has_bias and ctx.needs_input_grad[6]) else None
        g_pre = torch.empty(max(rel.num_edges, 1), dtype=torch.float32, device=dev)
        g_a_d = torch.empty(rel.num_dst, dtype=torch.float32, device=dev)
        call("hscn_gat_segment_bwd_dst", ptr(rel.csr.rowptr), ptr(rel.csr.col), ptr(a_s), ptr(a_d), ptr(hs),
             ptr(alpha), ptr(g), ptr(g_pre), ptr(g_a_d), rel.num_dst, H, ctx.slope, stream())
        g_a_s = torch.empty(rel.num_src, dtype=torch.float32, device=dev)
        g_hs = torch.empty(rel.num_src, H, dtype=torch.float32, device=dev)
        call("hscn_gat_segment_bwd_src", ptr(rel.csr_t.rowptr), ptr(rel.csr_t.col), ptr(rel.pos_t), ptr(alpha),
             ptr(g_pre), ptr(g), ptr(att_s), ptr(g_a_s), ptr(g_hs), rel.num_src, H, stream())
        g_att_s = linear_bwd_w_raw(g_a_s.view(-1, 1), hs, True, False)[0].view(1, 1, H)
        g_att_d = linear_bwd_w_raw(g_a_d.view(-1, 1), hd, True, False)[0].view(1, 1, H)
        gW_src = linear_bwd_w_raw(g_hs, x_src, True, False)[0] if ctx.needs_input_grad[2] else None
        gx_src = linear_raw(g_hs, W_src, w_layout=1)[0] if ctx.needs_input_grad[0] else None
        gW_dst = gx_dst = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            g_hd = linear_raw(g_a_d.view(-1, 1), att_d.view(H, 1))[0]
            if ctx.needs_input_grad[3]:
                gW_dst = linear_bwd_w_raw(g_hd, x_dst, True, False)[0]
            if ctx.needs_input_grad[1]:
                gx_dst = linear_raw(g_hd, W_dst, w_layout=1)[0]
        return gx_src, gx_dst, gW_src, gW_dst, g_att_s, g_att_d, gb, None, None, None


# --------------------------------------------------------------------------- #
# global_mean_pool
# --------------------------------------------------------------------------- #
class SegmentMeanFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, seg: CSR, batch: Tensor, identity_nodes: bool):
        x = _c(x)
        out = torch.empty(seg.num_rows, x.shape[1], dtype=torch.float32, device=x.device)
        call("hscn_segment_mean_fwd", ptr(seg.rowptr), None if identity_nodes else ptr(seg.col), ptr(x),
             ptr(out), seg.num_rows, x.shape[1], stream())
        ctx.seg = seg
        ctx.save_for_backward(batch)
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (batch,) = ctx.saved_tensors
        g = _c(g)
        gx = torch.empty(ctx.n, g.shape[1], dtype=torch.float32, device=g.device)
        call("hscn_segment_mean_bwd", ptr(ctx.seg.rowptr), ptr(batch), ptr(g), ptr(gx), ctx.n, g.shape[1],
             stream())
        return gx, None, None, None


# --------------------------------------------------------------------------- #
# MinCUT pooling losses on the edge-list route
# --------------------------------------------------------------------------- #
class MinCutSparseFn(Function):
    """(logits, x) -> (S, losses[2] = {mincut, ortho}, pooled_x, pooled_adj).
    Gradients flow from the two losses to ``logits`` (the only gradients the
    reference's stage A uses: model/hscn.py:63 discards out/out_adj)."""

    @staticmethod
    def forward(ctx, logits: Tensor, x: Optional[Tensor], rel: Relation, node_ptr: Tensor, num_graphs: int):
        logits, x = _c(logits), _c(x)
        n, K = logits.shape
        dev = logits.device
        Fx = x.shape[1] if x is not None else 0
        S = torch.empty_like(logits)
        stats = torch.empty(num_graphs, 4, dtype=torch.float32, device=dev)
        ss = torch.empty(num_graphs, K, K, dtype=torch.float32, device=dev)
        px = torch.empty(num_graphs, K, max(Fx, 1), dtype=torch.float32, device=dev) if x is not None else None
        padj = torch.empty(num_graphs, K, K, dtype=torch.float32, device=dev)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        row_csr = rel.csr_t  # keyed by edge ROW (= source side of edge_index)
        call("hscn_mincut_sparse_fwd", ptr(logits), ptr(x), ptr(row_csr.rowptr), ptr(row_csr.col),
             ptr(node_ptr), ptr(S), ptr(stats), ptr(ss), ptr(px), ptr(padj), ptr(losses), n, num_graphs, K, Fx,
             stream())
        ctx.rel, ctx.G = rel, num_graphs
        ctx.save_for_backward(S, stats, ss, node_ptr)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(S, padj)
        if px is not None:
            ctx.mark_non_differentiable(px)
        return S, losses[0], losses[1], px, padj

    @staticmethod
    def backward(ctx, gS, g_mc, g_o, g_px, g_padj):
        S, stats, ss, node_ptr = ctx.saved_tensors
        rel: Relation = ctx.rel
        dev = S.device
        gl = _pack_loss_grads(g_mc, g_o, dev)
        g_logits = torch.empty_like(S)
        row_csr, col_csr = rel.csr_t, rel.csr
        call("hscn_mincut_sparse_bwd", ptr(S), ptr(stats), ptr(ss), ptr(row_csr.rowptr), ptr(row_csr.col),
             ptr(col_csr.rowptr), ptr(col_csr.col), ptr(node_ptr), ptr(gl), ptr(g_logits), S.shape[0], ctx.G,
             S.shape[1], stream())
        return g_logits, None, None, None, None


def _pack_loss_grads(g_mc: Optional[Tensor], g_o: Optional[Tensor], dev) -> Tensor:
    """[dL/dmincut, dL/dortho] as one device pair: the two losses are separate autograd outputs (slicing
    one [2] output costs six fill / copy / add launches in the backward), packing their scalar gradients is one."""
    if g_mc is None and g_o is None:
        return torch.zeros(2, dtype=torch.float32, device=dev)
    z = None
    if g_mc is None or g_o is None:
        z = torch.zeros((), dtype=torch.float32, device=dev)
    return torch.stack([(g_mc if g_mc is not None else z).reshape(()), (g_o if g_o is not None else z).reshape(())])


# --------------------------------------------------------------------------- #
# MinCUT pooling, dense route (matrix cores)
# --------------------------------------------------------------------------- #
def _norm_ws(N: int, H: int, dev) -> Tensor:
    return torch.empty(max(1, _hip.lib().hscn_norm_workspace_bytes(N, H) // 4), dtype=torch.float32, device=dev)


class LayerNormFn(Function):
    """torch.nn.functional.layer_norm over the last dim of [N, H] (reference model/mpnn.py:55-56)."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        N, H = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)
        rstd = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)
        call("hscn_layer_norm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), N, H, float(eps), stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean, rstd = ctx.saved_tensors
        N, H = x.shape
        gy = _c(gy)
        gx = torch.empty_like(x)
        gg = torch.empty(H, dtype=torch.float32, device=x.device)
        gb = torch.empty(H, dtype=torch.float32, device=x.device)
        ws = _norm_ws(N, H, x.device)
        call("hscn_layer_norm_bwd", ptr(gy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(gx), ptr(gg), ptr(gb), N, H,
             ptr(ws), ws.numel() * 4, stream())
        return gx, gg, gb, None


class BatchNormFn(Function):
    """torch.nn.functional.batch_norm on [N, H] (reference model/mpnn.py:53-54): batch statistics and the running
    update in training mode, running statistics in eval mode."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor],
                running_var: Optional[Tensor], training: bool, momentum: float, eps: float):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        N, H = x.shape
        y = torch.empty_like(x)
        sm = torch.empty(H, dtype=torch.float32, device=x.device)
        sr = torch.empty(H, dtype=torch.float32, device=x.device)
        ws = _norm_ws(N, H, x.device)
        call("hscn_batch_norm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(y), ptr(sm),
             ptr(sr), N, H, float(eps), float(momentum), 1 if training else 0, ptr(ws), ws.numel() * 4, stream())
        ctx.save_for_backward(x, gamma, sm, sr)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, sm, sr = ctx.saved_tensors
        N, H = x.shape
        gy = _c(gy)
        gx = torch.empty_like(x)
        gg = torch.empty(H, dtype=torch.float32, device=x.device)
        gb = torch.empty(H, dtype=torch.float32, device=x.device)
        ws = _norm_ws(N, H, x.device)
        call("hscn_batch_norm_bwd", ptr(gy), ptr(x), ptr(gamma), ptr(sm), ptr(sr), ptr(gx), ptr(gg), ptr(gb), N, H,
             1 if ctx.training else 0, ptr(ws), ws.numel() * 4, stream())
        return gx, gg, gb, None, None, None, None, None


class MinCutDenseRaggedFn(Function):
    """``MinCutDenseFn`` for a batch of graphs of DIFFERENT sizes: (logits [N,K], x [N,F] | None, adj [B,nmax,nmax]
    float32 -- or uint8 [B,nmax,round_up(nmax,32)] -- zero beyond each graph, nptr int32 [B+1], gid int32 [N]) -> (S [N,K], mincut, ortho, pooled_x, pooled_adj);
    losses = mean over graphs, every graph's terms equal to the single-graph call's (hscn_mincut_dense_ragged_*)."""

    @staticmethod
    def forward(ctx, logits: Tensor, x: Optional[Tensor], adj: Tensor, nptr: Tensor, gid: Tensor,
                asym: Optional[Tensor] = None):
        """asym (uint8 adjacency only): int32 [B] from ``to_dense_adj_ragged(..., symmetry=True)`` -- graphs flagged 0 are
        symmetric, and the backward takes their ``A^T S`` from the forward's ``A S`` instead of computing it."""
        if adj.dtype not in (torch.float32, torch.uint8):
            raise TypeError(f"the dense adjacency is float32 or uint8 (got {adj.dtype})")
        logits, x, adj = _c(logits), _c(x), adj.contiguous()
        N, K = logits.shape
        B, nmax = adj.shape[0], adj.shape[1]
        dev = logits.device
        Fx = x.shape[1] if x is not None else 0
        S = torch.empty_like(logits)
        AS = torch.empty_like(logits)
        deg = torch.empty(N, dtype=torch.float32, device=dev)
        stats = torch.empty(B, 4, dtype=torch.float32, device=dev)
        ss = torch.empty(B, K, K, dtype=torch.float32, device=dev)
        px = torch.empty(B, K, Fx, dtype=torch.float32, device=dev) if x is not None else None
        padj = torch.empty(B, K, K, dtype=torch.float32, device=dev)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        call("hscn_mincut_dense_ragged_fwd", ptr(x), ptr(adj), adj.element_size(), ptr(logits), ptr(nptr), N, B, nmax, K, Fx, ptr(S), ptr(AS),
             ptr(deg), ptr(stats), ptr(ss), ptr(px), ptr(padj), ptr(losses), stream())
        ctx.has_asym = asym is not None and adj.dtype == torch.uint8
        if ctx.has_asym:
            ctx.save_for_backward(adj, S, AS, deg, stats, ss, nptr, gid, asym)
        else:
            ctx.save_for_backward(adj, S, AS, deg, stats, ss, nptr, gid)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(S, padj)
        if px is not None:
            ctx.mark_non_differentiable(px)
        return S, losses[0], losses[1], px, padj

    @staticmethod
    def backward(ctx, gS, g_mc, g_o, g_px, g_padj):
        if ctx.has_asym:
            adj, S, AS, deg, stats, ss, nptr, gid, asym = ctx.saved_tensors
        else:
            adj, S, AS, deg, stats, ss, nptr, gid = ctx.saved_tensors
            asym = None
        N, K = S.shape
        B, nmax = adj.shape[0], adj.shape[1]
        gl = _pack_loss_grads(g_mc, g_o, S.device)
        AtS = torch.empty_like(S)
        SG = torch.empty_like(S)
        Gss = torch.empty_like(ss)
        g_logits = torch.empty_like(S)
        call("hscn_mincut_dense_ragged_bwd_sym", ptr(adj), adj.element_size(), ptr(S), ptr(AS), ptr(deg), ptr(stats), ptr(ss), ptr(gl),
             ptr(nptr), ptr(gid), N, B, nmax, K, ptr(AtS), ptr(SG), ptr(Gss), ptr(g_logits), ptr(asym), stream())
        return g_logits, None, None, None, None, None


class MinCutDenseFn(Function):
    """(logits [B,n,K], x [B,n,F] | None, adj [B,n,n]) -> (S, mincut, ortho, pooled_x, pooled_adj).
    Gradients flow from the two losses to ``logits``."""

    @staticmethod
    def forward(ctx, logits: Tensor, x: Optional[Tensor], adj: Tensor):
        logits, x, adj = _c(logits), _c(x), _c(adj)
        B, n, K = logits.shape
        dev = logits.device
        Fx = x.shape[2] if x is not None else 0
        S = torch.empty_like(logits)
        AS = torch.empty_like(logits)
        deg = torch.empty(B, n, dtype=torch.float32, device=dev)
        stats = torch.empty(B, 4, dtype=torch.float32, device=dev)
        ss = torch.empty(B, K, K, dtype=torch.float32, device=dev)
        px = torch.empty(B, K, Fx, dtype=torch.float32, device=dev) if x is not None else None
        padj = torch.empty(B, K, K, dtype=torch.float32, device=dev)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        call("hscn_mincut_dense_fwd", ptr(x), ptr(adj), ptr(logits), B, n, K, Fx, ptr(S), ptr(AS), ptr(deg),
             ptr(stats), ptr(ss), ptr(px), ptr(padj), ptr(losses), stream())
        ctx.save_for_backward(adj, S, AS, deg, stats, ss)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(S, padj)
        if px is not None:
            ctx.mark_non_differentiable(px)
        return S, losses[0], losses[1], px, padj

    @staticmethod
    def backward(ctx, gS, g_mc, g_o, g_px, g_padj):
        adj, S, AS, deg, stats, ss = ctx.saved_tensors
        B, n, K = S.shape
        gl = _pack_loss_grads(g_mc, g_o, S.device)
        AtS = torch.empty_like(S)
        SG = torch.empty_like(S)
        Gss = torch.empty_like(ss)
        g_logits = torch.empty_like(S)
        call("hscn_mincut_dense_bwd", ptr(adj), ptr(S), ptr(AS), ptr(deg), ptr(stats), ptr(ss), ptr(gl), B, n, K,
             ptr(AtS), ptr(SG), ptr(Gss), ptr(g_logits), stream())
        return g_logits, None, None
