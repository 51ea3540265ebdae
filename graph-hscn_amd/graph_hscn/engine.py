"""Graph-resident execution of HSCN.forward / backward (csrc/resident.hip).

``HSCN.forward`` (reference model/hscn.py:102-114) normally runs L x 3 relation
convs, ReLUs, a pool and two linears as separate operators.  For block-diagonal
batches whose graphs fit one CU's LDS (every LRGB molecule / superpixel graph at
the reference's widths) the whole forward is ONE launch with a workgroup per
graph, and the backward is one launch plus an ordered per-parameter reduction
over graphs.  This module decides whether a call qualifies and wraps the two C
entry points in a single autograd Function.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _hip
from ._hip import ACT, call, ptr, stream

LL = ("local", "to", "local")
VV = ("virtual", "to", "virtual")
LV = ("local", "to", "virtual")


@dataclass
class ResidentMeta:
    """Per-batch segmentation the kernels need (device int32 ptrs + host maxima)."""
    lptr: Tensor
    vptr: Tensor
    eptr_ll: Tensor
    eptr_vv: Tensor
    eptr_lv: Tensor
    num_graphs: int
    max_n: int
    max_v: int
    max_ell: int
    max_evv: int
    flag: Tensor

    def check(self) -> None:
        """Synchronising validity check of the last launches that used this meta."""
        f = int(self.flag.item())
        if f & 2:
            raise IndexError("an edge connects nodes of different graphs: the batch is not block-diagonal")
        if f & 4:
            raise ValueError("a graph exceeds the sizes the resident launch was configured for")
        if f & 8:
            raise RuntimeError("a virtual-branch workgroup of the one-launch step gave up waiting for its graph's local "
                               "activations (the virtual features of that step are invalid; prediction, loss and "
                               "gradients are not affected)")


def meta_from_batch(batch, device) -> Optional[ResidentMeta]:
    """Read the segmentation a ``graph_hscn.data.HeteroBatch`` carries; ``None`` if the
    object does not provide it (foreign containers use the layered operators)."""
    try:
        loc, vir = batch["local"], batch["virtual"]
        ell, evv, elv = batch[LL], batch[VV], batch[LV]
        tensors = [loc.ptr32, vir.ptr32, ell.ptr32, evv.ptr32, elv.ptr32]
        maxima = (int(loc.max_nodes), int(vir.max_nodes), int(ell.max_edges), int(evv.max_edges))
    except (AttributeError, KeyError, TypeError):
        return None
    if any((not isinstance(t, Tensor)) or t.device != device or t.dtype != torch.int32 for t in tensors):
        return None
    cached = getattr(batch, "_resident_meta", None)
    if cached is not None and cached.lptr is tensors[0]:
        return cached
    meta = ResidentMeta(*[t.contiguous() for t in tensors], int(tensors[0].numel()) - 1, *maxima,
                        torch.zeros(1, dtype=torch.int32, device=device))
    try:
        batch._resident_meta = meta
    except AttributeError:
        pass
    return meta


def supported(F: int, H: int, L: int, C: int, meta: ResidentMeta, dtype=torch.float32) -> bool:
    if dtype == torch.float16 and H > 32:
        return False                 # half storage: H in {16, 32} (csrc/resident_f16.hip)
    if dtype not in (torch.float32, torch.float16):
        return False
    return bool(_hip.lib().hscn_resident_supported(F, H, L, C, meta.max_n, meta.max_v, meta.max_ell, meta.max_evv))


def storage_suffix(dtype) -> str:
    """Entry-point suffix for the storage type of node features / activations (include/hscn.h)."""
    return "_f16" if dtype == torch.float16 else ""


def _ptr_table(ts: List[Optional[Tensor]]):
    arr = (ctypes.c_void_p * len(ts))(*[None if t is None else ptr(t) for t in ts])
    return arr


class _VirtualJob(ctypes.Structure):
    """include/hscn.h: hscn_virtual_job."""
    _fields_ = [("x_virtual", ctypes.c_void_p), ("ei_vv", ctypes.c_void_p), ("E_vv", ctypes.c_int64),
                ("ei_lv", ctypes.c_void_p), ("E_lv", ctypes.c_int64), ("vptr", ctypes.c_void_p),
                ("eptr_vv", ctypes.c_void_p), ("eptr_lv", ctypes.c_void_p),
                ("layer_params_host", ctypes.c_void_p), ("xv_out", ctypes.c_void_p), ("V", ctypes.c_int64),
                ("max_v", ctypes.c_int32), ("max_evv", ctypes.c_int32), ("slope", ctypes.c_float),
                ("st_rowptr_lv", ctypes.c_void_p), ("st_col_lv", ctypes.c_void_p),
                ("st_rowptr_vv", ctypes.c_void_p), ("st_col_vv", ctypes.c_void_p),
                ("st_dinv_v", ctypes.c_void_p), ("st_xv", ctypes.c_void_p)]


class _Structure(ctypes.Structure):
    """include/hscn.h: hscn_structure."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("ll_rowptr_d", "ll_col_d", "ll_rowptr_s", "ll_col_s", "ll_dinv",
                                               "lv_rowptr", "lv_col", "vv_rowptr", "vv_col", "vv_dinv")]


class BatchStructure:
    """The epoch-invariant structure of every graph of a block-diagonal hetero batch (or of a dataset laid out as
    one): four stable CSRs with graph-local int32 ids and two degree norms (include/hscn.h: hscn_structure).  The
    resident launches rebuild these in LDS every step unless they are handed one of these (structure_build =
    "dataset-resident"); results are bit-identical either way."""

    FIELDS = ("ll_rowptr_d", "ll_col_d", "ll_rowptr_s", "ll_col_s", "ll_dinv", "lv_rowptr", "lv_col", "vv_rowptr",
              "vv_col", "vv_dinv")

    def __init__(self, device, N: int, V: int, B: int, E_ll: int, E_lv: int, E_vv: int):
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.t = {"ll_rowptr_d": torch.zeros(N + B, **i32), "ll_col_d": torch.zeros(max(E_ll, 1), **i32),
                  "ll_rowptr_s": torch.zeros(N + B, **i32), "ll_col_s": torch.zeros(max(E_ll, 1), **i32),
                  "ll_dinv": torch.zeros(max(N, 1), **f32),
                  "lv_rowptr": torch.zeros(V + B, **i32), "lv_col": torch.zeros(max(E_lv, 1), **i32),
                  "vv_rowptr": torch.zeros(V + B, **i32), "vv_col": torch.zeros(max(E_vv, 1), **i32),
                  "vv_dinv": torch.zeros(max(V, 1), **f32)}
        self.c = _Structure(*[ptr(self.t[k]) for k in self.FIELDS])

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())


def build_structure(batch, meta: Optional["ResidentMeta"] = None) -> BatchStructure:
    """``hscn_resident_structure`` on a ``graph_hscn.data.HeteroBatch`` that lives on the device: one launch, one
    workgroup per graph.  The result is attached as ``batch.structure``."""
    dev = batch["local"].x.device
    meta = meta or meta_from_batch(batch, dev)
    if meta is None:
        raise TypeError("build_structure needs a graph_hscn HeteroBatch on the device")
    ei = {k: batch[k].edge_index.contiguous() for k in (LL, VV, LV)}
    N, V, B = int(batch["local"].x.size(0)), int(batch["virtual"].x.size(0)), meta.num_graphs
    st = BatchStructure(dev, N, V, B, ei[LL].size(1), ei[LV].size(1), ei[VV].size(1))
    call("hscn_resident_structure", ptr(ei[LL]), ei[LL].size(1), ptr(ei[VV]), ei[VV].size(1), ptr(ei[LV]),
         ei[LV].size(1), ptr(meta.lptr), ptr(meta.vptr), ptr(meta.eptr_ll), ptr(meta.eptr_vv), ptr(meta.eptr_lv), B,
         meta.max_n, meta.max_v, meta.max_ell, meta.max_evv, ctypes.byref(st.c), ptr(meta.flag), stream())
    try:
        batch.structure = st
    except AttributeError:
        pass
    return st


class _LossTail(ctypes.Structure):
    """include/hscn.h: hscn_loss_tail."""
    _fields_ = [("pred", ctypes.c_void_p), ("target", ctypes.c_void_p), ("kind", ctypes.c_int32)]


# final virtual features of the last step whose virtual branch rode on the backward launch (tests)
last_deferred_virtual: Optional[Tensor] = None


_CUS: Dict[int, int] = {}


def _cu_count(dev: torch.device) -> int:
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _CUS:
        _CUS[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _CUS[idx]


class HSCNResidentFn(Function):
    """inputs: x_local, x_virtual, ei_ll, ei_vv, ei_lv, meta, cfg, then parameters in
    the order  [W_ll, b_ll, W_vv, b_vv, W_src, W_dst, att_src, att_dst, b_gat] x L,
    W1, b1, W2, b2.   cfg = (head_act code, slope, compute_virtual, keep_virtual, overlap).

    overlap: the virtual branch never feeds the prediction (reference model/hscn.py:111 pools
    "local" only), and a 128-graph batch occupies half of the CUs.  In a training step the branch
    therefore leaves the critical path: its CSR builds and layer 0 (which reads input features only)
    run as extra workgroups of the forward launch (hscn_resident_fwd_with_virtual), layers 1.. as
    extra workgroups of the backward launch (hscn_resident_bwd_with_virtual), reading the local
    activations the forward stored.  Without a backward to ride on (no_grad, or keep_virtual
    wanting the features right away) the one-launch forward computes both branches.  Results are
    identical either way."""

    @staticmethod
    def forward(ctx, x_local, x_virtual, ei_ll, ei_vv, ei_lv, meta: ResidentMeta, cfg, *params):
        head_act, slope, compute_virtual, keep_virtual, overlap = cfg
        L = (len(params) - 4) // 9
        params = [p.contiguous() for p in params]
        W1, b1, W2, b2 = params[9 * L:]
        H, C = W1.shape[0], W2.shape[0]
        x_local = x_local.contiguous()
        x_virtual = x_virtual.contiguous()
        sdt = x_local.dtype                       # storage type of features and activations: float32 or float16
        if x_virtual.dtype != sdt:
            raise TypeError("local and virtual node features must share one storage dtype")
        sfx = storage_suffix(sdt)
        N, F = x_local.shape
        V = x_virtual.shape[0]
        B = meta.num_graphs
        dev = x_local.device
        need_bwd = any(ctx.needs_input_grad[7:])   # False under no_grad
        # (the extra workgroups pay only while they land on CUs the batch leaves idle: 2B <= number of CUs)
        defer = bool(compute_virtual and overlap and V > 0 and need_bwd and not keep_virtual and L >= 2
                     and 2 * B <= _cu_count(dev))
        acts = torch.empty(L, N, H, dtype=sdt, device=dev)
        pooled = torch.empty(B, H, dtype=torch.float32, device=dev)
        z = torch.empty(B, H, dtype=torch.float32, device=dev)
        pred = torch.empty(B, C, dtype=torch.float32, device=dev)
        # sigmoid(pred) costs the head ten more stores; with it the loss tail can ride on the backward launch
        score = torch.empty(B, C, dtype=torch.float32, device=dev) if need_bwd else None
        xv_out = torch.empty(max(V, 1), H, dtype=sdt, device=dev) if (compute_virtual and keep_virtual) else None
        # source-keyed CSR + degree norm: built in LDS by the forward launch, reused by the backward launch
        E_ll = ei_ll.size(1)
        csr_rp = torch.empty(N + B, dtype=torch.int32, device=dev) if need_bwd else None
        csr_col = torch.empty(max(E_ll, 1), dtype=torch.int32, device=dev) if need_bwd else None
        dinv = torch.empty(max(N, 1), dtype=torch.float32, device=dev) if need_bwd else None
        table = _ptr_table(params[: 9 * L])
        ctx.virtual = None
        if defer:
            E_lv, E_vv = ei_lv.size(1), ei_vv.size(1)
            state = (torch.empty(V + B, dtype=torch.int32, device=dev), torch.empty(max(E_lv, 1), dtype=torch.int32, device=dev),
                     torch.empty(V + B, dtype=torch.int32, device=dev), torch.empty(max(E_vv, 1), dtype=torch.int32, device=dev),
                     torch.empty(V, dtype=torch.float32, device=dev), torch.empty(V, H, dtype=sdt, device=dev))
            job = _VirtualJob(ptr(x_virtual), ptr(ei_vv), E_vv, ptr(ei_lv), E_lv, ptr(meta.vptr), ptr(meta.eptr_vv),
                              ptr(meta.eptr_lv), ctypes.cast(table, ctypes.c_void_p), None, V, meta.max_v,
                              meta.max_evv, float(slope), *[ptr(t) for t in state])
            call("hscn_resident_fwd_with_virtual" + sfx, ptr(x_local), ptr(ei_ll), E_ll, ptr(meta.lptr), ptr(meta.eptr_ll),
                 N, B, F, H, L, C, head_act, table, ptr(W1), ptr(b1), ptr(W2), ptr(b2), meta.max_n, meta.max_ell,
                 ptr(acts), ptr(pooled), ptr(z), ptr(pred), ptr(score), ptr(csr_rp), ptr(csr_col), ptr(dinv),
                 ptr(meta.flag), ctypes.byref(job), stream())
            # what the backward launch needs to run the rest of the virtual branch beside itself
            ctx.virtual = (x_virtual, ei_vv, ei_lv, params[: 9 * L], table, float(slope), state)
        else:
            call("hscn_resident_fwd" + sfx, ptr(x_local), ptr(x_virtual), ptr(ei_ll), ei_ll.size(1), ptr(ei_vv),
                 ei_vv.size(1), ptr(ei_lv), ei_lv.size(1), ptr(meta.lptr), ptr(meta.vptr), ptr(meta.eptr_ll),
                 ptr(meta.eptr_vv), ptr(meta.eptr_lv), N, V, B, F, H, L, C, head_act, float(slope), table,
                 ptr(W1), ptr(b1), ptr(W2), ptr(b2), meta.max_n, meta.max_v, meta.max_ell, meta.max_evv,
                 int(bool(compute_virtual)), ptr(acts), ptr(pooled), ptr(z), ptr(pred), ptr(score), ptr(xv_out),
                 ptr(csr_rp), ptr(csr_col), ptr(dinv), ptr(meta.flag), stream())
        ctx.meta, ctx.head_act, ctx.dims = meta, head_act, (N, F, H, L, C, B)
        ctx.sfx = sfx
        ctx.csr = (csr_rp, csr_col, dinv)
        ctx.save_for_backward(x_local, ei_ll, acts, pooled, z, W1, W2, *[params[9 * l] for l in range(L)])
        ret_xv = xv_out if keep_virtual else None
        ctx.mark_non_differentiable(*[t for t in (ret_xv, score) if t is not None])
        ctx.set_materialize_grads(False)
        return pred, ret_xv, score

    @staticmethod
    def backward(ctx, g_pred, *_):
        x_local, ei_ll, acts, pooled, z, W1, W2, *W_ll = ctx.saved_tensors
        if g_pred is None:
            return (None,) * (7 + 9 * ((len(W_ll))) + 4)
        meta: ResidentMeta = ctx.meta
        N, F, H, L, C, B = ctx.dims
        dev = x_local.device
        P = int(_hip.lib().hscn_resident_param_count(F, H, L, C))
        # the loss node hands its gradient over unevaluated: either (unscaled gradient, scalar) -- the launch
        # applies the scalar -- or (pred, target, kind, scalar) -- the launch evaluates the loss tail itself
        # and returns the loss value as one more reduced column
        from .loss import LazyCriterionGrad, LazyScaled
        g_scale, tail, tail_ref = None, None, None
        if isinstance(g_pred, LazyCriterionGrad) and g_pred.pred.shape == (B, C):
            tail_ref = g_pred
            tail = _LossTail(ptr(g_pred.pred), ptr(g_pred.target), int(g_pred.kind))
            g_scale, g_pred = g_pred.scale, None
        elif isinstance(g_pred, LazyScaled):
            g_pred, g_scale = g_pred.grad_unscaled, g_pred.scale
        if g_pred is not None:
            g_pred = g_pred.contiguous()
        Pw = P + (1 if tail is not None else 0)
        partials = torch.empty(B, Pw, dtype=torch.float32, device=dev)
        grads = torch.empty(Pw, dtype=torch.float32, device=dev)
        table = _ptr_table(list(W_ll))
        args = (ptr(x_local), ptr(ei_ll), ei_ll.size(1), ptr(meta.lptr), ptr(meta.eptr_ll), N, B,
                F, H, L, C, ctx.head_act, table, ptr(W1), ptr(W2), ptr(acts), ptr(pooled), ptr(z), ptr(g_pred),
                ptr(g_scale), ptr(ctx.csr[0]), ptr(ctx.csr[1]), ptr(ctx.csr[2]), meta.max_n, meta.max_ell, ptr(partials),
                ptr(grads), ptr(meta.flag), ctypes.byref(tail) if tail is not None else None)
        if tail_ref is not None:
            tail_ref.state.fill(grads[P:P + 1].view(()))      # the loss value, once this launch has run
        if ctx.virtual is not None:
            global last_deferred_virtual
            x_virtual, ei_vv, ei_lv, _keep, vtable, slope, state = ctx.virtual
            V = x_virtual.shape[0]
            xv = torch.empty(max(V, 1), H, dtype=x_virtual.dtype, device=dev)
            job = _VirtualJob(ptr(x_virtual), ptr(ei_vv), ei_vv.size(1), ptr(ei_lv), ei_lv.size(1), ptr(meta.vptr),
                              ptr(meta.eptr_vv), ptr(meta.eptr_lv), ctypes.cast(vtable, ctypes.c_void_p), ptr(xv),
                              V, meta.max_v, meta.max_evv, slope, *[ptr(t) for t in state])
            call("hscn_resident_bwd_with_virtual" + ctx.sfx, *args, ctypes.byref(job), stream())
            last_deferred_virtual = xv
            ctx.virtual = None
        else:
            call("hscn_resident_bwd" + ctx.sfx, *args, stream())
        out: List[Optional[Tensor]] = [None] * (7 + 9 * L + 4)
        off = 0
        for l in range(L):
            fin = F if l == 0 else H
            out[7 + 9 * l] = grads[off: off + H * fin].view(H, fin)
            off += H * fin
            out[7 + 9 * l + 1] = grads[off: off + H]
            off += H
        base = 7 + 9 * L
        out[base] = grads[off: off + H * H].view(H, H); off += H * H
        out[base + 1] = grads[off: off + H]; off += H
        out[base + 2] = grads[off: off + C * H].view(C, H); off += C * H
        out[base + 3] = grads[off: off + C]
        return tuple(out)


# --------------------------------------------------------------------------- #
# stage A: gcn_norm + SCN.forward + MinCUT losses, graph-resident
# --------------------------------------------------------------------------- #
@dataclass
class ScnMeta:
    nptr: Tensor
    eptr: Tensor
    num_graphs: int
    max_n: int
    max_e: int
    flag: Tensor
    ticket: Optional[Tensor] = None     # device counter of the in-launch loss reduction (stays zero between launches)

    def check(self) -> None:
        f = int(self.flag.item())
        if f & 2:
            raise IndexError("an edge connects nodes of different graphs (or leaves the graph)")
        if f & 4:
            raise ValueError("a graph exceeds the sizes the resident launch was configured for")


def scn_meta(data, device) -> ScnMeta:
    """Segmentation of a ``graph_hscn.data.Batch`` (block-diagonal) or a single ``Data`` graph."""
    cached = getattr(data, "_scn_meta", None)
    if cached is not None and cached.nptr.device == device:
        return cached
    if "ptr32" in data and "eptr32" in data:
        meta = ScnMeta(data.ptr32.to(device), data.eptr32.to(device), int(data.num_graphs), int(data.max_nodes),
                       int(data.max_edges), torch.zeros(1, dtype=torch.int32, device=device),
                       torch.zeros(1, dtype=torch.int32, device=device))
    else:
        n, e = int(data.num_nodes), int(data.edge_index.size(1))
        meta = ScnMeta(torch.tensor([0, n], dtype=torch.int32, device=device),
                       torch.tensor([0, e], dtype=torch.int32, device=device), 1, n, e,
                       torch.zeros(1, dtype=torch.int32, device=device),
                       torch.zeros(1, dtype=torch.int32, device=device))
    try:
        data._scn_meta = meta
    except AttributeError:
        pass
    return meta


class SCNResidentFn(Function):
    """(x, raw edge_index, meta, act, W_rel, b_rel, W_root, W_mlp, b_mlp) -> (S, mc_loss, o_loss, mc_loss + o_loss).
    The two losses are separate autograd outputs (0-dim views of one [2] buffer the launch fills), so
    ``(mc + o).backward()`` reaches the backward launch as two scalar gradients with no glue kernels
    in between (slicing one [2] output cost six fill / copy / add launches per step)."""

    @staticmethod
    def forward(ctx, x, edge_index, meta: ScnMeta, act: int, W_rel, b_rel, W_root, W_mlp, b_mlp):
        x = x.contiguous()
        edge_index = edge_index.contiguous()
        W_rel, b_rel, W_root, W_mlp, b_mlp = (t.contiguous() for t in (W_rel, b_rel, W_root, W_mlp, b_mlp))
        N, F = x.shape
        H, K = W_rel.shape[0], W_mlp.shape[0]
        B = meta.num_graphs
        dev = x.device
        if x.dtype not in (torch.float32, torch.float16):
            raise TypeError("node features must be float32 or float16")
        ctx.sfx = storage_suffix(x.dtype)
        S = torch.empty(N, K, dtype=torch.float32, device=dev)
        y = torch.empty(N, H, dtype=x.dtype, device=dev)        # saved hidden activation: the features' storage type
        stats = torch.empty(B, 4, dtype=torch.float32, device=dev)
        ss = torch.empty(B, K, K, dtype=torch.float32, device=dev)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        if meta.ticket is None:
            meta.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        E = edge_index.size(1)
        # both CSRs, agg = A_hat x and the out-degree: built in LDS by the forward launch, kept for the backward
        need_bwd = any(ctx.needs_input_grad[4:])
        ex = None
        if need_bwd:
            ex = (torch.empty(N + B, dtype=torch.int32, device=dev), torch.empty(max(E, 1), dtype=torch.int32, device=dev),
                  torch.empty(N + B, dtype=torch.int32, device=dev), torch.empty(max(E, 1), dtype=torch.int32, device=dev),
                  torch.empty(max(N, 1), 16, dtype=torch.float32, device=dev), torch.empty(max(N, 1), dtype=torch.float32, device=dev))
        call("hscn_scn_resident_fwd" + ctx.sfx, ptr(x), ptr(edge_index) if E else None, E, ptr(meta.nptr), ptr(meta.eptr), N, B,
             F, H, K, act, ptr(W_rel), ptr(b_rel), ptr(W_root), ptr(W_mlp), ptr(b_mlp), meta.max_n, meta.max_e,
             ptr(S), ptr(y), ptr(stats), ptr(ss), ptr(losses), ptr(meta.ticket), *([ptr(t) for t in ex] if ex else [None] * 6),
             ptr(meta.flag), stream())
        ctx.ex = ex
        ctx.meta, ctx.act, ctx.dims = meta, act, (N, F, H, K, B, E)
        ctx.save_for_backward(x, edge_index, W_mlp, S, y, stats, ss)
        ctx.mark_non_differentiable(S)
        ctx.set_materialize_grads(False)
        return S, losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, gS, g_mc, g_o, g_total):
        x, edge_index, W_mlp, S, y, stats, ss = ctx.saved_tensors
        meta: ScnMeta = ctx.meta
        N, F, H, K, B, E = ctx.dims
        dev = x.device
        P = int(_hip.lib().hscn_scn_resident_param_count(F, H, K))
        partials = torch.empty(B, P, dtype=torch.float32, device=dev)
        grads = torch.empty(P, dtype=torch.float32, device=dev)
        if g_total is not None:            # d(mc + o): the same scalar reaches both losses
            g_mc = g_total if g_mc is None else g_mc + g_total
            g_o = g_total if g_o is None else g_o + g_total
        g_mc = None if g_mc is None else g_mc.reshape(1).contiguous()
        g_o = None if g_o is None else g_o.reshape(1).contiguous()
        call("hscn_scn_resident_bwd" + ctx.sfx, ptr(x), ptr(edge_index) if E else None, E, ptr(meta.nptr), ptr(meta.eptr), N, B,
             F, H, K, ctx.act, ptr(W_mlp), ptr(S), ptr(y), ptr(stats), ptr(ss), ptr(g_mc), ptr(g_o), *[ptr(t) for t in ctx.ex], meta.max_n, meta.max_e,
             ptr(partials), ptr(grads), ptr(meta.flag), stream())
        o = 0
        gW_rel = grads[o:o + H * F].view(H, F); o += H * F
        gb_rel = grads[o:o + H]; o += H
        gW_root = grads[o:o + H * F].view(H, F); o += H * F
        gW_mlp = grads[o:o + K * H].view(K, H); o += K * H
        gb_mlp = grads[o:o + K]
        return None, None, None, None, gW_rel, gb_rel, gW_root, gW_mlp, gb_mlp
