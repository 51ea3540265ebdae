"""Training steps as plain sequences of C-ABI launches on preallocated buffers (no autograd inside).

The reference's iteration is ``pred = model(...); loss = criterion(...); loss.backward()``
(/root/reference/graph_hscn/train/train.py:73-95; stage A: train/train_clustering.py:44-50).  The
eager product path keeps that form (``engine.HSCNResidentFn`` / ``SCNResidentFn`` are autograd
Functions).  A step that is going to be REPLAYED (hipGraph) should not run the autograd engine while
it is being captured: autograd binds every parameter's gradient-accumulation node to the stream it
was first used on, and a node that an earlier eager step left alive ties the capture to that
non-capturing stream -- ``capture_end`` then fails inside the HIP runtime (a host segfault, round 1:
gpurun_out/t12.log, scn_dbg.log).  The classes here issue exactly the launches the autograd path
issues -- same entry points, same arguments, bit-identical outputs (tests/test_gpu_step.py) -- into
buffers allocated once, so a capture contains kernel nodes only and nothing of an earlier step can
reach into it.

  ResidentTrainStep  stage C: zero grads -> HSCN.forward -> criterion -> backward
  ScnTrainStep       stage A: zero grads -> gcn_norm + SCN.forward -> (mc + o).backward
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _hip
from . import engine as _engine
from ._hip import call, ptr, stream

LL, VV, LV = _engine.LL, _engine.VV, _engine.LV


def _loss_kind(loss_fn: str, C_pred: Tuple[int, int], target: Tensor) -> int:
    if target.dim() != 2 or tuple(target.shape) != tuple(C_pred):
        raise ValueError("the fused step takes [B, C] targets (the multilabel BCE / L1 branches of "
                         "loss.py:8-10,16-19); class-index targets go through loss.criterion")
    return 0 if loss_fn == "cross_entropy" else 1


class ResidentTrainStep:
    """``for p: p.grad = None; pred = model(x_dict, edge_index_dict, batch); loss, score =
    criterion(loss_fn, pred, batch["local"].y); loss.backward()`` on the graph-resident engine,
    issued by ``run()`` as two or three launches on the current stream.

    Outputs (tensors that every ``run()`` refreshes in place): ``pred`` [B,C], ``score`` [B,C]
    (= sigmoid(pred), what ``criterion`` returns beside the loss), ``loss`` (0-dim), ``grads``
    (flat, the parameter gradients in launch order) and ``virtual`` (final virtual features, as
    ``HSCN.last_virtual``).  ``bind_grads()`` points every ``p.grad`` at its slice of ``grads``
    (parameters the prediction does not depend on -- the whole virtual branch, DESIGN.md section 2 --
    get ``None``, as autograd leaves them).

    The batch may be a ``replay.StaticHeteroBatch.batch``: tensor shapes are capacities then, the
    kernels read the real per-graph ranges from the device-side segment tables."""

    def __init__(self, model, batch, loss_fn: str, target: Optional[Tensor] = None, one_launch: Optional[bool] = None,
                 structure=None):
        """``one_launch``: None = take the one-launch step (include/hscn.h: hscn_resident_train_step -- forward, loss
        tail and backward of a graph in one workgroup, nothing exported in between) whenever the graphs fit it,
        else the forward + backward launch pair; False = always the pair; True = insist (raises if unsupported).
        ``structure``: None = every step rebuilds the graphs' CSRs and degree norms in LDS from the COO slices
        (structure_build "per-step"); an ``engine.BatchStructure`` (or "batch" = ``batch.structure``, which
        ``engine.build_structure`` / ``DeviceHeteroDataset(resident_structure=True)`` attach) = the one-launch step
        loads them (structure_build "dataset-resident": graph structure is epoch-invariant).  Same results."""
        from .model.hscn import HSCN, _act_name
        if not isinstance(model, HSCN):
            raise TypeError("ResidentTrainStep drives graph_hscn.model.hscn.HSCN")
        x_dict, ei_dict = batch.x_dict, batch.edge_index_dict
        dev = x_dict["local"].device
        if dev.type != "cuda":
            raise RuntimeError("the graph-resident step runs on the MI355X HIP path: move the batch to 'cuda'")
        keep = model.engine
        model.engine = "resident"            # raises with the reason when the batch / model does not qualify
        try:
            meta, act_name = model._resident_plan(x_dict, ei_dict, batch)
        finally:
            model.engine = keep
        self.model, self.batch, self.meta = model, batch, meta
        self.loss_fn = loss_fn
        params = [p.contiguous() for p in model._resident_params()]
        L = (len(params) - 4) // 9
        W1, b1, W2, b2 = params[9 * L:]
        H, C = W1.shape[0], W2.shape[0]
        self.x_local = x_dict["local"].contiguous()
        self.x_virtual = x_dict["virtual"].contiguous()
        sdt = self.x_local.dtype        # storage type of features and activations (include/hscn.h: *_f16 twins)
        if sdt not in (torch.float32, torch.float16) or self.x_virtual.dtype != sdt:
            raise TypeError("node features must be float32 (train/train.py:79 casts them) or float16, local and "
                            "virtual alike")
        self._sfx = _engine.storage_suffix(sdt)
        self.ei = {k: ei_dict[k].contiguous() for k in (LL, VV, LV)}
        N, F = self.x_local.shape
        V = self.x_virtual.shape[0]
        B = meta.num_graphs
        self.dims = (N, V, F, H, L, C, B)
        self.head_act = _engine.ACT[act_name]
        self.slope = float(model.convs[0].convs["__".join(LV)].negative_slope)
        y = target if target is not None else batch["local"].y
        if y.dtype != torch.float32 or not y.is_contiguous() or y.device != dev:
            raise TypeError("targets must be a contiguous float32 tensor on the batch's device")
        self.kind = _loss_kind(loss_fn, (B, C), y)
        self.target = y
        self._params = params
        self._table = _engine._ptr_table(params[: 9 * L])
        self._wll_table = _engine._ptr_table([params[9 * l] for l in range(L)])
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.acts = torch.empty(L, N, H, dtype=sdt, device=dev)
        self.pooled = torch.empty(B, H, **f32)
        self.z = torch.empty(B, H, **f32)
        self.pred = torch.empty(B, C, **f32)
        self.score = torch.empty(B, C, **f32)
        E_ll, E_vv, E_lv = (self.ei[k].size(1) for k in (LL, VV, LV))
        self.csr = (torch.empty(N + B, **i32), torch.empty(max(E_ll, 1), **i32), torch.empty(max(N, 1), **f32))
        self.virtual = torch.empty(max(V, 1), H, dtype=sdt, device=dev) if model.compute_virtual else None
        # the virtual branch rides on the two launches as extra workgroups while they land on idle CUs
        cus = _engine._cu_count(dev)
        self.defer = bool(model.compute_virtual and model.overlap_virtual and V > 0 and L >= 2 and 2 * B <= cus)
        # the one-launch step's virtual workgroups also fit beside small graphs' 4-wave workgroups (several per CU)
        per_cu = int(_hip.lib().hscn_resident_train_step_wgs_per_cu(F, H, L, C, meta.max_n, meta.max_ell, meta.max_v,
                                                                    meta.max_evv))
        # ... and, H = 16, at ANY batch size: 2 B workgroups then take several rounds of the chip, local programs first
        # (block ids [0, B)), so a virtual workgroup's producer has always been dispatched before it; measured against the
        # launch pair at B = 256 / 512 / 1024: 47.9 / 86.2 / 163 us against 56.5 / 105 / 193 (profiles/r03_large_batches.txt).
        # HSCN_ONE_LAUNCH_LARGE_B=0 restores the idle-CU rule.
        import os
        _lb = os.environ.get("HSCN_ONE_LAUNCH_LARGE_B", "1")     # ("all": H = 32 too -- measurement)
        # (graphs of the 4-wave class, n <= 64, keep the rule: PCQM-Contact at B = 2 048 / 4 096 runs 101 / 183 us as the
        # launch pair against 116 / 217)
        large_b = _lb == "all" or (H == 16 and meta.max_n > 64 and _lb != "0")
        self.idle_cus = bool(model.compute_virtual and model.overlap_virtual and V > 0 and
                             (2 * B <= cus * max(per_cu, 1) or large_b))
        self._state = None
        if self.defer:
            self._state = (torch.empty(V + B, **i32), torch.empty(max(E_lv, 1), **i32), torch.empty(V + B, **i32),
                           torch.empty(max(E_vv, 1), **i32), torch.empty(V, **f32), torch.empty(V, H, dtype=sdt, device=dev))
        # one-launch step: the virtual branch rides as B more workgroups of the same launch when they land on idle
        # CUs (same condition as `defer`); a batch that fills the chip by itself keeps the launch pair (its virtual
        # branch shares the local workgroups there)
        import os
        can = bool(_hip.lib().hscn_resident_train_step_supported(F, H, L, C, meta.max_n, meta.max_ell,
                                                                 meta.max_v if model.compute_virtual else 0,
                                                                 meta.max_evv if model.compute_virtual else 0))
        can = can and (self.idle_cus or not model.compute_virtual or V == 0)
        if one_launch is None:
            one_launch = can and os.environ.get("HSCN_ONE_LAUNCH", "1") != "0"
        elif one_launch and not can:
            raise RuntimeError("the one-launch step does not take this batch / model (H in {16, 32}, graphs that fit "
                               "its LDS layout, virtual branch on idle CUs only)")
        self.one_launch = bool(one_launch)
        if structure == "batch":
            structure = getattr(batch, "structure", None)
            if structure is None:
                raise ValueError("the batch carries no structure (engine.build_structure(batch) attaches one)")
        if structure is not None and not self.one_launch:
            raise RuntimeError("dataset-resident structure is a mode of the one-launch step")
        self.structure = structure
        self._sync = torch.zeros(32 + B, dtype=torch.int32, device=dev) if self.one_launch else None
        P = int(_hip.lib().hscn_resident_param_count(F, H, L, C))
        self.P = P
        self.partials = torch.empty(B, P + 1, **f32)
        self.grads = torch.zeros(P + 1, **f32)
        self.loss = self.grads[P:P + 1].view(())
        self._tail = _engine._LossTail(ptr(self.pred), ptr(self.target), int(self.kind))
        # parameter -> slice of the flat gradient buffer (launch order: {W_ll, b_ll} per layer, W1, b1, W2, b2)
        views: List[Tuple[Tensor, Tensor]] = []
        off = 0
        mparams = model._resident_params()
        for l in range(L):
            fin = F if l == 0 else H
            views.append((mparams[9 * l], self.grads[off: off + H * fin].view(H, fin))); off += H * fin
            views.append((mparams[9 * l + 1], self.grads[off: off + H])); off += H
        for p, n in zip(mparams[9 * L:], (H * H, H, C * H, C)):
            views.append((p, self.grads[off: off + n].view_as(p))); off += n
        assert off == P
        self.param_grads = views

    def bind_grads(self) -> None:
        has = {id(p) for p, _ in self.param_grads}
        for p in self.model.parameters():
            if id(p) not in has:
                p.grad = None
        for p, g in self.param_grads:
            p.grad = g

    def _job(self, xv_out: Optional[Tensor]) -> "_engine._VirtualJob":
        N, V, F, H, L, C, B = self.dims
        m = self.meta
        return _engine._VirtualJob(ptr(self.x_virtual), ptr(self.ei[VV]), self.ei[VV].size(1), ptr(self.ei[LV]),
                                   self.ei[LV].size(1), ptr(m.vptr), ptr(m.eptr_vv), ptr(m.eptr_lv),
                                   ctypes.cast(self._table, ctypes.c_void_p), ptr(xv_out), V, m.max_v, m.max_evv,
                                   self.slope, *([ptr(t) for t in self._state] if self._state is not None else [None] * 6))

    @property
    def advances_sync(self) -> bool:
        """Every ``run()`` adds one to word 0 of ``_sync`` on the device (the one-launch step with the virtual branch
        on its own workgroups does: its gradient fold advances the hand-off epoch) -- a per-step counter that a
        captured batch gather can read its slice number from (``DeviceHeteroDataset.gather_next(step)``)."""
        return bool(self.one_launch and self.idle_cus and self.virtual is not None and self.dims[4] >= 2
                    and self._sync is not None)

    def run(self) -> Tensor:
        """Issue the step on the current stream; returns ``loss`` (valid once the stream has run)."""
        N, V, F, H, L, C, B = self.dims
        m = self.meta
        W1, b1, W2, b2 = self._params[9 * L:]
        st = stream()
        ei_ll = self.ei[LL]
        E_ll = ei_ll.size(1)
        csr_rp, csr_col, dinv = self.csr
        bwd_args = (ptr(self.x_local), ptr(ei_ll), E_ll, ptr(m.lptr), ptr(m.eptr_ll), N, B, F, H, L, C,
                    self.head_act, self._wll_table, ptr(W1), ptr(W2), ptr(self.acts), ptr(self.pooled), ptr(self.z),
                    None, None, ptr(csr_rp), ptr(csr_col), ptr(dinv), m.max_n, m.max_ell, ptr(self.partials),
                    ptr(self.grads), ptr(m.flag), ctypes.byref(self._tail))
        if self.one_launch:
            with_v = self.idle_cus and self.virtual is not None
            call("hscn_resident_train_step" + self._sfx, ptr(self.x_local), ptr(ei_ll), E_ll, ptr(m.lptr),
                 ptr(m.eptr_ll), N, B, F, H, L, C, self.head_act, self._table, ptr(W1), ptr(b1), ptr(W2), ptr(b2),
                 m.max_n, m.max_ell, ptr(self.target), int(self.kind), ptr(self.pred), ptr(self.score),
                 ptr(self.partials), ptr(self.grads), ptr(self.acts) if with_v else None,
                 ptr(self._sync) if with_v else None, ptr(m.flag),
                 ctypes.byref(self._job(self.virtual)) if with_v else None,
                 ctypes.byref(self.structure.c) if self.structure is not None else None, st)
        elif self.defer:
            call("hscn_resident_fwd_with_virtual" + self._sfx, ptr(self.x_local), ptr(ei_ll), E_ll, ptr(m.lptr), ptr(m.eptr_ll),
                 N, B, F, H, L, C, self.head_act, self._table, ptr(W1), ptr(b1), ptr(W2), ptr(b2), m.max_n,
                 m.max_ell, ptr(self.acts), ptr(self.pooled), ptr(self.z), ptr(self.pred), ptr(self.score),
                 ptr(csr_rp), ptr(csr_col), ptr(dinv), ptr(m.flag), ctypes.byref(self._job(None)), st)
            call("hscn_resident_bwd_with_virtual" + self._sfx, *bwd_args, ctypes.byref(self._job(self.virtual)), st)
        else:
            cv = int(bool(self.model.compute_virtual))
            call("hscn_resident_fwd" + self._sfx, ptr(self.x_local), ptr(self.x_virtual), ptr(ei_ll), E_ll, ptr(self.ei[VV]),
                 self.ei[VV].size(1), ptr(self.ei[LV]), self.ei[LV].size(1), ptr(m.lptr), ptr(m.vptr),
                 ptr(m.eptr_ll), ptr(m.eptr_vv), ptr(m.eptr_lv), N, V, B, F, H, L, C, self.head_act, self.slope,
                 self._table, ptr(W1), ptr(b1), ptr(W2), ptr(b2), m.max_n, m.max_v, m.max_ell, m.max_evv, cv,
                 ptr(self.acts), ptr(self.pooled), ptr(self.z), ptr(self.pred), ptr(self.score),
                 ptr(self.virtual) if cv else None, ptr(csr_rp), ptr(csr_col), ptr(dinv), ptr(m.flag), st)
            call("hscn_resident_bwd" + self._sfx, *bwd_args, st)
        return self.loss

    def check(self) -> None:
        """Synchronising validity check of the launches issued so far (block-diagonal batch, sizes)."""
        self.meta.check()


class ScnWorkspace:
    """Output / scratch buffers of the stage-A launches sized for the largest step of a loop, shared by all of its
    ``ScnTrainStep`` objects (a step uses a prefix): one set of gradient buffers means one captured optimizer step
    serves every graph of the reference's one-optimizer-step-per-graph loop (train/train_clustering.py:36-50)."""

    def __init__(self, device, max_nodes: int, max_edges: int, max_graphs: int, F: int, H: int, K: int,
                 dtype=torch.float32):
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        N, E, B = max(int(max_nodes), 1), max(int(max_edges), 1), max(int(max_graphs), 1)
        self.cap = (N, E, B)
        self.S = torch.empty(N, K, **f32)
        self.y = torch.empty(N, H, dtype=dtype, device=device)
        self.stats = torch.empty(B, 4, **f32)
        self.ss = torch.empty(B, K, K, **f32)
        self.losses = torch.empty(3, **f32)
        self.ex = (torch.empty(N + B, **i32), torch.empty(E, **i32), torch.empty(N + B, **i32), torch.empty(E, **i32),
                   torch.empty(N, 16, **f32), torch.empty(N, **f32))
        P = int(_hip.lib().hscn_scn_resident_param_count(F, H, K))
        self.partials = torch.empty(B, P, **f32)
        self.grads = torch.zeros(P, **f32)
        self.one = torch.ones(1, **f32)
        self.ticket = torch.zeros(1, **i32)
        self.flag = torch.zeros(1, **i32)


class _ScnStructC(ctypes.Structure):      # include/hscn.h: hscn_scn_structure
    _fields_ = [("rowptr_d", ctypes.c_void_p), ("col_d", ctypes.c_void_p), ("rowptr_s", ctypes.c_void_p),
                ("col_s", ctypes.c_void_p), ("agg", ctypes.c_void_p), ("dout", ctypes.c_void_p), ("xpad", ctypes.c_void_p),
                ("ready", ctypes.c_int)]


class ScnStructurePool:
    """HBM for the per-batch structure (``hscn_scn_structure``) of every step of a stage-A loop, allocated once:
    ``take(N, E, B)`` hands a step its slices.  What it holds is derived from the graphs and the input features
    alone, so a step builds it on its first visit and loads it on the others (train/train_clustering.py:34 revisits
    every graph ``cluster_epochs`` times)."""

    def __init__(self, device, total_nodes: int, total_edges: int, total_graphs: int):
        i32 = dict(dtype=torch.int32, device=device)
        self._rp = [torch.empty(max(total_nodes + total_graphs, 1), **i32) for _ in range(2)]
        self._col = [torch.empty(max(total_edges, 1), **i32) for _ in range(2)]
        # (node-indexed arrays are sliced at the row-pointer offsets, which count one more word per graph)
        self._agg = torch.empty(max(total_nodes + total_graphs, 1), 16, dtype=torch.float32, device=device)
        self._dout = torch.empty(max(total_nodes + total_graphs, 1), dtype=torch.float32, device=device)
        self._xpad = torch.empty(max(total_nodes + total_graphs, 1), 16, dtype=torch.float32, device=device)
        self._n = self._e = 0

    def take(self, N: int, E: int, B: int) -> _ScnStructC:
        n, e = self._n, self._e
        if n + N + B > self._rp[0].numel() or e + E > self._col[0].numel():
            raise ValueError("the structure pool is smaller than the loop's graphs")
        self._n, self._e = n + N + B, e + E
        p = _hip.ptr
        return _ScnStructC(p(self._rp[0][n:]), p(self._col[0][e:]), p(self._rp[1][n:]), p(self._col[1][e:]),
                           p(self._agg[n:]), p(self._dout[n:]), p(self._xpad[n:]), 0)


class ScnTrainStep:
    """``optimizer.zero_grad(); S, mc, o = model.forward_graphs(data); (mc + o).backward()`` -- the body of
    the reference's clustering loop (train/train_clustering.py:37-49) for one ``Data`` graph or a block-diagonal
    ``Batch`` of raw graphs -- as the two graph-resident launches on preallocated buffers.

    Outputs refreshed by ``run()``: ``S`` [N,K] (softmax assignment), ``losses`` [3] = {mincut, ortho, their sum},
    ``grads`` (flat: W_rel, b_rel, W_root, W_mlp, b_mlp)."""

    def __init__(self, model, data, workspace: Optional[ScnWorkspace] = None, one_launch: Optional[bool] = None,
                 structure_pool: Optional["ScnStructurePool"] = None):
        """``one_launch``: forward, losses and backward of a graph in ONE workgroup program
        (``hscn_scn_resident_train_step``).  Default: whenever the graphs fit; ``False`` keeps the forward /
        backward pair.  ``structure_pool``: keep this batch's CSRs, out-degrees and ``A_hat x`` in HBM after the first
        ``run()`` and load them on later visits (one-launch step only; same results bit for bit)."""
        from .model.hscn import SCN
        if not isinstance(model, SCN):
            raise TypeError("ScnTrainStep drives graph_hscn.model.hscn.SCN")
        if not model.resident_ok(data):
            raise RuntimeError("the fused stage-A step needs the reference's SCN shape (mp_units=[H], mlp_units=[]) "
                               "and graphs that fit one CU's LDS")
        conv, lin = model.mp.module_0, list(model.mlp)[0]
        dev = conv.lin_rel.weight.device
        self.model = model
        self.meta = meta = _engine.scn_meta(data, dev)
        x = data.x if data.x.is_cuda else data.x.to(dev)
        self.x = (x if x.dtype == torch.float16 else x.float()).contiguous()
        self._sfx = _engine.storage_suffix(self.x.dtype)
        self.ei = (data.edge_index if data.edge_index.is_cuda else data.edge_index.to(dev)).contiguous()
        self.act = _engine.ACT[model.mp.act]
        self._mp = [conv.lin_rel.weight, conv.lin_rel.bias, conv.lin_root.weight, lin.weight, lin.bias]
        N, F = self.x.shape
        H, K = conv.lin_rel.weight.shape[0], lin.weight.shape[0]
        B, E = meta.num_graphs, self.ei.size(1)
        self.dims = (N, F, H, K, B, E)
        fits = bool(_hip.lib().hscn_scn_resident_train_step_supported(F, H, K, meta.max_n, meta.max_e))
        if one_launch and not fits:
            raise RuntimeError("one_launch=True: a graph of this batch does not fit the one-launch stage-A step")
        self.one_launch = fits if one_launch is None else bool(one_launch)
        ws = workspace if workspace is not None else ScnWorkspace(dev, N, E, B, F, H, K, self.x.dtype)
        if N > ws.cap[0] or E > ws.cap[1] or B > ws.cap[2] or ws.y.dtype != self.x.dtype:
            raise ValueError("the shared workspace is smaller than this step (or of another storage type)")
        self.workspace = ws
        if workspace is not None:        # the loop's steps share one ticket / flag word as well
            meta.ticket, meta.flag = ws.ticket, ws.flag
        elif meta.ticket is None:
            meta.ticket = ws.ticket
        self.S, self.y, self.stats, self.ss = ws.S[:N], ws.y[:N], ws.stats[:B], ws.ss[:B]
        self.losses, self.ex, self.partials, self.grads, self.one = ws.losses, ws.ex, ws.partials, ws.grads, ws.one
        P = int(self.grads.numel())
        views, off = [], 0
        for p in self._mp:
            views.append((p, self.grads[off: off + p.numel()].view_as(p)))
            off += p.numel()
        assert off == P
        self.param_grads = views
        self._one_args = {}
        self._cache = structure_pool.take(N, E, B) if (structure_pool is not None and self.one_launch) else None

    @property
    def loss(self) -> Tensor:
        return self.losses[2]

    def bind_grads(self) -> None:
        for p, g in self.param_grads:
            p.grad = g

    def fuses_optimizer(self, opt) -> bool:
        """``run(opt=opt)`` can apply ``opt`` (an ``optim.FlatAdam`` over THIS step's ``param_grads``) inside the
        step's own launch: one graph per step, the one-launch kernel, parameters that are their own storage."""
        return (self.one_launch and self.dims[4] == 1 and opt is not None and getattr(opt, "grads", None) is self.grads
                and all(p.is_contiguous() for p in self._mp)
                and [id(p) for p in getattr(opt, "params", [])] == [id(p) for p in self._mp])

    def run(self, opt=None) -> Tensor:
        """``opt``: ``optimizer.step()`` (train/train_clustering.py:50) in the tail of the same launch -- see
        ``fuses_optimizer``; the parameters are updated in place, ``grads`` still receives the gradient."""
        if self.one_launch:
            # the argument list of a step is the same every visit (device pointers of buffers allocated once, the
            # parameters' own storage): built on the first call, per optimizer; a visit is then ONE foreign call
            key = id(opt)
            hit = self._one_args.get(key)
            if hit is None or hit[0] != tuple(p.data_ptr() for p in self._mp):
                if opt is not None and not self.fuses_optimizer(opt):
                    raise ValueError("this step cannot carry the optimizer step (see ScnTrainStep.fuses_optimizer)")
                N, F, H, K, B, E = self.dims
                m = self.meta
                W = [p if p.is_contiguous() else None for p in self._mp]
                if any(w is None for w in W):
                    hit = None          # (non-contiguous parameters: the slow path below makes copies every call)
                else:
                    hit = (tuple(p.data_ptr() for p in self._mp),
                           (ptr(self.x), ptr(self.ei) if E else None, E, ptr(m.nptr), ptr(m.eptr), N, B, F, H, K,
                            self.act, *[ptr(w) for w in W], ptr(self.one), ptr(self.one), m.max_n, m.max_e,
                            ptr(self.S), ptr(self.stats), ptr(self.losses), ptr(m.ticket), ptr(self.partials),
                            ptr(self.grads), ptr(m.flag), ctypes.byref(opt.c) if opt is not None else None,
                            ctypes.byref(self._cache) if self._cache is not None else None))
                    self._one_args[key] = hit
            if hit is not None:
                call("hscn_scn_resident_train_step" + self._sfx, *hit[1], stream())
                if self._cache is not None:
                    self._cache.ready = 1          # (read at issue time: the launch in flight saw 0 and exports)
                return self.losses[2]
        N, F, H, K, B, E = self.dims
        m = self.meta
        W_rel, b_rel, W_root, W_mlp, b_mlp = (p.contiguous() for p in self._mp)
        st = stream()
        eip = ptr(self.ei) if E else None
        if opt is not None:
            raise ValueError("this step cannot carry the optimizer step (see ScnTrainStep.fuses_optimizer)")
        if self.one_launch:
            call("hscn_scn_resident_train_step" + self._sfx, ptr(self.x), eip, E, ptr(m.nptr), ptr(m.eptr), N, B, F, H,
                 K, self.act, ptr(W_rel), ptr(b_rel), ptr(W_root), ptr(W_mlp), ptr(b_mlp), ptr(self.one),
                 ptr(self.one), m.max_n, m.max_e, ptr(self.S), ptr(self.stats), ptr(self.losses), ptr(m.ticket),
                 ptr(self.partials), ptr(self.grads), ptr(m.flag), None,
                 ctypes.byref(self._cache) if self._cache is not None else None, st)
            if self._cache is not None:
                self._cache.ready = 1
            return self.losses[2]
        call("hscn_scn_resident_fwd" + self._sfx, ptr(self.x), eip, E, ptr(m.nptr), ptr(m.eptr), N, B, F, H, K, self.act,
             ptr(W_rel), ptr(b_rel), ptr(W_root), ptr(W_mlp), ptr(b_mlp), m.max_n, m.max_e, ptr(self.S), ptr(self.y),
             ptr(self.stats), ptr(self.ss), ptr(self.losses), ptr(m.ticket), *[ptr(t) for t in self.ex], ptr(m.flag), st)
        call("hscn_scn_resident_bwd" + self._sfx, ptr(self.x), eip, E, ptr(m.nptr), ptr(m.eptr), N, B, F, H, K, self.act,
             ptr(W_mlp), ptr(self.S), ptr(self.y), ptr(self.stats), ptr(self.ss), ptr(self.one), ptr(self.one),
             *[ptr(t) for t in self.ex], m.max_n, m.max_e, ptr(self.partials), ptr(self.grads), ptr(m.flag), st)
        return self.losses[2]

    def run_forward(self) -> Tensor:
        """The forward launch alone (the assignment pass, train/train_clustering.py:57-69): refreshes ``S``."""
        N, F, H, K, B, E = self.dims
        m = self.meta
        W_rel, b_rel, W_root, W_mlp, b_mlp = (p.contiguous() for p in self._mp)
        call("hscn_scn_resident_fwd" + self._sfx, ptr(self.x), ptr(self.ei) if E else None, E, ptr(m.nptr), ptr(m.eptr),
             N, B, F, H, K, self.act, ptr(W_rel), ptr(b_rel), ptr(W_root), ptr(W_mlp), ptr(b_mlp), m.max_n, m.max_e,
             ptr(self.S), ptr(self.y), ptr(self.stats), ptr(self.ss), ptr(self.losses), ptr(m.ticket),
             *([None] * 6), ptr(m.flag), stream())
        return self.S

    def check(self) -> None:
        self.meta.check()


class ScnEpochRunner:
    """The reference's stage-A loop (train/train_clustering.py:34-69: one optimizer step per graph, graph after
    graph, ``cluster_epochs`` times, then the assignment pass) driven from ONE foreign call
    (``hscn_scn_resident_train_epoch``): the dataset lies in HBM as one block-diagonal ``Batch``, ONE forward launch
    over it builds every graph's CSRs / out-degrees / ``A_hat x`` (``hscn_scn_structure``), the library then issues
    one launch per graph visit (step + optimizer in its tail, structure loaded) back to back, and ONE more forward
    launch over the dataset is the assignment pass.  The arithmetic of ``ScnTrainStep.run(opt=...)`` per graph, bit
    for bit; no per-graph Python objects, uploads or calls.

    ``eligible(...)`` says whether a model / dataset / optimizer qualifies (Adam or AdamW, the one-launch step's
    shapes)."""

    @staticmethod
    def eligible(model, big, optim_type: str) -> bool:
        from .model.hscn import SCN
        if not isinstance(model, SCN) or optim_type not in ("adam", "adamW") or not model.resident_ok(big):
            return False
        conv, lin = model.mp.module_0, list(model.mlp)[0]
        H, F = conv.lin_rel.weight.shape
        K = lin.weight.shape[0]
        meta = _engine.scn_meta(big, conv.lin_rel.weight.device)
        P = int(_hip.lib().hscn_scn_resident_param_count(F, H, K))
        return bool(_hip.lib().hscn_scn_resident_train_step_supported(F, H, K, meta.max_n, meta.max_e))

    def __init__(self, model, big, optim_type: str, lr: float, weight_decay: float):
        from .optim import FlatAdam
        conv, lin = model.mp.module_0, list(model.mlp)[0]
        dev = conv.lin_rel.weight.device
        self.model = model
        self.meta = meta = _engine.scn_meta(big, dev)
        x = big.x if big.x.is_cuda else big.x.to(dev)
        self.x = (x if x.dtype == torch.float16 else x.float()).contiguous()
        self._sfx = _engine.storage_suffix(self.x.dtype)
        self.ei = (big.edge_index if big.edge_index.is_cuda else big.edge_index.to(dev)).contiguous()
        self.act = _engine.ACT[model.mp.act]
        self._mp = [conv.lin_rel.weight, conv.lin_rel.bias, conv.lin_root.weight, lin.weight, lin.bias]
        if not all(p.is_contiguous() for p in self._mp):
            raise ValueError("the epoch kernel updates the parameters in their own storage: they must be contiguous")
        N, F = self.x.shape
        H, K = conv.lin_rel.weight.shape[0], lin.weight.shape[0]
        G, E = meta.num_graphs, self.ei.size(1)
        self.dims = (N, F, H, K, G, E)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.S = torch.empty(N, K, **f32)
        self._y = torch.empty(N, H, dtype=self.x.dtype, device=dev)
        self._stats = torch.empty(G, 4, **f32)
        self._ss = torch.empty(G, K, K, **f32)
        self.losses = torch.zeros(3, **f32)
        self._ticket = torch.zeros(1, **i32)
        self._one = torch.ones(1, **f32)
        self._cache_t = (torch.empty(N + G, **i32), torch.empty(max(E, 1), **i32), torch.empty(N + G, **i32),
                         torch.empty(max(E, 1), **i32), torch.empty(max(N, 1), 16, **f32), torch.empty(max(N, 1), **f32))
        P = int(_hip.lib().hscn_scn_resident_param_count(F, H, K))
        self.grads = torch.zeros(P, **f32)
        views, off = [], 0
        for p_ in self._mp:
            views.append((p_, self.grads[off: off + p_.numel()].view_as(p_)))
            off += p_.numel()
        self.param_grads = views
        self.optimizer = FlatAdam.from_config(optim_type, views, self.grads, lr, weight_decay)
        self._forward(export=True)                      # one launch: every graph's structure
        c = self._cache_t
        self._xpad = torch.zeros(max(N, 1), 16, **f32)  # the features as LDS holds them: float, 16 columns
        self._xpad[:N, :F] = self.x.float()
        self._cache = _ScnStructC(ptr(c[0]), ptr(c[1]), ptr(c[2]), ptr(c[3]), ptr(c[4]), ptr(c[5]), ptr(self._xpad), 1)

    def _forward(self, export: bool) -> None:
        N, F, H, K, G, E = self.dims
        m = self.meta
        W_rel, b_rel, W_root, W_mlp, b_mlp = self._mp
        ex = [ptr(t) for t in self._cache_t] if export else [None] * 6
        call("hscn_scn_resident_fwd" + self._sfx, ptr(self.x), ptr(self.ei) if E else None, E, ptr(m.nptr), ptr(m.eptr),
             N, G, F, H, K, self.act, ptr(W_rel), ptr(b_rel), ptr(W_root), ptr(W_mlp), ptr(b_mlp), m.max_n, m.max_e,
             ptr(self.S), ptr(self._y), ptr(self._stats), ptr(self._ss), ptr(self.losses), ptr(self._ticket), *ex,
             ptr(m.flag), stream())

    def run(self, visits: int) -> None:
        """``visits`` graph visits in dataset order (visit v takes graph v mod G).  One call; the launches are
        asynchronous."""
        N, F, H, K, G, E = self.dims
        m = self.meta
        call("hscn_scn_resident_train_epoch" + self._sfx, ptr(self.x), ptr(m.nptr), ptr(m.eptr), N, G, int(visits), F, H,
             K, self.act, *[ptr(p_) for p_ in self._mp], ptr(self._one), ptr(self._one), m.max_n, m.max_e,
             ctypes.byref(self._cache),
             ctypes.byref(self.optimizer.c), ptr(self.grads), ptr(self._stats), ptr(self.losses), ptr(self._ticket),
             ptr(m.flag), stream())

    def assign(self) -> Tensor:
        """The assignment pass (train/train_clustering.py:57-69) as ONE forward launch over the dataset: the soft
        assignments ``S`` [N, K] of all graphs with the weights as they are now."""
        self._forward(export=False)
        return self.S

    def check(self) -> None:
        self.meta.check()
        self.optimizer.check()
