"""hipGraph replay of the training step across DIFFERENT batches.

A captured step replays only if every pointer and every host-side size it was captured with stays
valid.  The graph-resident kernels read the per-graph node / edge ranges from device arrays, so the
only host constants are capacities: ``StaticHeteroBatch`` owns device buffers sized for the largest
batch of a loader (nodes, edges per relation, virtual nodes; per-graph maxima that size the LDS), each
step's batch is copied INTO them (``load``), and the captured launches run on whatever the buffers
hold.  ``CapturedStep`` captures ``zero grads -> HSCN.forward -> criterion -> backward`` once (as direct C-ABI
launches, ``step.ResidentTrainStep``: no autograd inside the capture); afterwards
a training iteration is ``static.load(batch); step.replay(); optimizer.step()``.

The reference loop (train/train.py:73-95) issues ~150 small launches per step from Python; the eager
path here issues 4 but is still host-bound -- replay is what lets the GPU, not
the interpreter, set the pace of an epoch.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
from torch import Tensor

from .data import HeteroBatch

LL = ("local", "to", "local")
VV = ("virtual", "to", "virtual")
LV = ("local", "to", "virtual")


class StaticHeteroBatch:
    """Fixed-capacity device buffers with the ``HeteroBatch`` protocol (``.batch`` is the object to hand
    to the model).  Only for the graph-resident engine: edge lists are valid inside the per-graph ranges
    of ``ptr32`` only, the tail of every buffer is stale data of earlier, larger batches.

    All fields are views of ONE flat byte buffer, so a batch that was ``pack``-ed ahead of time (on the
    device or in pinned host memory) is loaded by a single copy."""

    def __init__(self, batches: Iterable[HeteroBatch], device, num_classes: Optional[int] = None,
                 feature_dtype=torch.float32):
        """``feature_dtype``: storage type of the node features in the static buffers (``torch.float16``: the
        half-storage mode of the graph-resident engine; ``load`` converts)."""
        self.feature_dtype = feature_dtype
        batches = list(batches)
        if not batches:
            raise ValueError("need at least one batch to size the buffers")
        B = {int(b.num_graphs) for b in batches}
        if len(B) != 1:
            raise ValueError("all batches must hold the same number of graphs (drop or pad the last one)")
        cap = lambda f: max(int(f(b)) for b in batches)
        y0 = batches[0]["local"].y if "y" in batches[0]["local"] else None
        C = None if y0 is None else (int(y0.size(1)) if num_classes is None else int(num_classes))
        self._allocate(B.pop(), device, cap(lambda b: b["local"].num_nodes), cap(lambda b: b["virtual"].num_nodes),
                       {et: cap(lambda b, et=et: b[et].edge_index.size(1)) for et in (LL, VV, LV)},
                       {"local": cap(lambda b: b["local"].max_nodes), "virtual": cap(lambda b: b["virtual"].max_nodes)},
                       {et: cap(lambda b, et=et: b[et].max_edges) for et in (LL, VV, LV)},
                       int(batches[0]["local"].x.size(1)), C)

    @classmethod
    def from_capacities(cls, num_graphs: int, device, num_nodes: int, num_virtual: int, num_edges: dict,
                        max_nodes: dict, max_edges: dict, num_features: int, num_classes: Optional[int]):
        """Buffers for batches of ``num_graphs`` graphs with at most the given totals (nodes, virtual nodes,
        edges per relation) and per-graph maxima (what sizes the LDS of the graph-resident launches)."""
        self = cls.__new__(cls)
        self.feature_dtype = torch.float32
        self._allocate(num_graphs, device, num_nodes, num_virtual, num_edges, max_nodes, max_edges, num_features,
                       num_classes)
        return self

    def _allocate(self, G, device, N, V, E, max_nodes, max_edges, F, C):
        self.num_graphs = int(G)
        self.device = torch.device(device)
        self.N, self.V = int(N), int(V)
        self.E = {et: max(int(E[et]), 1) for et in (LL, VV, LV)}
        self.max_nodes = {k: int(v) for k, v in max_nodes.items()}
        self.max_edges = {et: int(max_edges[et]) for et in (LL, VV, LV)}
        fdt = getattr(self, "feature_dtype", torch.float32)
        fields = [("x_local", fdt, (self.N, F)), ("x_virtual", fdt, (self.V, F))]
        for nt, n in (("local", self.N), ("virtual", self.V)):
            fields += [(f"ptr_{nt}", torch.int64, (G + 1,)), (f"ptr32_{nt}", torch.int32, (G + 1,)),
                       (f"batch_{nt}", torch.int64, (n,))]
        if C is not None:
            fields.append(("y", torch.float32, (G, C)))
        for i, et in enumerate((LL, VV, LV)):
            fields += [(f"ei_{i}", torch.int64, (2, self.E[et])), (f"eptr_{i}", torch.int32, (G + 1,))]
        self._layout, off = [], 0
        for name, dt, shape in fields:
            nbytes = torch.empty((), dtype=dt).element_size()
            for d in shape:
                nbytes *= d
            self._layout.append((name, dt, shape, off, nbytes))
            off += (nbytes + 15) // 16 * 16
        self.nbytes = off
        self.flat = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.batch = self._view(self.flat)

    def _view(self, flat: Tensor) -> HeteroBatch:
        t = {name: flat[off: off + nb].view(dt).view(shape) for name, dt, shape, off, nb in self._layout}
        hb = HeteroBatch()
        hb.num_graphs = self.num_graphs
        for nt, n in (("local", self.N), ("virtual", self.V)):
            st = hb[nt]
            st.x = t[f"x_{nt}"]
            st.ptr, st.ptr32, st.batch = t[f"ptr_{nt}"], t[f"ptr32_{nt}"], t[f"batch_{nt}"]
            st.num_nodes = n
            st.max_nodes = self.max_nodes[nt]
        if "y" in t:
            hb["local"].y = t["y"]
        for i, et in enumerate((LL, VV, LV)):
            hb[et].edge_index = t[f"ei_{i}"]
            hb[et].ptr32 = t[f"eptr_{i}"]
            hb[et].max_edges = self.max_edges[et]
        return hb

    def _fill(self, dst: HeteroBatch, src: HeteroBatch) -> None:
        if int(src.num_graphs) != self.num_graphs:
            raise ValueError("batch holds a different number of graphs than the static buffers")
        for nt, capn in (("local", self.N), ("virtual", self.V)):
            s, d = src[nt], dst[nt]
            n = int(s.num_nodes)
            if n > capn or int(s.max_nodes) > self.max_nodes[nt]:
                raise ValueError(f"batch exceeds the static capacity of '{nt}' nodes")
            d.x[:n].copy_(s.x, non_blocking=True)
            d.batch[:n].copy_(s.batch, non_blocking=True)
            d.ptr.copy_(s.ptr, non_blocking=True)
            d.ptr32.copy_(s.ptr32, non_blocking=True)
        if "y" in src["local"] and "y" in dst["local"]:
            dst["local"].y.copy_(src["local"].y, non_blocking=True)
        for et in (LL, VV, LV):
            s, d = src[et], dst[et]
            e = int(s.edge_index.size(1))
            if e > self.E[et] or int(s.max_edges) > self.max_edges[et]:
                raise ValueError(f"batch exceeds the static capacity of relation {et}")
            d.edge_index[:, :e].copy_(s.edge_index, non_blocking=True)
            d.ptr32.copy_(s.ptr32, non_blocking=True)

    def pack(self, src: HeteroBatch, device=None, pin_memory: bool = False) -> Tensor:
        """``src`` laid out like the static buffers, as one flat byte tensor (on ``device``, default: the
        static buffers' device; or in pinned host memory): what ``load`` takes with a single copy."""
        dev = torch.device(device) if device is not None else self.device
        flat = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev, pin_memory=(pin_memory and dev.type == "cpu"))
        self._fill(self._view(flat), src)
        return flat

    def load(self, src) -> HeteroBatch:
        """Make the static buffers hold ``src``: a packed byte tensor (one copy) or a ``HeteroBatch`` on the
        host or the device (one copy per field).  Asynchronous on the current stream."""
        if isinstance(src, Tensor):
            if src.dtype != torch.uint8 or src.numel() != self.nbytes:
                raise ValueError("not a batch packed by this StaticHeteroBatch")
            self.flat.copy_(src, non_blocking=True)
        else:
            self._fill(self.batch, src)
        return self.batch


def capture_optimizer_step(params, optimizer, warmup: int = 3) -> "torch.cuda.CUDAGraph":
    """``optimizer.step()`` (an optimizer built with ``capturable=True``; ``fused=True`` keeps it to one launch) on
    the gradient buffers ``p.grad`` points at NOW, as a hipGraph of its own.  The warm-up steps PyTorch's capture
    recipe asks for are undone: parameters and optimizer state are put back in place (state the warm-up created is
    zeroed: the initial state of the Adam family)."""
    params = list(params)
    snap_p = [p.detach().clone() for p in params]
    snap_s = {id(p): {k: v.clone() for k, v in st.items() if isinstance(v, Tensor)} for p, st in optimizer.state.items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            optimizer.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        optimizer.step()
    with torch.no_grad():
        for p, s0 in zip(params, snap_p):
            p.copy_(s0)
        for p, st in optimizer.state.items():
            for k, v in st.items():
                if isinstance(v, Tensor):
                    old = snap_s.get(id(p), {}).get(k)
                    v.copy_(old) if old is not None else v.zero_()
    return g


class CapturedStep:
    """``zero grads -> model(batch) -> criterion -> backward`` captured once on ``static.batch``.
    ``replay()`` runs it on whatever was last loaded; ``loss`` / ``pred`` / ``score`` are the captured
    output tensors (refreshed by every replay), parameter gradients are slices of one flat buffer
    (``grads``; ``p.grad`` points into it).

    The captured region holds kernel nodes only: the step is issued through ``step.ResidentTrainStep``
    (direct C-ABI launches on preallocated buffers), never through the autograd engine.  Round 1 captured
    ``loss.backward()``; autograd's gradient-accumulation nodes are bound to the stream of the first step that
    used them, a node kept alive by ANY earlier eager ``loss`` / ``pred`` tied the capture to that
    non-capturing stream and ``capture_end`` died inside the HIP runtime (a host SIGSEGV).  With no autograd
    in the capture, live tensors of earlier eager steps and earlier ``CapturedStep`` objects on the same model
    cannot reach into it (tests/test_gpu_step.py)."""

    def __init__(self, model, static: StaticHeteroBatch, loss_fn: str, warmup: int = 3, optimizer=None, pre=None,
                 one_launch: Optional[bool] = None, reducer=None, structure=None):
        """``one_launch``: passed to ``step.ResidentTrainStep`` (None: the one-launch step whenever the batch fits it;
        False: the forward + backward launch pair, whose gradients are bit-identical to the eager autograd path).
        ``reducer``: a ``distributed.FlatGradReducer``; its all-reduce of the flat gradient buffer (RCCL kernels are
        capturable) is captured between the backward and the optimizer step, so one replay per iteration is all a
        rank issues -- no eager collective launch between replays.  Every rank must hold the same number of graphs
        (``static.num_graphs``): the reduction is the mean over ranks.
        ``structure``: passed to ``step.ResidentTrainStep`` ("batch": the step loads the graphs' CSRs / degree norms
        that ``DeviceHeteroDataset(resident_structure=True)`` gathers with every batch instead of rebuilding them).
        ``pre``: a callable captured in front of the step that refreshes the static buffers from device-side
        state only (``DeviceHeteroDataset.gather_next``); note that the warm-up iterations and the capture call
        it too (rewind with ``new_epoch`` afterwards).
        ``optimizer``: a ``torch.optim`` optimizer built with ``capturable=True`` (``fused=True`` keeps it to two
        launches), or a callable ``step -> optim.FlatAdam`` (built on the step's flat gradient buffer: ONE launch;
        afterwards ``self.optimizer``); its ``step()`` is captured behind the backward, so a replay is a whole
        training iteration.
        The ``warmup`` eager iterations that precede the capture run the optimizer too (PyTorch's whole-network
        capture recipe); parameters and optimizer state are put back afterwards, IN PLACE (the captured launches
        hold their addresses): state that existed before is restored, state the warm-up created is zeroed (the
        initial state of the Adam family)."""
        from .step import ResidentTrainStep
        self.model, self.static, self.loss_fn, self.optimizer = model, static, loss_fn, optimizer
        snap_p = snap_s = None
        make_flat = callable(optimizer) and not hasattr(optimizer, "step")     # ``lambda step: optim.FlatAdam(...)``
        if optimizer is not None:
            snap_p = [p.detach().clone() for p in model.parameters()]
            if not make_flat:
                snap_s = {id(p): {k: v.clone() for k, v in st.items() if isinstance(v, Tensor)}
                          for p, st in optimizer.state.items()}
        hb = static.batch
        if "y" not in hb["local"]:
            raise ValueError("the static batch carries no targets")
        try:
            self.step = ResidentTrainStep(model, hb, loss_fn, one_launch=one_launch, structure=structure)
        except RuntimeError as e:
            raise RuntimeError("CapturedStep needs the graph-resident engine (the layered operators size their "
                               "work by tensor shapes, which a static-capacity batch does not carry): " + str(e)) from e
        self.step.bind_grads()
        model.last_engine = "resident"
        if make_flat:          # built here: it needs the step's flat gradient buffer
            optimizer = self.optimizer = optimizer(self.step)

        pre_takes_step = False
        if pre is not None:
            import inspect
            try:
                pre_takes_step = "step" in inspect.signature(pre).parameters
            except (TypeError, ValueError):
                pass

        def step():
            if pre is not None:
                pre(self.step) if pre_takes_step else pre()      # (DeviceHeteroDataset.gather_next(step))
            self.step.run()
            if reducer is not None:
                reducer.reduce(float(static.num_graphs), float(static.num_graphs * reducer.world_size))
            if optimizer is not None:
                optimizer.step()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step()
        self.pred, self.loss, self.score = self.step.pred, self.step.loss, self.step.score
        # the gradient tensors the captured backward writes (a later eager step re-points ``p.grad`` elsewhere)
        self.grads = list(self.step.param_grads)
        if optimizer is not None:
            with torch.no_grad():
                for p, s0 in zip(model.parameters(), snap_p):
                    p.copy_(s0)
                if make_flat:
                    optimizer.reset_state()
                else:
                    for p, st in optimizer.state.items():
                        for k, v in st.items():
                            if isinstance(v, Tensor):
                                old = snap_s.get(id(p), {}).get(k)
                                v.copy_(old) if old is not None else v.zero_()

    def replay(self) -> Tensor:
        self.graph.replay()
        return self.loss

    def bind_grads(self) -> None:
        """Point every ``p.grad`` at the captured step's gradient buffers again (after eager steps of the same
        model): an optimizer stepped OUTSIDE the graph reads ``p.grad``."""
        self.step.bind_grads()
