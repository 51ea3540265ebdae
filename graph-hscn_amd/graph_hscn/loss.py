"""Loss selection of the training loop (reference graph_hscn/loss.py:6-19).

On the device the multilabel BCE-with-logits and L1 branches are one fused HIP
launch (loss + sigmoid score + dL/dpred, csrc/loss.hip) -- or none at all: a prediction
of the graph-resident HSCN forward arrives with its score, and the loss and its gradient
are evaluated inside the backward launch of the same step (``LazyLoss``); the multiclass branch
(``true.ndim == 1``) and CPU tensors use the plain torch ops the reference
uses.  Quirk kept: the L1 branch scores with ``sigmoid(pred)`` (loss.py:17-19)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from ._hip import call, ptr, stream


class LazyScaled(torch.Tensor):
    """``scale[0] * grad`` not yet multiplied out.  The loss node returns its input gradient in this
    form; a consumer that can apply the scalar itself (the graph-resident HSCN backward takes it as
    ``g_scale``) reads ``.grad_unscaled`` / ``.scale`` and no scaling launch happens; any other use
    (an ordinary torch op, a hook, gradient accumulation) dispatches through ``materialize``."""

    @staticmethod
    def __new__(cls, grad, scale):
        r = torch.Tensor._make_wrapper_subclass(cls, grad.shape, dtype=grad.dtype, device=grad.device,
                                                requires_grad=False)
        r.grad_unscaled = grad
        r.scale = scale
        r._dense = None
        return r

    def materialize(self) -> torch.Tensor:
        if self._dense is None:
            out = torch.empty_like(self.grad_unscaled)
            call("hscn_scale", ptr(self.scale), ptr(self.grad_unscaled), ptr(out), out.numel(), stream())
            self._dense = out
        return self._dense

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda t: t.materialize() if isinstance(t, LazyScaled) else t
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))

    def __repr__(self):
        return f"LazyScaled(shape={tuple(self.shape)})"


def _run_criterion(pred, true, kind, want_score=True):
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    score = torch.empty_like(pred) if want_score else None
    grad = torch.empty_like(pred)
    call("hscn_criterion_fwd", ptr(pred), ptr(true), pred.numel(), kind, ptr(loss), ptr(score), ptr(grad), stream())
    return loss.view(()), score, grad


class _LossState:
    """A loss whose evaluation was left to the backward launch of the graph-resident HSCN step
    (include/hscn.h: hscn_loss_tail).  ``value`` appears when that launch has been issued (``fill``); a
    read before that -- ``loss.item()`` ahead of ``backward()``, or no backward at all -- evaluates the
    loss with a launch of its own and keeps its gradient for the backward."""
    __slots__ = ("pred", "target", "kind", "value", "grad")

    def __init__(self, pred, target, kind):
        # VALUES only: the state outlives the step (``loss.detach()`` of a LazyLoss shares it, and a training loop
        # collects those, train/train.py:85), so it must not hold the prediction's autograd graph -- that kept every
        # iteration's saved activations alive until the epoch ended, and the parameters' gradient-accumulation
        # nodes with them (round 1: the precursor of the capture_end crash)
        self.pred, self.target, self.kind = pred.detach(), target.detach(), kind
        self.value = self.grad = None

    def fill(self, value):
        self.value = value

    def get(self):
        if self.value is None:
            self.value, _, self.grad = _run_criterion(self.pred, self.target, self.kind, want_score=False)
        return self.value


_ONES = {}


def root_grad(device):
    """Cached scalar 1 on ``device``: ``loss.backward(root_grad(dev))`` spares the fill launch of the
    implicit ``ones_like(loss)`` (4.5 us per step on a 50 us step)."""
    return _one(torch.device(device))


def _one(device):
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


class LazyLoss(torch.Tensor):
    """0-dim loss backed by a ``_LossState``: any use as a value dispatches through ``state.get()``;
    ``detach`` / ``alias`` stay lazy (a training loop may collect ``loss.detach()`` before ``backward()``),
    and ``ones_like`` -- the implicit root gradient of ``loss.backward()`` -- does not need the value."""

    @staticmethod
    def __new__(cls, state, device):
        r = torch.Tensor._make_wrapper_subclass(cls, (), dtype=torch.float32, device=device, requires_grad=False)
        r.state = state
        return r

    def materialize(self) -> torch.Tensor:
        return self.state.get()

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        if func in (torch.ops.aten.detach.default, torch.ops.aten.alias.default):
            return LazyLoss(args[0].state, args[0].device)
        if func is torch.ops.aten.ones_like.default and (kwargs or {}).get("dtype") in (None, torch.float32):
            return _one(args[0].device)        # (the implicit root gradient of loss.backward())
        un = lambda t: t.state.get() if isinstance(t, LazyLoss) else t
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))

    def __repr__(self):
        return f"LazyLoss({'pending' if self.state.value is None else float(self.state.value)})"


class LazyCriterionGrad(torch.Tensor):
    """``scale[0] * d criterion(pred, target) / d pred`` not yet evaluated: the graph-resident HSCN backward
    takes (pred, target, kind, scale) and evaluates it inside its launch; anything else gets the dense
    tensor through ``materialize``."""

    @staticmethod
    def __new__(cls, pred, target, kind, scale, state):
        r = torch.Tensor._make_wrapper_subclass(cls, pred.shape, dtype=pred.dtype, device=pred.device,
                                                requires_grad=False)
        r.pred, r.target, r.kind, r.scale, r.state = pred, target, kind, scale, state
        r._dense = None
        return r

    def materialize(self) -> torch.Tensor:
        if self._dense is None:
            self.state.get()
            out = torch.empty_like(self.state.grad)
            call("hscn_scale", ptr(self.scale), ptr(self.state.grad), ptr(out), out.numel(), stream())
            self._dense = out
        return self._dense

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda t: t.materialize() if isinstance(t, LazyCriterionGrad) else t
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))

    def __repr__(self):
        return f"LazyCriterionGrad(shape={tuple(self.shape)})"


class _TailCriterionFn(Function):
    """criterion on a prediction of the graph-resident HSCN forward (which already wrote the score):
    no launch here; the backward launch of the step evaluates the loss tail (csrc/resident.hip)."""

    @staticmethod
    def forward(ctx, pred, true, kind):
        ctx.state = _LossState(pred, true, kind)
        ctx.set_materialize_grads(False)
        return LazyLoss(ctx.state, pred.device)

    @staticmethod
    def backward(ctx, g_loss):
        if g_loss is None:
            return None, None, None
        st = ctx.state
        scale = g_loss.reshape(1).contiguous()
        if st.grad is not None:                  # the loss was read before the backward: its gradient exists
            return LazyScaled(st.grad, scale), None, None
        return LazyCriterionGrad(st.pred, st.target, st.kind, scale, st), None, None


class _CriterionFn(Function):
    @staticmethod
    def forward(ctx, pred, true, kind):
        loss, score, grad = _run_criterion(pred.contiguous(), true.contiguous(), kind)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(score)
        ctx.set_materialize_grads(False)  # no zero tensor (a fill launch) for the score output
        return loss, score

    @staticmethod
    def backward(ctx, g_loss, _g_score):
        (grad,) = ctx.saved_tensors
        if g_loss is None:
            return None, None, None
        return LazyScaled(grad, g_loss.reshape(1).contiguous()), None, None


def criterion(loss_fn: str, pred: torch.Tensor, true: torch.Tensor):
    multiclass = loss_fn == "cross_entropy" and pred.ndim > 1 and true.ndim == 1
    if pred.is_cuda and not multiclass and pred.dtype == torch.float32 and pred.shape == true.shape:
        kind = 0 if loss_fn == "cross_entropy" else 1
        true = true.float()
        score = getattr(pred, "_hscn_score", None)
        if score is not None:                                   # (score, version of pred it belongs to)
            score = score[0] if score[1] == pred._version else None
        if (score is not None and pred.requires_grad and torch.is_grad_enabled() and pred.dim() == 2
                and true.is_contiguous() and pred.is_contiguous() and true.device == pred.device):
            _one(pred.device)          # the root gradient of loss.backward(), created outside any capture
            return _TailCriterionFn.apply(pred, true, kind), score
        return _CriterionFn.apply(pred, true, kind)
    if loss_fn == "cross_entropy":
        if multiclass:
            pred = F.log_softmax(pred, dim=-1)
            return F.nll_loss(pred, true), pred
        true = true.float()
        return F.binary_cross_entropy_with_logits(pred, true, reduction="mean"), torch.sigmoid(pred)
    return F.l1_loss(pred, true), torch.sigmoid(pred)
