"""Loss selection of the training loop (reference graph_hscn/loss.py:6-19).

On the device the multilabel BCE-with-logits and L1 branches are one fused HIP
launch (loss + sigmoid score + dL/dpred, csrc/loss.hip); the multiclass branch
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


class _CriterionFn(Function):
    @staticmethod
    def forward(ctx, pred, true, kind):
        pred = pred.contiguous()
        true = true.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        score = torch.empty_like(pred)
        grad = torch.empty_like(pred)
        call("hscn_criterion_fwd", ptr(pred), ptr(true), pred.numel(), kind, ptr(loss), ptr(score), ptr(grad),
             stream())
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(score)
        ctx.set_materialize_grads(False)  # no zero tensor (a fill launch) for the score output
        return loss.view(()), score

    @staticmethod
    def backward(ctx, g_loss, _g_score):
        (grad,) = ctx.saved_tensors
        if g_loss is None:
            return None, None, None
        return LazyScaled(grad, g_loss.reshape(1).contiguous()), None, None


def criterion(loss_fn: str, pred: torch.Tensor, true: torch.Tensor):
    multiclass = loss_fn == "cross_entropy" and pred.ndim > 1 and true.ndim == 1
    if pred.is_cuda and not multiclass and pred.dtype == torch.float32 and pred.shape == true.shape:
        return _CriterionFn.apply(pred, true.float(), 0 if loss_fn == "cross_entropy" else 1)
    if loss_fn == "cross_entropy":
        if multiclass:
            pred = F.log_softmax(pred, dim=-1)
            return F.nll_loss(pred, true), pred
        true = true.float()
        return F.binary_cross_entropy_with_logits(pred, true, reduction="mean"), torch.sigmoid(pred)
    return F.l1_loss(pred, true), torch.sigmoid(pred)
