"""Loss selection of the training loop (reference graph_hscn/loss.py:6-19).

On the device the multilabel BCE-with-logits and L1 branches are one fused HIP
launch (loss + sigmoid score + dL/dpred, csrc/loss.hip); the multiclass branch
(``true.ndim == 1``) and CPU tensors use the plain torch ops the reference
uses.  Quirk kept: the L1 branch scores with ``sigmoid(pred)`` (loss.py:17-19)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from ._hip import call, ptr, stream


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
        out = torch.empty_like(grad)
        call("hscn_scale", ptr(g_loss.reshape(1).contiguous()), ptr(grad), ptr(out), grad.numel(), stream())
        return out, None, None


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
