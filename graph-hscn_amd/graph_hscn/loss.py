"""Loss selection of the training loop (reference graph_hscn/loss.py:6-19).
Plain torch ops on the [B, C] prediction: host-side glue, not a hot-path kernel
(component 6).  Quirk kept: the L1 branch scores with
``sigmoid(pred)`` (loss.py:17-19)."""
import torch
import torch.nn.functional as F


def criterion(loss_fn: str, pred: torch.Tensor, true: torch.Tensor):
    if loss_fn == "cross_entropy":
        if pred.ndim > 1 and true.ndim == 1:
            pred = F.log_softmax(pred, dim=-1)
            return F.nll_loss(pred, true), pred
        true = true.float()
        return F.binary_cross_entropy_with_logits(pred, true, reduction="mean"), torch.sigmoid(pred)
    return F.l1_loss(pred, true), torch.sigmoid(pred)
