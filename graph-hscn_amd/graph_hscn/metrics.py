"""Epoch metrics of the training loop (reference graph_hscn/metrics.py:6-36), computed where the
tensors live: the reference moves every epoch's `[N, C]` labels and scores to the host and loops
over classes in sklearn; here the sort / cumulative sums run on the device (plain torch ops, they
are not on the hot path) and one scalar comes back.  Same definitions and error behaviour:
``eval_ap`` = mean over the classes that have both a positive and a negative label of sklearn's
``average_precision_score`` (step-wise integral of the precision-recall curve over distinct score
thresholds, NaN labels ignored); ``eval_mae`` = mean absolute error, raising on NaN predictions."""
from __future__ import annotations

import torch


def _ap_one(y: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """sklearn.metrics.average_precision_score for one binary column (float64)."""
    order = torch.argsort(s, descending=True, stable=True)
    s, y = s[order], y[order]
    tp = torch.cumsum(y, 0)
    n = torch.arange(1, y.numel() + 1, dtype=torch.float64, device=y.device)
    last = torch.ones_like(s, dtype=torch.bool)          # last element of every run of equal scores
    last[:-1] = s[1:] != s[:-1]
    precision = (tp / n)[last]
    recall = (tp / tp[-1])[last]
    prev = torch.cat([recall.new_zeros(1), recall[:-1]])
    return ((recall - prev) * precision).sum()


def eval_ap(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:      # metrics.py:6-27
    y_true = y_true.detach().to(torch.float64)
    y_pred = y_pred.detach().to(torch.float64)
    aps = []
    for i in range(y_true.shape[1]):
        col = y_true[:, i]
        if bool((col == 1).any()) and bool((col == 0).any()):
            labeled = col == col                                          # ignore NaN labels
            aps.append(_ap_one(col[labeled], y_pred[labeled, i]))
    if not aps:
        raise RuntimeError("No positively labeled data available. Cannot compute Average"
                           "Precision.")
    return float(torch.stack(aps).sum().item() / len(aps))


def eval_mae(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:     # metrics.py:30-36
    y_pred = y_pred.detach()
    if bool(torch.isnan(y_pred).any()):
        raise Exception("Model is predicting NaN.")
    d = (y_true.detach().to(torch.float64) - y_pred.to(torch.float64)).abs()
    return float(d.mean().item())
