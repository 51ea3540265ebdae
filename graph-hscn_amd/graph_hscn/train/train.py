"""Stage C training loop with the reference's structure
(/root/reference/graph_hscn/train/train.py:54-214): epoch loop, gradient
accumulation, optional clipping, eval + early stopping.  Differences: wandb is
optional, ``loss.item()`` is read once per epoch instead of once per iteration
(train.py:85 syncs the device every step), and an optional
``FlatGradReducer`` all-reduces gradients on stepping iterations only."""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch
import torch.nn as nn

from ..config.config import OPTIM_DICT
from ..loss import criterion
from ..model.hscn import HSCN


def is_eval_epoch(epoch: int, max_epochs: int, eval_period: int) -> bool:  # train/utils.py:1-6
    return (epoch + 1) % eval_period == 0 or epoch == 0 or (epoch + 1) == max_epochs


def _run_batch(model, batch, device):
    if isinstance(model, HSCN):
        batch = batch.to(device)
        return model(batch.x_dict, batch.edge_index_dict, batch), batch["local"].y
    # train.py:78-80 leaves the batch where the loader put it; the HIP operators take device tensors only
    batch = batch.to(device)
    batch.x = batch.x.float()
    return model(batch), batch.y


def train_epoch(epoch, logger, loader, model, optimizer, loss_fn: str, metric_fn: Optional[Callable],
                batch_accumulation: int, clip_grad_norm: bool, reducer=None):
    start = time.time()
    model.train()
    optimizer.zero_grad()
    device = next(model.parameters()).device
    losses, y_true, y_pred = [], [], []
    num = len(loader)
    for it, batch in enumerate(loader):
        pred, true = _run_batch(model, batch, device)
        loss, score = criterion(loss_fn, pred, true)
        y_true.append(true)
        y_pred.append(score.detach())
        losses.append(loss.detach())
        loss.backward()
        if (it + 1) % batch_accumulation == 0 or it + 1 == num:
            if reducer is not None:
                reducer.reduce(float(pred.size(0)))
            if clip_grad_norm:
                nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            optimizer.step()
            optimizer.zero_grad()
    mean_loss = float(torch.stack(losses).mean().item())
    perf = metric_fn(torch.cat(y_true), torch.cat(y_pred)) if metric_fn else float("nan")
    if logger is not None:
        logger.info(f"epoch {epoch} train loss {mean_loss:.5f} perf {perf:.5f} ({time.time() - start:.2f}s)")
    return mean_loss, perf


@torch.no_grad()
def eval_epoch(epoch, logger, loader, model, loss_fn: str, metric_fn: Optional[Callable], split: str):
    model.eval()
    device = next(model.parameters()).device
    losses, y_true, y_pred = [], [], []
    for batch in loader:
        pred, true = _run_batch(model, batch, device)
        loss, score = criterion(loss_fn, pred, true)
        y_true.append(true)
        y_pred.append(score)
        losses.append(loss)
    mean_loss = float(torch.stack(losses).mean().item())
    perf = metric_fn(torch.cat(y_true), torch.cat(y_pred)) if metric_fn else float("nan")
    if logger is not None:
        logger.info(f"epoch {epoch} {split} loss {mean_loss:.5f} perf {perf:.5f}")
    return mean_loss, perf


def train(logger, optim_cfg, training_cfg, loaders, model, metric_fn: Optional[Callable] = None, reducer=None):
    optimizer = OPTIM_DICT[optim_cfg.optim_type](lr=optim_cfg.lr, weight_decay=optim_cfg.weight_decay,
                                                 params=model.parameters())
    best, stale = float("inf"), 0
    history = []
    for epoch in range(training_cfg.epochs):
        history.append(train_epoch(epoch, logger, loaders[0], model, optimizer, training_cfg.loss_fn, metric_fn,
                                   optim_cfg.batch_accumulation, optim_cfg.clip_grad_norm, reducer))
        if is_eval_epoch(epoch, training_cfg.epochs, training_cfg.eval_period):
            for split, loader in zip(["Validation", "Test"], loaders[1:]):
                loss, _ = eval_epoch(epoch, logger, loader, model, training_cfg.loss_fn, metric_fn, split)
                if split == "Validation":
                    if loss < best - training_cfg.min_delta:
                        best, stale = loss, 0
                    else:
                        stale += 1
                    if stale >= training_cfg.patience and epoch != training_cfg.epochs - 1:
                        if logger is not None:
                            logger.info("stopping early")
                        return history
    return history
