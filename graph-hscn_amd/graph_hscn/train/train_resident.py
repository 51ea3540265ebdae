"""Stage C training loop at the speed of the kernels (extension; the reference's loop is train/train.py:54-214,
mirrored in ``train.py`` next to this file).

Same schedule -- epochs of shuffled mini-batches, an optimizer step per batch, evaluation every ``eval_period``
epochs, early stopping on the validation loss -- but the training split lives in HBM
(``loader.device_dataset.DeviceHeteroDataset``), a step is "permutation slice -> one gather launch -> one replay of
the captured forward + loss + backward + optimizer step", and the host reads nothing back until the epoch ends.
The last, shorter batch of an epoch (the reference's loader keeps it, loader/hetero_data.py:96-104) runs through
the eager path on a host-collated batch with the same optimizer, so every graph is visited once per epoch.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Sequence

import torch

from ..config.config import OPTIM_DICT
from ..data import HeteroBatch, HeteroData
from ..loader.device_dataset import DeviceHeteroDataset
from ..loss import criterion
from ..replay import CapturedStep
from .train import eval_epoch, is_eval_epoch


def fit_resident(logger, optim_cfg, training_cfg, train_graphs: Sequence[HeteroData], eval_loaders: Sequence, model,
                 batch_size: int, metric_fn: Optional[Callable] = None, seed: int = 0, reducer=None,
                 flat_optimizer: bool = True) -> List[tuple]:
    """Returns ``[(mean train loss, train metric), ...]`` per epoch, like ``train.train``.  ``eval_loaders`` =
    ``[validation, test]`` loaders of host batches (evaluated with ``train.eval_epoch``).

    Data parallel: every rank calls this with ITS shard of the training graphs (``distributed.shard_list``; equal
    shard sizes, so that all ranks take the same number of steps) and a ``distributed.FlatGradReducer`` as
    ``reducer`` (built with ``equal_weights=True`` it averages with no scaling launch).
    The iteration stays ONE replay: forward + loss + backward, the RCCL all-reduce of the flat gradient buffer where
    the backward left it, the optimizer step -- all captured (an optimizer without a capturable step is stepped, and
    the collective issued, outside the graph).  The eager tail batch is reduced the same way."""
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("fit_resident runs on the MI355X HIP path: move the model to 'cuda'")
    # what the captured iteration does not implement must not be dropped silently (train/train.py:89-95 honours both)
    if int(getattr(optim_cfg, "batch_accumulation", 1) or 1) != 1:
        raise NotImplementedError("fit_resident steps the optimizer every batch: batch_accumulation != 1 needs "
                                  "train.train (the reference-shaped loop)")
    if getattr(optim_cfg, "clip_grad_norm", False):
        raise NotImplementedError("fit_resident has no gradient clipping between the captured backward and the "
                                  "optimizer step: use train.train for clip_grad_norm")
    G, B = len(train_graphs), int(batch_size)
    if G < B:
        raise ValueError("fewer training graphs than one batch")
    ds = DeviceHeteroDataset(train_graphs, dev, B)
    opt_cls = OPTIM_DICT[optim_cfg.optim_type]
    kw = dict(lr=optim_cfg.lr, weight_decay=optim_cfg.weight_decay)
    flat = flat_optimizer and optim_cfg.optim_type in ("adam", "adamW")   # optim.FlatAdam: the update as ONE launch
    if flat:
        from ..optim import FlatAdam
        optimizer = lambda st: FlatAdam.from_config(optim_cfg.optim_type, st.param_grads, st.grads, **kw)  # noqa: E731
        capturable = True
    else:
        try:
            optimizer = opt_cls(model.parameters(), capturable=True, fused=True, **kw)
        except (TypeError, RuntimeError):          # (Adagrad has neither switch: its step stays outside the graph)
            optimizer = opt_cls(model.parameters(), **kw)
        capturable = bool(optimizer.defaults.get("capturable", False))
    in_graph = capturable      # the whole iteration -- backward, gradient all-reduce (if any), optimizer step -- is one graph
    gen = torch.Generator(device=dev).manual_seed(seed)
    model.train()
    model.engine = "resident"
    ds.new_epoch(gen)
    # the gather of the next permutation slice is captured in front of the step: a replay = next batch + iteration
    step = CapturedStep(model, ds.static, training_cfg.loss_fn, optimizer=optimizer if in_graph else None,
                        pre=ds.gather_next, reducer=reducer if in_graph else None)
    if flat:
        optimizer = step.optimizer
    steps, tail = G // B, G % B
    C = ds.C
    loss_log = torch.zeros(steps + (1 if tail else 0), dtype=torch.float32, device=dev)
    scores = torch.zeros(G, C, dtype=torch.float32, device=dev) if metric_fn else None
    targets = torch.zeros(G, C, dtype=torch.float32, device=dev) if metric_fn else None
    history, best, stale = [], float("inf"), 0
    for epoch in range(training_cfg.epochs):
        start = time.time()
        model.train()
        perm = ds.new_epoch(gen)               # permutation + batch counter on the device
        if not in_graph:
            step.bind_grads()                  # (the eager tail of the previous epoch re-pointed p.grad)
        for i in range(steps):
            step.replay()
            if reducer is not None and not in_graph:
                reducer.reduce(float(B), float(B * reducer.world_size))
            if not in_graph:
                optimizer.step()
            loss_log[i].copy_(step.loss)
            if metric_fn:
                scores[i * B:(i + 1) * B].copy_(step.score)
                targets[i * B:(i + 1) * B].copy_(ds.static.batch["local"].y)
        if tail:
            hb = HeteroBatch.from_data_list([train_graphs[j] for j in perm[steps * B:].tolist()]).to(dev)
            optimizer.zero_grad(set_to_none=True)
            pred = model(hb.x_dict, hb.edge_index_dict, hb)
            loss, score = criterion(training_cfg.loss_fn, pred, hb["local"].y)
            loss.backward()
            if reducer is not None:
                reducer.reduce(float(tail), float(tail * reducer.world_size))
            optimizer.step_from_autograd() if flat else optimizer.step()
            loss_log[steps].copy_(loss.detach())
            if metric_fn:
                scores[steps * B:].copy_(score.detach())
                targets[steps * B:].copy_(hb["local"].y)
            del pred, loss, score, hb
        mean_loss = float(loss_log.mean().item())                     # the epoch's only read-back
        perf = metric_fn(targets, scores) if metric_fn else float("nan")
        history.append((mean_loss, perf))
        if logger is not None:
            logger.info(f"epoch {epoch} train loss {mean_loss:.5f} perf {perf:.5f} ({time.time() - start:.2f}s)")
        if is_eval_epoch(epoch, training_cfg.epochs, training_cfg.eval_period):
            for split, loader in zip(["Validation", "Test"], eval_loaders):
                vloss, _ = eval_epoch(epoch, logger, loader, model, training_cfg.loss_fn, metric_fn, split)
                if split == "Validation":
                    if reducer is not None and reducer.world_size > 1:
                        # every rank must take the same stop decision (the next collective would hang otherwise):
                        # the ranks agree on the mean of their validation losses
                        import torch.distributed as dist
                        t = torch.tensor([vloss], dtype=torch.float64, device=dev)
                        dist.all_reduce(t, group=reducer.group)
                        vloss = float(t.item()) / reducer.world_size
                    if vloss < best - training_cfg.min_delta:
                        best, stale = vloss, 0
                    else:
                        stale += 1
                    if stale >= training_cfg.patience and epoch != training_cfg.epochs - 1:
                        if logger is not None:
                            logger.info("stopping early")
                        ds.check()
                        return history
    ds.check()
    return history
