"""Stage A driver: fit the spectral-clustering net with MinCUT + orthogonality
losses, then hard-assign every node (reference
/root/reference/graph_hscn/train/train_clustering.py:20-70).

``batch_graphs=1`` (default) is the reference's trajectory: ``gcn_norm`` with self
loops, one optimizer step PER GRAPH (:36-50), then an assignment pass (:57-69, run
under ``no_grad`` here -- the reference builds autograd graphs it never uses).
``batch_graphs>1`` is an extension: B graphs per step on the device as one
block-diagonal batch; the loss is the mean of the per-graph losses (exactly
``dense_mincut_pool``'s own mean over its batch dimension), which changes the
optimisation trajectory and shards across GPUs like stage C.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .. import _hip
from ..config.config import OPTIM_DICT
from ..data import Batch
from ..loss import root_grad
from ..model.hscn import SCN
from ..nn.pool import gcn_norm


def _assign(S: torch.Tensor) -> torch.Tensor:
    ids = torch.empty(S.size(0), dtype=torch.int64, device=S.device)
    _hip.call("hscn_assign_argmax", _hip.ptr(S.contiguous()), _hip.ptr(ids), S.size(0), S.size(1), _hip.stream())
    return ids


def _forward(model: SCN, graphs: Sequence, device, cache: dict = None, key=None) -> tuple:
    # fused graph-resident path (gcn_norm folded into the kernel) whenever the model/graphs qualify
    hit = cache.get(key) if cache is not None else None
    if hit is not None:
        S, mc, o, total = model.forward_graphs(hit[0], with_total=True)
        return (S, mc, o, total), hit[1]
    data = graphs[0] if len(graphs) == 1 else Batch.from_data_list(list(graphs))
    if getattr(data, "edge_weight", None) is None and model.resident_ok(data):
        ptr = None if len(graphs) == 1 else data.ptr
        if cache is not None:
            # the loop visits the same graphs in the same order every epoch (train_clustering.py:36,57): the
            # collated batch goes to the device once and stays there (features as float32, train/train.py:79)
            data = data.to(device)
            data.x = data.x.float()
            cache[key] = (data, ptr)
        S, mc, o, total = model.forward_graphs(data, with_total=True)
        return (S, mc, o, total), ptr
    if len(graphs) == 1:
        g = graphs[0]
        ei, ew = gcn_norm(g.edge_index.to(device), getattr(g, "edge_weight", None), g.num_nodes,
                          add_self_loops=True)
        return model(g.x.to(device).float(), ei, ew), None
    big = Batch.from_data_list(list(graphs))
    ei, ew = gcn_norm(big.edge_index.to(device), None, big.num_nodes, add_self_loops=True)
    ptr = big.ptr.to(device).to(torch.int32)
    return model(big.x.to(device).float(), ei, ew, node_ptr=ptr), big.ptr


def _train_clustering_direct(logger, dataset, model: SCN, model_cfg, optim_cfg, batch_graphs: int, device,
                             flat_optimizer: bool = True, epoch_kernel: bool = True):
    """The same loop with every step issued as direct C-ABI launches (``step.ScnTrainStep`` on one shared workspace:
    one launch for forward + losses + backward where the shape fits, else the launch pair + ordered reduction)
    followed by the optimizer step as one more launch (``optim.FlatAdam``; a replay of torch's captured step for
    optimizers it does not cover) -- two host calls per step instead of an autograd graph and an eager
    ``optimizer.step()`` (~170 us of Python per graph).  Same schedule, same arithmetic: the reference's trajectory (tests/test_gpu_pipeline.py::
    test_stage_a_driver_follows_the_reference_trajectory).  Returns None when the model / optimizer / graphs do not
    qualify (the caller then takes the autograd loop)."""
    from ..optim import FlatAdam
    from ..replay import capture_optimizer_step
    from ..step import ScnEpochRunner, ScnStructurePool, ScnTrainStep, ScnWorkspace
    n = len(dataset)
    if batch_graphs == 1 and epoch_kernel and flat_optimizer and n > 0 and \
            not any(getattr(dataset[j], "edge_weight", None) is not None for j in range(n)):
        # the reference's own form -- one optimizer step per graph -- with the whole chain of visits issued by ONE call
        big = Batch.from_data_list([dataset[j] for j in range(n)])
        big.x = big.x.float()                     # (as the per-visit loop below: float features)
        if ScnEpochRunner.eligible(model, big, optim_cfg.optim_type):
            runner = ScnEpochRunner(model, big.to(device), optim_cfg.optim_type, optim_cfg.lr, optim_cfg.weight_decay)
            if logger is not None:
                logger.info(f"Fitting clustering, {model_cfg.cluster_epochs} epochs in one call...")
            runner.run(model_cfg.cluster_epochs * n)
            if logger is not None:
                logger.info("Generating cluster assignments...")
            ids = _assign(runner.assign())
            torch.cuda.synchronize(device)
            runner.check()
            ids = ids.cpu().numpy()
            p_ = big.ptr.cpu().numpy()
            model.last_engine = "resident"
            return [ids[p_[k]:p_[k + 1]] for k in range(n)]
    groups = [[dataset[j] for j in range(i, min(i + batch_graphs, n))] for i in range(0, n, batch_graphs)]
    datas = [g[0] if len(g) == 1 else Batch.from_data_list(g) for g in groups]
    if any(getattr(d, "edge_weight", None) is not None for d in datas) or not all(model.resident_ok(d) for d in datas):
        return None
    try:
        optimizer = OPTIM_DICT[optim_cfg.optim_type](model.parameters(), lr=optim_cfg.lr,
                                                      weight_decay=optim_cfg.weight_decay, capturable=True, fused=True)
    except (TypeError, RuntimeError):
        return None                          # (Adagrad has no capturable step)
    conv, lin = model.mp.module_0, list(model.mlp)[0]
    H, F = conv.lin_rel.weight.shape
    K = lin.weight.shape[0]
    ws = ScnWorkspace(device, max(int(d.num_nodes) for d in datas), max(int(d.edge_index.size(1)) for d in datas),
                      max(len(g) for g in groups), F, H, K)
    # CSRs, out-degrees and A_hat x of every step stay in HBM after the first epoch (built from the graph and the
    # input features alone): later visits load them
    pool = ScnStructurePool(device, sum(int(d.num_nodes) for d in datas), sum(int(d.edge_index.size(1)) for d in datas),
                            sum(len(g) for g in groups)) if model_cfg.cluster_epochs > 1 else None
    steps = []
    for d in datas:                          # the graphs go to the device once (the loop revisits them every epoch)
        d = d.to(device)
        d.x = d.x.float()
        steps.append(ScnTrainStep(model, d, workspace=ws, structure_pool=pool))
    steps[0].bind_grads()                    # one set of gradient buffers for every step
    # Adam / AdamW: torch's update from the flat gradient buffer as ONE launch (optim.FlatAdam); other optimizers:
    # one replay of torch's captured step
    flat = FlatAdam.from_config(optim_cfg.optim_type, steps[0].param_grads, ws.grads, optim_cfg.lr,
                                optim_cfg.weight_decay) if flat_optimizer else None
    opt_step = flat.step if flat is not None else capture_optimizer_step(model.parameters(), optimizer).replay
    fused = [flat is not None and st.fuses_optimizer(flat) for st in steps]   # one graph per step: the update rides
    for epoch in range(model_cfg.cluster_epochs):                            # in the tail of the step's own launch
        if logger is not None:
            logger.info(f"Fitting clustering, epoch {epoch}...")
        for st, fu in zip(steps, fused):     # train_clustering.py:36-50: one optimizer step per graph (or per batch)
            if fu:
                st.run(opt=flat)
            else:
                st.run()
                opt_step()
    if flat is not None:
        flat.check()
    if logger is not None:
        logger.info("Generating cluster assignments...")
    ids_dev = []
    with torch.no_grad():                    # train_clustering.py:57-69 (one read-back for the whole pass)
        for st in steps:
            st.run_forward()
            ids_dev.append(_assign(st.S))
    torch.cuda.synchronize(device)
    steps[0].check()
    out: List[np.ndarray] = []
    for d, ids in zip(datas, ids_dev):
        ids = ids.cpu().numpy()
        if "ptr" in d and d.ptr is not None:
            p = d.ptr.cpu().numpy()
            out.extend(ids[p[k]:p[k + 1]] for k in range(len(p) - 1))
        else:
            out.append(ids)
    model.last_engine = "resident"
    return out


def train_clustering(logger, dataset, model: SCN, model_cfg, optim_cfg, training_cfg,
                     batch_graphs: int = 1, direct: bool = True, epoch_kernel: bool = True) -> List[np.ndarray]:
    device = next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError("train_clustering runs on the MI355X HIP path: move the SCN to 'cuda'")
    if direct:
        done = _train_clustering_direct(logger, dataset, model, model_cfg, optim_cfg, batch_graphs, device,
                                        epoch_kernel=epoch_kernel)
        if done is not None:
            return done
    optimizer = OPTIM_DICT[optim_cfg.optim_type](lr=optim_cfg.lr, weight_decay=optim_cfg.weight_decay,
                                                 params=model.parameters())
    n = len(dataset)
    resident: dict = {}      # first graph of a step -> (collated batch on the device, host ptr)
    for epoch in range(model_cfg.cluster_epochs):
        if logger is not None:
            logger.info(f"Fitting clustering, epoch {epoch}...")
        for i in range(0, n, batch_graphs):
            graphs = [dataset[j] for j in range(i, min(i + batch_graphs, n))]
            optimizer.zero_grad()
            (_, mc_loss, o_loss, total), _ = _forward(model, graphs, device, resident, i)
            # train_clustering.py:48  loss = mc_loss + o_loss (the fused launch already holds the sum;
            # on the layered path the 4th slot is the dense adjacency placeholder, not a loss)
            loss = total if model.last_engine == "resident" and total is not None else mc_loss + o_loss
            loss.backward(root_grad(loss.device))      # no ones_like fill launch
            optimizer.step()
    cluster_all_lst: List[np.ndarray] = []
    if logger is not None:
        logger.info("Generating cluster assignments...")
    with torch.no_grad():
        for i in range(0, n, batch_graphs):
            graphs = [dataset[j] for j in range(i, min(i + batch_graphs, n))]
            (S, _, _, _), ptr = _forward(model, graphs, device, resident, i)
            ids = _assign(S).cpu().numpy()
            if ptr is None:
                cluster_all_lst.append(ids)
            else:
                p = ptr.numpy()
                cluster_all_lst.extend(ids[p[k]:p[k + 1]] for k in range(len(graphs)))
    return cluster_all_lst
