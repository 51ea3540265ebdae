"""The MPNN baseline (BASELINE config 1: GCN stack, reference model/mpnn.py:13-62) through the HIP path
vs the CPU oracle with identical weights; the library's dropout against its stated contract."""
import math

import numpy as np
import pytest
import torch

from oracle import models as OM
from oracle import pyg_ops as P
from tests.helpers import ATOL, DEV, close, rand_graph

pytestmark = pytest.mark.gpu


def _peptides_batch(B, seed):
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    return Batch.from_data_list(make_dataset("peptides_func", B, seed=seed))


def _dev(b):
    d = b.to(DEV)
    d.x = d.x.float()            # train/train.py:79
    return d


def _cora_shaped(seed):
    """One graph of Cora's shape: n=2708, 10556 directed edges, F=1433 binary, C=7."""
    from graph_hscn.data import Batch, Data
    g = torch.Generator().manual_seed(seed)
    half = rand_graph(2708, 5400, seed)[:, :5278]
    ei = torch.cat([half, half.flip(0)], 1)
    x = (torch.rand(2708, 1433, generator=g) < 0.0127).float()
    return Batch.from_data_list([Data(x=x, edge_index=ei, y=torch.zeros(1, 7))])


def _pair(F, H, C, L, act, dropout, seed):
    from graph_hscn.config.config import ACT_DICT, CONV_DICT
    from graph_hscn.model.mpnn import MPNN
    torch.manual_seed(seed)
    om = OM.MPNN(OM.ACT[act], F, H, C, L, dropout)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pm = MPNN(CONV_DICT["gcn"], ACT_DICT[act], F, H, C, L, dropout).to(DEV)
    assert sorted(pm.state_dict()) == sorted(om.state_dict())
    pm.load_state_dict(om.state_dict())
    return om, pm


def _check_grads(om, pm, atol=1e-4, rtol=1e-3):
    prod = dict(pm.named_parameters())
    for n_, po in om.named_parameters():
        pp = prod[n_]
        if po.grad is None:                 # (a module the forward never calls: mpnn.py builds bns under use_layer_norm)
            assert pp.grad is None, n_
            continue
        assert close(pp.grad, po.grad, atol=atol, rtol=rtol), n_


@pytest.mark.parametrize("loops", [False, True])
def test_gcnconv_default_self_loops_matches_oracle(loops):
    """GCNConv(add_self_loops=True): degrees count the appended loop; loops already in the edge list are
    replaced, not doubled; isolated nodes keep their own feature (dinv = 1)."""
    from graph_hscn.nn.conv import GCNConv
    n, F, H = 300, 9, 16
    ei = rand_graph(n - 20, 900, 5, self_loops=loops)        # nodes n-20.. are isolated
    if loops:
        ei = torch.cat([ei, torch.tensor([[3, 3, 7], [3, 3, 7]])], 1)   # a doubled loop as well
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(2))
    torch.manual_seed(0)
    oc = P.GCNConv(F, H)
    with torch.no_grad():
        oc.bias.normal_(0, 0.1)
    pc = GCNConv(F, H).to(DEV)
    pc.load_state_dict(oc.state_dict())
    xo = x.clone().requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    yo = oc(xo, ei)
    yd = pc(xd, ei.to(DEV))
    assert close(yd, yo)
    g = torch.randn(n, H, generator=torch.Generator().manual_seed(3))
    yo.backward(g)
    yd.backward(g.to(DEV))
    assert close(xd.grad, xo.grad, atol=1e-5, rtol=1e-4)
    assert close(pc.lin.weight.grad, oc.lin.weight.grad, atol=1e-4, rtol=1e-4)
    assert close(pc.bias.grad, oc.bias.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("act", ["relu", "tanh", "elu"])
def test_mpnn_peptides_batch_matches_oracle(act):
    """configs/GCN/peptides_func_GCN.yaml: gcn, hidden 16, 3 layers, batch 32; eval mode (dropout off)."""
    b = _peptides_batch(32, seed=4)
    om, pm = _pair(9, 16, 10, 3, act, 0.2, seed=1)
    om.eval(), pm.eval()
    out_o = om(b.x.float(), b.edge_index, b.batch, 32)
    out_d = pm(_dev(b))
    assert out_d.shape == (32, 10)
    assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
    g = torch.randn(32, 10, generator=torch.Generator().manual_seed(1))
    out_o.backward(g)
    out_d.backward(g.to(DEV))
    _check_grads(om, pm)


def test_mpnn_cora_shaped_single_graph_matches_oracle():
    """BASELINE config 1: one graph, batch = 1, F = 1433 (the transform's weight tile fills 92 KB of LDS)."""
    b = _cora_shaped(0)
    om, pm = _pair(1433, 16, 7, 3, "relu", 0.0, seed=2)
    out_o = om(b.x, b.edge_index, b.batch, 1)
    out_d = pm(_dev(b))
    assert out_d.shape == (1, 7)
    assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
    out_o.sum().backward()
    out_d.sum().backward()
    _check_grads(om, pm, atol=1e-5, rtol=1e-3)


def test_mpnn_training_step_with_the_librarys_dropout_mask():
    """Train mode, p = 0.2: the oracle is fed the mask the library drew (recovered by dropping a tensor of
    ones with the same seeds), then prediction and gradients must agree."""
    from graph_hscn.nn import functional as Fh
    b = _peptides_batch(16, seed=9)
    om, pm = _pair(9, 16, 10, 3, "relu", 0.2, seed=3)
    om.train(), pm.train()
    pm.dropout_seed = 1234
    N = b.x.size(0)
    ones = torch.ones(N, 16, device=DEV)
    masks = [(Fh.dropout(ones, 0.2, True, seed=1234 + i) != 0).float().cpu() for i in range(2)]
    out_o = om(b.x.float(), b.edge_index, b.batch, 16, masks=masks)
    out_d = pm(_dev(b))
    assert close(out_d, out_o, atol=ATOL, rtol=1e-5)
    g = torch.randn(16, 10, generator=torch.Generator().manual_seed(1))
    out_o.backward(g)
    out_d.backward(g.to(DEV))
    _check_grads(om, pm)


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 1 << 20])
def test_dropout_contract(n):
    from graph_hscn.nn import functional as Fh
    p = 0.2
    x = torch.randn(n, device=DEV).requires_grad_(True)
    y = Fh.dropout(x, p, True, seed=77)
    kept = y != 0
    assert torch.equal(y[kept], (x.detach() * (1.0 / (1.0 - p)))[kept]) or close(y[kept], x.detach()[kept] / (1 - p), 0, 1e-6)
    assert torch.equal(y, Fh.dropout(x.detach(), p, True, seed=77))                 # a function of (seed, index)
    if n >= 1023:
        assert not torch.equal(kept, Fh.dropout(x.detach(), p, True, seed=78) != 0)
        frac = float(kept.float().mean())
        assert abs(frac - (1 - p)) < 5 * math.sqrt(p * (1 - p) / n)
    g = torch.randn(n, device=DEV)
    y.backward(g)
    assert torch.equal(x.grad != 0, kept & (g != 0))                                # same mask in the backward
    assert close(x.grad[kept], g[kept] / (1 - p), 0, 1e-6)
    # unaligned views take the scalar tail path and draw the same per-element decisions
    if n >= 4:
        buf = torch.zeros(n + 1, device=DEV)
        buf[1:] = x.detach()
        assert torch.equal(Fh.dropout(buf[1:].clone(), p, True, seed=77), y.detach())
    assert Fh.dropout(x, p, False) is x and Fh.dropout(x, 0.0, True) is x           # F.dropout's identities
    with pytest.raises(ValueError):
        Fh.dropout(x, 1.5, True)


def test_train_epoch_runs_the_mpnn_branch():
    """train/train.py:78-80: a model that is not an HSCN gets the batch object itself."""
    from graph_hscn.config.config import MPNNConfig
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.mpnn import build_mpnn
    from graph_hscn.train.train import eval_epoch, train_epoch
    graphs = make_dataset("peptides_func", 24, seed=1)
    loader = [Batch.from_data_list(graphs[i:i + 8]) for i in range(0, 24, 8)]
    torch.manual_seed(0)
    model = build_mpnn(MPNNConfig("gcn", "relu"), 9, 10).to(DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=0.01)
    first = train_epoch(0, None, loader, model, opt, "cross_entropy", None, 1, False)[0]
    for e in range(1, 15):
        last = train_epoch(e, None, loader, model, opt, "cross_entropy", None, 1, False)[0]
    assert math.isfinite(last) and last < first
    assert math.isfinite(eval_epoch(0, None, loader, model, "cross_entropy", None, "Validation")[0])


@pytest.mark.parametrize("use_bn,use_ln,act", [(False, True, "relu"), (True, True, "elu"), (True, True, "tanh")])
def test_mpnn_with_normalisation_layers_matches_oracle(use_bn, use_ln, act):
    """model/mpnn.py:34-44,53-56: LayerNorm alone, and BatchNorm1d + LayerNorm (both lists exist under
    use_layer_norm), forward / gradients in training mode -- batch statistics and the running-statistics update --
    then eval mode on the updated running statistics."""
    from graph_hscn.config.config import ACT_DICT, CONV_DICT
    from graph_hscn.model.mpnn import MPNN
    b = _peptides_batch(12, seed=5)
    torch.manual_seed(3)
    om = OM.MPNN(OM.ACT[act], 9, 16, 10, 3, 0.0, use_batch_norm=use_bn, use_layer_norm=use_ln)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias") or "bns" in n_ or "lns" in n_:
                p.add_(torch.randn_like(p) * 0.2)
    pm = MPNN(CONV_DICT["gcn"], ACT_DICT[act], 9, 16, 10, 3, 0.0, use_batch_norm=use_bn, use_layer_norm=use_ln).to(DEV)
    assert sorted(pm.state_dict()) == sorted(om.state_dict())
    pm.load_state_dict(om.state_dict())
    d = _dev(b)
    g = torch.randn(12, 10, generator=torch.Generator().manual_seed(1))
    om.train(); pm.train()
    for step in range(2):                                   # twice: the running statistics move twice
        om.zero_grad(); pm.zero_grad()
        out_o = om(b.x.float(), b.edge_index, b.batch, 12)
        out_d = pm(d)
        assert close(out_d, out_o, atol=2e-5, rtol=1e-4)
        out_o.backward(g)
        out_d.backward(g.to(DEV))
        _check_grads(om, pm, atol=2e-4, rtol=2e-3)
    pbuf = dict(pm.named_buffers())
    for n_, bo in om.named_buffers():
        assert close(pbuf[n_].float(), bo.float(), atol=1e-5, rtol=1e-5), n_
    om.eval(); pm.eval()
    with torch.no_grad():
        assert close(pm(d), om(b.x.float(), b.edge_index, b.batch, 12), atol=2e-5, rtol=1e-4)


def test_mpnn_batch_norm_alone_fails_as_in_the_reference():
    """mpnn.py:35 creates ``bns`` under use_layer_norm, :53-54 reads it under use_batch_norm: AttributeError."""
    from graph_hscn.config.config import ACT_DICT, CONV_DICT
    from graph_hscn.model.mpnn import MPNN
    pm = MPNN(CONV_DICT["gcn"], ACT_DICT["relu"], 9, 16, 10, 3, 0.0, use_batch_norm=True).to(DEV)
    with pytest.raises(AttributeError):
        pm(_dev(_peptides_batch(3, seed=1)))


@pytest.mark.parametrize("N,H", [(1, 16), (300, 16), (1000, 33), (70, 128)])
def test_norm_layers_match_torch(N, H):
    from graph_hscn.nn import BatchNorm1d, LayerNorm
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn(N, H, generator=g) * 3 + 1
    gy = torch.randn(N, H, generator=g)
    for cls_p, cls_o in ((LayerNorm, torch.nn.LayerNorm), (BatchNorm1d, torch.nn.BatchNorm1d)):
        if cls_p is BatchNorm1d and N < 2:
            continue
        mo, mp = cls_o(H), cls_p(H).to(DEV)
        with torch.no_grad():
            mo.weight.add_(torch.randn(H, generator=g) * 0.3); mo.bias.add_(torch.randn(H, generator=g) * 0.3)
        mp.load_state_dict(mo.state_dict())
        xo = x.clone().requires_grad_(True)
        xp = x.clone().to(DEV).requires_grad_(True)
        yo, yp = mo(xo), mp(xp)
        yo.backward(gy); yp.backward(gy.to(DEV))
        assert close(yp, yo, atol=2e-5, rtol=1e-5)
        assert close(xp.grad, xo.grad, atol=2e-5, rtol=1e-4)
        assert close(mp.weight.grad, mo.weight.grad, atol=1e-4, rtol=1e-4) and close(mp.bias.grad, mo.bias.grad, atol=1e-4, rtol=1e-4)
        for (n_, bo), (_, bp) in zip(mo.named_buffers(), mp.named_buffers()):
            assert close(bp.float(), bo.float(), atol=1e-5, rtol=1e-5), n_
