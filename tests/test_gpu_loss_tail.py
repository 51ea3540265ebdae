"""The loss tail of a graph-resident HSCN step (reference loss.py:6-19 after model/hscn.py:102-114) evaluated
inside the backward launch (include/hscn.h: hscn_loss_tail) against the three-call route
forward -> hscn_criterion_fwd -> backward, and against the CPU oracle."""
import pytest
import torch

from oracle import models as OM
from tests.helpers import DEV, close, hetero_batch

pytestmark = pytest.mark.gpu


def _setup(B=12, K=8, C=10, seed=3, name="peptides_func", H=16):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    import numpy as np
    graphs = make_dataset(name, B, seed=seed)
    rng = np.random.default_rng(seed)
    hb = HeteroBatch.from_data_list([hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs])
    from graph_hscn.model.hscn import HSCN
    torch.manual_seed(seed)
    m = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], hb["local"].x.size(1), H, C, 3).to(DEV)
    m.engine = "resident"
    d = hb.to(DEV)
    y = (torch.rand(B, C, generator=torch.Generator().manual_seed(seed)) < 0.3).float().to(DEV)
    return m, d, y, hb


def _grads(m):
    return [None if p.grad is None else p.grad.clone() for p in m.parameters()]


def _step(m, d, y, loss_fn, tail, read_first=False, scale=None):
    from graph_hscn.loss import LazyLoss, criterion
    for p in m.parameters():
        p.grad = None
    pred = m(d.x_dict, d.edge_index_dict, d)
    assert m.last_engine == "resident" and hasattr(pred, "_hscn_score")
    if not tail:
        del pred._hscn_score
    loss, score = criterion(loss_fn, pred, y)
    assert isinstance(loss, LazyLoss) == tail
    early = float(loss) if read_first else None
    held = loss.detach()                       # what a loop keeps for its epoch mean (train/train.py)
    if tail and not read_first:
        assert isinstance(held, LazyLoss) and loss.state.value is None      # still nothing launched for the loss
    (loss if scale is None else loss * scale).backward()
    return float(held), early, score.clone(), pred.detach().clone(), _grads(m)


@pytest.mark.parametrize("loss_fn,H,C", [("cross_entropy", 16, 10), ("l1", 16, 10), ("cross_entropy", 32, 100),
                                         ("l1", 64, 3)])
def test_tail_route_equals_three_call_route(loss_fn, H, C):
    """(H = 32, C = 100: the head block is larger than two passes of the workgroup, the loss row is parked by
    the strided tail of the loader)"""
    m, d, y, _ = _setup(H=H, C=C, B=5 if H == 64 else 12, name="pcqm_contact" if H == 64 else "peptides_func")
    l0, _, s0, p0, g0 = _step(m, d, y, loss_fn, tail=False)
    l1, _, s1, p1, g1 = _step(m, d, y, loss_fn, tail=True)
    assert torch.equal(p0, p1) and torch.equal(s0, s1)            # the forward launch wrote the same score
    for a, b in zip(g0, g1):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)                               # same per-element arithmetic, same reduction
    assert abs(l0 - l1) < 1e-6                                     # (sum order of the loss terms differs)


def test_tail_route_matches_oracle():
    m, d, y, hb = _setup(B=9, seed=5)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], 9, 16, 10, 3)
    om.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    out = om(hb.x_dict, hb.edge_index_dict, hb["local"].batch, 9)
    lo, so = OM.criterion("cross_entropy", out, y.cpu())
    lo.backward()
    l1, _, s1, p1, g1 = _step(m, d, y, "cross_entropy", tail=True)
    assert abs(l1 - float(lo.detach())) < 1e-6 and close(s1, so) and close(p1, out)
    for (n_, po), g in zip(om.named_parameters(), g1):
        if po.grad is None:
            assert g is None, n_
        else:
            assert close(g, po.grad, atol=1e-5, rtol=1e-3), n_


def test_loss_read_before_backward_and_scaled_loss():
    """``loss.item()`` ahead of ``backward()`` (the reference loop, train/train.py:85) evaluates the loss with
    its own launch and the backward uses that gradient; a scaled loss scales the gradients."""
    m, d, y, _ = _setup(B=7, seed=8)
    l0, _, _, _, g0 = _step(m, d, y, "cross_entropy", tail=False)
    l1, early, _, _, g1 = _step(m, d, y, "cross_entropy", tail=True, read_first=True)
    assert early == l0 == l1                                        # the same kernel produced all three
    for a, b in zip(g0, g1):
        assert (a is None and b is None) or torch.equal(a, b)
    _, _, _, _, g2 = _step(m, d, y, "cross_entropy", tail=True, scale=0.5)
    for a, b in zip(g0, g2):
        assert (a is None and b is None) or close(b, 0.5 * a, atol=1e-7, rtol=1e-6)


def test_gradient_accumulation_over_two_batches_and_eval():
    from graph_hscn.loss import LazyLoss, criterion
    m, d, y, _ = _setup(B=6, seed=1)
    m2, d2, y2, _ = _setup(B=10, seed=2)
    losses = []
    for p in m.parameters():
        p.grad = None
    for dd, yy in ((d, y), (d2, y2)):
        loss, _ = criterion("cross_entropy", m(dd.x_dict, dd.edge_index_dict, dd), yy)
        losses.append(loss.detach())
        loss.backward()
    acc = _grads(m)
    _, _, _, _, ga = _step(m, d, y, "cross_entropy", tail=False)
    la = float(criterion("cross_entropy", m(d.x_dict, d.edge_index_dict, d), y)[0])
    _, _, _, _, gb = _step(m, d2, y2, "cross_entropy", tail=False)
    for a, b, c in zip(acc, ga, gb):
        assert (a is None and b is None) or close(a, b + c, atol=1e-7, rtol=1e-5)
    assert abs(float(losses[0]) - la) < 1e-6 and torch.stack(losses).shape == (2,)
    with torch.no_grad():                                            # no backward to ride on: the plain route
        loss, score = criterion("cross_entropy", m(d.x_dict, d.edge_index_dict, d), y)
    assert not isinstance(loss, LazyLoss) and abs(float(loss) - la) < 1e-6


def test_a_prediction_changed_in_place_takes_the_plain_route():
    """The score that travels with the prediction belongs to the values the forward wrote: after an in-place
    edit the criterion evaluates loss and score itself."""
    from graph_hscn.loss import LazyLoss, criterion
    m, d, y, _ = _setup(B=4, seed=4)
    pred = m(d.x_dict, d.edge_index_dict, d)
    with torch.no_grad():
        pred.mul_(0.5)
    loss, score = criterion("cross_entropy", pred, y)
    assert not isinstance(loss, LazyLoss)
    assert close(score, torch.sigmoid(pred.detach()), atol=1e-6)
