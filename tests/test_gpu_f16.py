"""BASELINE.json configs[4]: Graph-HSCN on PCQM-Contact with half-precision feature storage ("fp16 feat +
bf16 accum"): node features and inter-layer activations are IEEE half in HBM, every sum accumulates in float
(csrc/resident_f16.hip, csrc/resident_scn.hip: hscn_*_f16).

The reference has no reduced-precision mode, so the tolerance is derived, not inherited
(DESIGN.md section 2b):
  * vs the float32 oracle with the SAME rounding points emulated (oracle.models.half_storage applied to the inputs
    and to every layer's output, float32 arithmetic in between): the kernels implement exactly that function, so
    they agree to float32 rounding -- except where the two float32 evaluations fall on different sides of a half
    rounding boundary, which moves ONE stored activation by one half ulp (2^-10 relative).  Bound used:
    HALF_ULP * scale = 2^-10 * max|tensor| for activations, and the same relative to the prediction scale for
    predictions (a flip is diluted by the mean pool, so this is generous);
  * vs the plain float32 oracle on half-rounded INPUTS (what storage costs): every one of the L stored activation
    tensors carries a relative rounding error of at most u = 2^-11 that the following layers propagate with gain
    <= G per layer (||A_hat|| <= 1, ||W|| ~ 1..2): |pred_f16 - pred_f32| <= u * (G + G^2 + ... + G^L) * scale.
    With G = 2: 14 u * scale for L = 3.  Asserted with that constant; the measured figure is printed.
"""
import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from oracle import models as OM
from tests.helpers import DEV

pytestmark = pytest.mark.gpu

U = 2.0 ** -11            # unit roundoff of IEEE half
HALF_ULP = 2.0 ** -10     # one ulp, relative


def _batches(name, B, K, seed):
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset(name, B, seed=seed)
    rng = np.random.default_rng(seed)
    ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
    ob = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, g.y, i, K) for g, i in zip(graphs, ids)])
    pb = HeteroBatch.from_data_list([hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)])
    return ob, pb, graphs


def _models(F, H, C, L, act="relu", seed=0):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN
    torch.manual_seed(seed)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT[act], F, H, C, L)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT[act], F, H, C, L).to(DEV)
    pm.load_state_dict(om.state_dict())
    pm.engine = "resident"
    return om, pm


def _maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.mark.parametrize("name,B,K,H,L,C,act,loss_fn", [
    ("pcqm_contact", 9, 16, 16, 3, 1, "relu", "l1"),            # configs[4] shape, B = 9
    ("pcqm_contact", 9, 16, 32, 2, 10, "tanh", "cross_entropy"),
    ("peptides_func", 6, 16, 16, 3, 10, "relu", "cross_entropy")])
def test_half_storage_step_matches_the_emulating_oracle(name, B, K, H, L, C, act, loss_fn):
    from graph_hscn.loss import criterion
    ob, pb, _ = _batches(name, B, K, seed=B + K)
    F = ob["x_dict"]["local"].size(1)
    om, pm = _models(F, H, C, L, act, seed=B)
    y = torch.randn(B, C, generator=torch.Generator().manual_seed(2))
    if loss_fn == "cross_entropy":
        y = (y > 0).float()
    keep = {}
    out_o = om(ob["x_dict"], ob["edge_index_dict"], ob["batch_local"], B, store=OM.half_storage, keep=keep)
    loss_o, _ = OM.criterion(loss_fn, out_o, y)
    loss_o.backward()
    # plain float32 oracle on half-rounded inputs: what the storage mode costs
    with torch.no_grad():
        out_plain = om({k: v.half().float() for k, v in ob["x_dict"].items()}, ob["edge_index_dict"], ob["batch_local"], B)
    pbd = pb.to(DEV).with_feature_dtype(torch.float16)
    pm.keep_virtual = True
    out_d = pm(pbd.x_dict, pbd.edge_index_dict, pbd)
    assert pm.last_engine == "resident" and out_d.dtype == torch.float32
    assert pm.last_virtual.dtype == torch.float16
    loss_d, score_d = criterion(loss_fn, out_d, y.to(DEV))
    loss_d.backward()
    torch.cuda.synchronize()
    pbd._resident_meta.check()
    ps = max(1.0, float(out_o.abs().max()))
    d_pred = _maxdiff(out_d, out_o)
    d_plain = _maxdiff(out_d, out_plain)
    vs = max(1.0, float(keep["virtual"].abs().max()))
    d_virt = _maxdiff(pm.last_virtual.float(), keep["virtual"])
    gain = sum(2.0 ** l for l in range(1, L + 1))
    print(f"[f16] {name} H={H} L={L}: |pred - emulating oracle| = {d_pred:.3e} (scale {ps:.2f}, bound {HALF_ULP * ps:.3e}); "
          f"|pred - f32 oracle on half inputs| = {d_plain:.3e} (bound {U * gain * ps:.3e}); "
          f"|virtual - emulating oracle| = {d_virt:.3e} (scale {vs:.2f})")
    assert d_pred <= HALF_ULP * ps
    assert d_plain <= U * gain * ps
    assert d_virt <= 2 * HALF_ULP * vs
    assert abs(float(loss_d) - float(loss_o.detach())) <= HALF_ULP * max(1.0, abs(float(loss_o.detach())))
    for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        if po.grad is None:
            assert pp.grad is None, n_
        else:
            gs = max(1e-3, float(po.grad.abs().max()))
            assert _maxdiff(pp.grad, po.grad) <= 4 * HALF_ULP * gs, (n_, _maxdiff(pp.grad, po.grad), gs)


def test_half_storage_direct_step_equals_autograd_step_and_replays():
    """step.ResidentTrainStep and a captured replay on half features == the autograd path, bit for bit."""
    from graph_hscn.loss import criterion
    from graph_hscn.replay import CapturedStep, StaticHeteroBatch
    from graph_hscn.step import ResidentTrainStep
    ob, pb, _ = _batches("pcqm_contact", 9, 16, seed=3)
    om, pm = _models(9, 16, 1, 3, seed=1)
    y = torch.randn(9, 1, generator=torch.Generator().manual_seed(0))
    pb["local"].y = y
    d = pb.to(DEV).with_feature_dtype(torch.float16)
    pm.zero_grad(set_to_none=True)
    pred = pm(d.x_dict, d.edge_index_dict, d)
    loss, score = criterion("l1", pred, d["local"].y)
    loss.backward()
    want = {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}
    rs = ResidentTrainStep(pm, d, "l1", one_launch=False)
    rs.bind_grads()
    rs.run()
    torch.cuda.synchronize()
    assert rs.acts.dtype == torch.float16 and rs.virtual.dtype == torch.float16
    assert torch.equal(rs.pred, pred.detach()) and torch.equal(rs.loss, loss.detach()) and torch.equal(rs.score, score)
    for n, p in pm.named_parameters():
        if n in want:
            assert torch.equal(p.grad, want[n]), n
    static = StaticHeteroBatch([pb], DEV, feature_dtype=torch.float16)
    static.load(pb)
    from tests.helpers import grads_close, pool_order_close
    for one_launch in (False, None):
        step = CapturedStep(pm, static, "l1", one_launch=one_launch)
        step.replay()
        torch.cuda.synchronize()
        same = torch.equal if one_launch is False else pool_order_close
        assert same(step.pred, pred.detach()) and same(step.loss, loss.detach())
        for n, p in pm.named_parameters():
            if n in want:
                assert (torch.equal(p.grad, want[n]) if one_launch is False else grads_close(p.grad, want[n])), n


def test_half_storage_full_size_properties():
    """configs[4] at its full per-step size (B = 256 PCQM-Contact-shaped graphs, K = 16): properties that need no
    oracle run.  (i) a graph's prediction is bit-identical alone and inside the batch; (ii) the batch gradient is
    the mean of the per-graph gradients; (iii) half storage stays within the derived bound of the float32 product
    path on the same (half-representable) inputs; (iv) the stored activations are half-representable by
    construction and the integer atom features survive the narrowing exactly."""
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.loss import criterion
    B, K, C, L = 256, 16, 1, 3
    graphs = make_dataset("pcqm_contact", B, seed=17)
    rng = np.random.default_rng(17)
    hs = [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    for h in hs:
        h["local"].y = torch.randn(1, C, generator=torch.Generator().manual_seed(int(h["local"].x.sum()) % 1000))
    pb = HeteroBatch.from_data_list(hs)
    _, pm = _models(9, 16, C, L, seed=4)
    d32 = pb.to(DEV)
    d16 = d32.with_feature_dtype(torch.float16)
    assert torch.equal(d16["local"].x.float(), d32["local"].x)              # (iv) atom features are exact in half
    pm.zero_grad(set_to_none=True)
    p16 = pm(d16.x_dict, d16.edge_index_dict, d16)
    l16, _ = criterion("l1", p16, d16["local"].y)
    l16.backward()
    g16 = {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}
    pm.zero_grad(set_to_none=True)
    p32 = pm(d32.x_dict, d32.edge_index_dict, d32)
    ps = max(1.0, float(p32.abs().max()))
    gain = sum(2.0 ** l for l in range(1, L + 1))
    d = _maxdiff(p16, p32)
    print(f"[f16 full size] B={B}: |pred_f16 - pred_f32| = {d:.3e}, bound {U * gain * ps:.3e} (scale {ps:.2f})")
    assert d <= U * gain * ps                                                # (iii)
    # (i) + (ii) on a sample of graphs spread over the batch
    acc = {n: torch.zeros_like(g) for n, g in g16.items()}
    for j in range(B):
        single = HeteroBatch.from_data_list([hs[j]]).to(DEV).with_feature_dtype(torch.float16)
        pm.zero_grad(set_to_none=True)
        pj = pm(single.x_dict, single.edge_index_dict, single)
        if j % 16 == 0:
            assert torch.equal(pj[0], p16[j]), j                             # (i)
        lj, _ = criterion("l1", pj, single["local"].y)
        lj.backward()
        for n, p in pm.named_parameters():
            if p.grad is not None:
                acc[n] += p.grad
    for n in g16:
        mean = acc[n] / B
        s = max(1e-4, float(g16[n].abs().max()))
        assert _maxdiff(mean, g16[n]) <= 1e-5 * max(1.0, s) + 1e-6, n        # (ii)


def test_half_storage_stage_a_matches_emulating_oracle():
    """Stage A on half features (hscn_scn_resident_*_f16): assignments, both losses, gradients against the float32
    oracle with x and the hidden activation rounded to half; cluster ids equal wherever the oracle's top-2 margin
    exceeds one half ulp of the assignment scale."""
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    K = 16
    graphs = make_dataset("pcqm_contact", 9, seed=5)
    torch.manual_seed(2)
    om = OM.SCN([16], "elu", 9, K)
    pm = SCN([16], "elu", 9, K).to(DEV)
    pm.load_state_dict(om.state_dict())
    S_o, mcs, os_ = [], [], []
    om.zero_grad()
    for g in graphs:
        ei, ew = OM.P.gcn_norm(g.edge_index, None, g.num_nodes, add_self_loops=True)
        s, mc, o, _ = om(g.x.float(), ei, ew, store=OM.half_storage)
        ((mc + o) / len(graphs)).backward()
        S_o.append(s.detach()); mcs.append(float(mc)); os_.append(float(o))
    S_o = torch.cat(S_o)
    big = Batch.from_data_list(graphs).to(DEV)
    big.x = big.x.half()
    S_d, mc_d, o_d, total = pm.forward_graphs(big, with_total=True)
    assert pm.last_engine == "resident"
    total.backward()
    torch.cuda.synchronize()
    big._scn_meta.check()
    assert _maxdiff(S_d, S_o) <= HALF_ULP
    assert abs(float(mc_d) - np.mean(mcs)) <= HALF_ULP and abs(float(o_d) - np.mean(os_)) <= HALF_ULP
    top = S_o.topk(2, 1).values
    sure = (top[:, 0] - top[:, 1]) > 2 * HALF_ULP
    ids_d, ids_o = S_d.max(1)[1].cpu(), S_o.max(1)[1]
    assert torch.equal(ids_d[sure], ids_o[sure])
    print(f"[f16 stage A] |S - emulating oracle| = {_maxdiff(S_d, S_o):.3e}; id flips {int((ids_d != ids_o).sum())} of "
          f"{ids_o.numel()} nodes ({int(sure.sum())} with margin > 2 half ulp, all equal)")
    for (n_, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        gs = max(1e-3, float(po.grad.abs().max()))
        assert _maxdiff(pp.grad, po.grad) <= 4 * HALF_ULP * gs, (n_, _maxdiff(pp.grad, po.grad), gs)
