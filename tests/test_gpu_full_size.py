"""The bench's own training step against the CPU oracle AT THE SIZES AND HYPER-PARAMETERS BASELINE.json states
(one test per workload; the oracle's full step takes milliseconds -- bench.py's cpu_baseline times it -- so nothing
has to be scaled down):

  configs[1] Peptides-func   B = 128, K = 16, H = 16, L = 3, C = 10, BCE-with-logits
  configs[2] Peptides-struct B = 32 per GPU (256 over 8), K = 32, L = 3, C = 11, L1
  configs[3] PascalVOC-SP    B = 128, K = 64, C = 21 (stage C through whatever route the product takes)
  configs[4] PCQM-Contact    B = 256, K = 16, half feature storage, L1

The step under test is ``graph_hscn.step.ResidentTrainStep`` in its DEFAULT issue form -- what bench.py captures and
times -- on cluster ids drawn uniformly (~K virtual nodes per graph: what a trained SCN produces).

Tolerances:
  * prediction, score, loss: north_star's 1e-5 against the float32 oracle;
  * parameter gradients: float64 as the referee between two float32 evaluations (the stage-A gradients got this
    in round 2, tests/test_gpu_step.py): the oracle is evaluated in float32 and in float64 on the same inputs, and for
    every parameter tensor, in max norm,
        |HIP - f64|  <=  2 |oracle_f32 - f64| + 8 ulp(scale),     ulp(scale) = 2^-23 max|grad_f64|.
    Both float32 evaluations are sums of the same ~N x H terms in different orders, each within the same a-priori
    bound of the exact value; the factor 2 covers that their actual rounding errors are independent draws (one may
    be lucky), the 8 ulp a tensor whose float32 oracle happens to be exact.  A kernel that drops or doubles a term
    is off by a whole term -- orders of magnitude beyond either;
  * half storage (configs[4]): the rounding points are part of the function, so the oracle emulates them
    (oracle.models.half_storage, derivation in tests/test_gpu_f16.py) and the bounds are that file's: 2^-10 of the
    scale for predictions, 4 * 2^-10 of a gradient's magnitude.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import hetero_data as OH
from oracle import models as OM
from tests.helpers import DEV

pytestmark = pytest.mark.gpu

HALF_ULP = 2.0 ** -10


def _build(name, B, K, C, loss_fn, seed):
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    graphs = make_dataset(name, B, seed=seed)
    rng = np.random.default_rng(seed)
    ids = [rng.integers(0, K, g.num_nodes) for g in graphs]
    gen = torch.Generator().manual_seed(seed)
    y = torch.randn(B, C, generator=gen)
    if loss_fn == "cross_entropy":
        y = (y > 0.8).float()                          # ~Bernoulli(0.2), SURVEY.md 8(d)
    # the product's vectorised host transform (equal to the oracle's per-node loop: tests/test_host_logic.py)
    hs = [hetero_from_clusters(g, i, K) for g, i in zip(graphs, ids)]
    for h, row in zip(hs, y):
        h["local"].y = row.view(1, C).clone()
    pb = HeteroBatch.from_data_list(hs)
    ob = OH.collate_hetero([OH.hetero_from_clusters(g.x, g.edge_index, row.view(1, C), i, K)
                            for g, i, row in zip(graphs, ids, y)])
    return ob, pb, y


def _models(F, H, C, L, seed):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN
    torch.manual_seed(seed)
    om = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], F, H, C, L)
    with torch.no_grad():
        for n_, p in om.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    pm = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], F, H, C, L).to(DEV)
    pm.load_state_dict(om.state_dict())
    pm.engine = "resident"
    return om, pm


def _oracle_step(m, ob, y, B, loss_fn, dtype, store=None):
    m.zero_grad(set_to_none=True)
    x = {k: v.to(dtype) for k, v in ob["x_dict"].items()}
    kw = {"store": store} if store is not None else {}
    out = m(x, ob["edge_index_dict"], ob["batch_local"], B, **kw)
    loss, score = OM.criterion(loss_fn, out, y.to(dtype))
    loss.backward()
    grads = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
    return out.detach().double(), loss.detach().double(), score.detach().double(), grads


@pytest.mark.parametrize("name,B,K,H,L,C,loss_fn,dtype", [
    ("peptides_func", 128, 16, 16, 3, 10, "cross_entropy", torch.float32),
    ("peptides_struct", 32, 32, 16, 3, 11, "l1", torch.float32),
    ("pascalvoc_sp", 128, 64, 16, 3, 21, "cross_entropy", torch.float32),
    ("pcqm_contact", 256, 16, 16, 3, 1, "l1", torch.float16),
    # batches of several rounds of the chip (2 B workgroups > 256 CUs): the one-launch step's default issue form there,
    # and config 3's GLOBAL batch on one GPU (the strong-scaling leg of bench.py --gpus 8)
    ("peptides_func", 512, 16, 16, 3, 10, "cross_entropy", torch.float32),
    ("peptides_struct", 256, 32, 16, 3, 11, "l1", torch.float32),
])
def test_bench_step_against_the_oracle_at_the_stated_size(name, B, K, H, L, C, loss_fn, dtype):
    from graph_hscn.step import ResidentTrainStep
    ob, pb, y = _build(name, B, K, C, loss_fn, seed=B + K)
    F = ob["x_dict"]["local"].size(1)
    om, pm = _models(F, H, C, L, seed=B)
    half = dtype == torch.float16
    d = pb.to(DEV)
    if half:
        d = d.with_feature_dtype(torch.float16)
    rs = ResidentTrainStep(pm, d, loss_fn)               # default issue form: what bench.py captures
    rs.bind_grads()
    rs.run()
    torch.cuda.synchronize()
    rs.check()
    store = OM.half_storage if half else None
    p32, l32, s32, g32 = _oracle_step(om, ob, y, B, loss_fn, torch.float32, store)
    print(f"[full size] {name} B={B} K={K} L={L} C={C} {loss_fn} {'f16 storage' if half else 'f32'}: "
          f"issue form: {'one launch' if rs.one_launch else 'launch pair'}; N={int(d['local'].num_nodes)} "
          f"V={int(d['virtual'].num_nodes)}")
    pred, score, loss = rs.pred.cpu().double(), rs.score.cpu().double(), rs.loss.cpu().double()
    ps = max(1.0, float(p32.abs().max()))
    bar = HALF_ULP * ps if half else 1e-5 * ps
    d_pred, d_score, d_loss = float((pred - p32).abs().max()), float((score - s32).abs().max()), float((loss - l32).abs())
    print(f"   |pred - oracle| = {d_pred:.3e} (bar {bar:.1e})  |score - oracle| = {d_score:.3e}  |loss - oracle| = {d_loss:.3e}")
    assert d_pred <= bar and d_score <= bar
    assert d_loss <= (HALF_ULP if half else 1e-5) * max(1.0, abs(float(l32)))
    got = {n: p.grad.detach().cpu().double() for n, p in pm.named_parameters() if p.grad is not None}
    assert got.keys() == g32.keys()                      # the virtual branch's parameters stay grad-less on both sides
    if half:
        for n in g32:
            gs = max(1e-3, float(g32[n].abs().max()))
            dd = float((got[n] - g32[n]).abs().max())
            assert dd <= 4 * HALF_ULP * gs, (n, dd, gs)
        return
    o64 = copy.deepcopy(om).double()
    _, _, _, g64 = _oracle_step(o64, ob, y, B, loss_fn, torch.float64)
    worst = 0.0
    for n in g64:
        e_hip = float((got[n] - g64[n]).abs().max())
        e_o32 = float((g32[n] - g64[n]).abs().max())
        ulp = 2.0 ** -23 * float(g64[n].abs().max())
        lim = 2.0 * e_o32 + 8.0 * ulp
        worst = max(worst, e_hip / max(lim, 1e-300))
        print(f"   grad {n:48s} |HIP-f64| = {e_hip:.3e}  |oracle32-f64| = {e_o32:.3e}  ulp(scale) = {ulp:.3e}")
        assert e_hip <= lim, (n, e_hip, e_o32, ulp)
    print(f"   worst |HIP - f64| / (2 |oracle32 - f64| + 8 ulp) = {worst:.3f}")


def test_virtual_features_at_the_headline_size():
    """The virtual branch cannot reach the prediction, so the step test above does not see it:
    its final features, as the one-launch step's virtual workgroups leave them, against the oracle at B = 128."""
    from graph_hscn.step import ResidentTrainStep
    from tests.helpers import scale_close
    ob, pb, y = _build("peptides_func", 128, 16, 10, "cross_entropy", seed=7)
    om, pm = _models(9, 16, 10, 3, seed=2)
    d = pb.to(DEV)
    rs = ResidentTrainStep(pm, d, "cross_entropy")
    rs.run()
    torch.cuda.synchronize()
    rs.check()
    assert rs.virtual is not None
    with torch.no_grad():
        xo = ob["x_dict"]
        for conv in om.convs:
            xo = {k: v.relu() for k, v in conv(xo, ob["edge_index_dict"]).items()}
    assert scale_close(rs.virtual, xo["virtual"])
