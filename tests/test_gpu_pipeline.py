"""End-to-end stage order of the reference's main.py:100-119 on the HIP path:
SCN -> train_clustering -> generate_hetero_data -> hetero_loaders -> HSCN -> train."""
import logging

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stage_a_b_c_pipeline_runs_and_learns():
    from graph_hscn.config.config import DataConfig, HSCNConfig, OptimConfig, TrainingConfig
    from graph_hscn.loader.hetero_data import generate_hetero_data, hetero_loaders
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN, build_hscn
    from graph_hscn.train.train import train
    from graph_hscn.train.train_clustering import train_clustering
    torch.manual_seed(0)
    log = logging.getLogger("t")
    graphs = make_dataset("peptides_func", 24, seed=0)
    mc = HSCNConfig("relu", num_clusters=8, cluster_epochs=1)
    oc = OptimConfig("adam", lr=0.01)
    tc = TrainingConfig("hscn", "cross_entropy", "ap", epochs=3, eval_period=1, patience=10)
    dc = DataConfig("peptides_func", batch_size=8)
    scn = SCN(mc.mp_units, "elu", 9, mc.num_clusters).to("cuda")
    clusters = train_clustering(log, graphs, scn, mc, oc, tc)                       # reference trajectory: 1 graph / step
    assert len(clusters) == 24
    assert all(c.shape[0] == g.num_nodes and c.dtype == np.int64 for c, g in zip(clusters, graphs))
    assert all(0 <= c.min() and c.max() < 8 for c in clusters)
    clusters_b = train_clustering(log, graphs, SCN(mc.mp_units, "elu", 9, 8).to("cuda"), mc, oc, tc, batch_graphs=8)
    assert len(clusters_b) == 24 and all(c.shape[0] == g.num_nodes for c, g in zip(clusters_b, graphs))
    split = {"train": torch.arange(0, 16), "val": torch.arange(16, 20), "test": torch.arange(20, 24)}
    hs = generate_hetero_data(clusters, graphs, split, dc, mc, log)
    loaders = hetero_loaders(dc, hs, split)
    model = build_hscn(mc, 9, 10).to("cuda")
    hist = train(log, oc, tc, loaders, model)
    assert len(hist) == 3 and hist[-1][0] < hist[0][0]                              # loss goes down
    assert model.last_engine == "resident"


def test_loss_gradient_scalar_is_applied_by_whoever_consumes_it():
    """criterion's backward returns (unscaled gradient, scalar) as a LazyScaled tensor: the resident
    backward applies the scalar in its launch, every other consumer sees the product.  All routes give
    the gradients of the plain formulation, bit for bit where the arithmetic is the same."""
    import numpy as np
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.loss import LazyCriterionGrad, LazyScaled, criterion
    from graph_hscn.model.hscn import HSCN
    dev = torch.device("cuda:0")
    graphs = make_dataset("peptides_func", 6, seed=1)
    rng = np.random.default_rng(0)
    hb = HeteroBatch.from_data_list([hetero_from_clusters(g, rng.integers(0, 16, g.num_nodes), 16) for g in graphs]).to(dev)
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to(dev)
    y = (torch.rand(6, 10, device=dev) > 0.5).float()
    res = {}
    for eng in ("resident", "layered"):
        for root in (None, 0.37):
            model.engine = eng
            model.zero_grad(set_to_none=True)
            pred = model(hb.x_dict, hb.edge_index_dict, hb)
            seen = []
            pred.register_hook(lambda g: seen.append(type(g)))
            loss, _ = criterion("cross_entropy", pred, y)
            if root is None:
                loss.backward()
            else:
                loss.backward(torch.tensor(root, device=dev))
            # (the resident engine's prediction comes with its score: there the whole loss tail is handed over)
            assert seen and seen[0] is (LazyCriterionGrad if eng == "resident" else LazyScaled)
            res[(eng, root)] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # reference: plain torch loss on the layered engine
    for root in (None, 0.37):
        model.engine = "layered"
        model.zero_grad(set_to_none=True)
        pred = model(hb.x_dict, hb.edge_index_dict, hb)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, y)
        loss.backward() if root is None else loss.backward(torch.tensor(root, device=dev))
        ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        for eng in ("resident", "layered"):
            got = res[(eng, root)]
            assert got.keys() == ref.keys()
            for n in ref:
                assert torch.allclose(got[n], ref[n], atol=1e-6, rtol=1e-4), (eng, root, n)


def test_captured_step_replays_on_different_batches():
    """graph_hscn.replay: one captured training step (static-capacity buffers) replayed on three different
    batches == the eager step on each of them, bit for bit (prediction, loss, gradients)."""
    import numpy as np
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.loss import criterion
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.replay import CapturedStep, StaticHeteroBatch
    dev = torch.device("cuda:0")
    B, K, C = 10, 16, 10
    rng = np.random.default_rng(0)
    batches = []
    for seed in (1, 2, 3):
        graphs = make_dataset("peptides_func", B, seed=seed)
        for g in graphs:
            g.y = torch.from_numpy((rng.random((1, C)) < 0.3).astype(np.float32))
        batches.append(HeteroBatch.from_data_list(
            [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]))
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, C, 3).to(dev)
    model.engine = "resident"
    ref = []
    for hb in batches:
        d = hb.to(dev)
        model.zero_grad(set_to_none=True)
        pred = model(d.x_dict, d.edge_index_dict, d)
        loss, _ = criterion("cross_entropy", pred, d["local"].y)
        loss.backward()
        ref.append((pred.detach().clone(), loss.detach().clone(),
                    {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    del pred, loss, d      # (tensors of an eager step keep autograd nodes tied to the stream they ran on)
    static = StaticHeteroBatch(batches, dev)
    static.load(batches[0])
    from tests.helpers import grads_close, pool_order_close
    packed = [static.pack(hb) for hb in batches]          # laid out like the static buffers: one copy per load
    for one_launch in (False, None):      # the launch pair (bit-identical to eager), then the default one-launch step
        step = CapturedStep(model, static, "cross_entropy", one_launch=one_launch)
        assert step.step.one_launch == (one_launch is None)
        for k, i in enumerate((1, 2, 0, 2, 1)):
            static.load(packed[i] if k % 2 else batches[i])
            loss = step.replay()
            torch.cuda.synchronize()
            same = torch.equal if one_launch is False else pool_order_close
            assert same(step.pred, ref[i][0])
            assert same(loss, ref[i][1])
            for n, p in model.named_parameters():
                if n in ref[i][2]:
                    if one_launch is False:
                        assert torch.equal(p.grad, ref[i][2][n]), (i, n)
                    else:
                        assert grads_close(p.grad, ref[i][2][n]), (i, n)
    static.batch._resident_meta.check()


def test_resident_training_loop_learns_and_visits_every_graph():
    """train/train_resident.py: epochs of device-collated shuffled batches through one captured iteration
    (forward + loss + backward + AdamW), the short last batch eagerly; loss falls, metric is computed from the
    scores of all graphs, evaluation and early stopping work on host loaders."""
    import numpy as np
    from graph_hscn.config.config import ACT_DICT, OptimConfig, TrainingConfig
    from graph_hscn.data import DataLoader
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.metrics import eval_ap
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.train.train_resident import fit_resident
    graphs = make_dataset("peptides_func", 70, seed=3)
    rng = np.random.default_rng(0)
    hs = [hetero_from_clusters(g, rng.integers(0, 8, g.num_nodes), 8) for g in graphs]
    torch.manual_seed(0)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to("cuda")
    tc = TrainingConfig("hscn", "cross_entropy", "ap", epochs=12, eval_period=4, patience=50)
    loaders = [DataLoader(hs[50:60], batch_size=5), DataLoader(hs[60:], batch_size=5)]
    hist = fit_resident(None, OptimConfig("adamW", lr=0.01), tc, hs[:50], loaders, model, batch_size=16, metric_fn=eval_ap)
    assert len(hist) == 12                                          # 3 captured steps + a 2-graph eager tail per epoch
    assert all(np.isfinite(l) and 0.0 <= p <= 1.0 for l, p in hist)
    assert hist[-1][0] < hist[0][0]
    # the update as one launch (optim.FlatAdam, the default for Adam / AdamW) against torch's captured fused AdamW:
    # the same trajectory (eager ragged tail included) to float rounding
    torch.manual_seed(0)
    model3 = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to("cuda")
    hist3 = fit_resident(None, OptimConfig("adamW", lr=0.01), tc, hs[:50], loaders, model3, batch_size=16,
                         metric_fn=eval_ap, flat_optimizer=False)
    assert abs(hist3[-1][0] - hist[-1][0]) <= 1e-4 * max(1.0, abs(hist[-1][0]))
    for a, b in zip(model.parameters(), model3.parameters()):
        assert float((a - b).detach().abs().max()) <= 2e-4 * max(1.0, float(b.detach().abs().max()))
    # an optimizer without a capturable step (Adagrad) is stepped outside the graph on the captured gradients
    torch.manual_seed(0)
    model2 = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to("cuda")
    tc2 = TrainingConfig("hscn", "cross_entropy", "ap", epochs=6, eval_period=6, patience=50)
    hist2 = fit_resident(None, OptimConfig("adagrad", lr=0.05), tc2, hs[:50], loaders, model2, batch_size=16)
    assert hist2[-1][0] < hist2[0][0]


def test_resident_training_loop_with_the_rccl_reducer_single_rank():
    """The data-parallel form of fit_resident (replay, RCCL all-reduce of the flat gradient buffer, optimizer step
    as its own graph) with a world of one: same parameters as the single-process loop, bit for bit."""
    import os
    import numpy as np
    import torch.distributed as dist
    from graph_hscn.config.config import ACT_DICT, OptimConfig, TrainingConfig
    from graph_hscn.data import DataLoader
    from graph_hscn.distributed import FlatGradReducer
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.train.train_resident import fit_resident
    graphs = make_dataset("peptides_func", 40, seed=5)
    rng = np.random.default_rng(1)
    hs = [hetero_from_clusters(g, rng.integers(0, 8, g.num_nodes), 8) for g in graphs]
    tc = TrainingConfig("hscn", "cross_entropy", "ap", epochs=3, eval_period=3, patience=50)
    loaders = [DataLoader(hs[30:35], batch_size=5), DataLoader(hs[35:], batch_size=5)]

    def run(reducer_factory):
        torch.manual_seed(0)
        m = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, 16, 10, 3).to("cuda")
        fit_resident(None, OptimConfig("adamW", lr=0.01), tc, hs[:30], loaders, m, batch_size=8,
                     reducer=reducer_factory(m) if reducer_factory else None)
        return [p.detach().clone() for p in m.parameters()]

    want = run(None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        got = run(lambda m: FlatGradReducer(m, single_rank_collective=True))
    finally:
        dist.destroy_process_group()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("optim", ["adam", "adamW"])
def test_stage_a_driver_follows_the_reference_trajectory(optim):
    """train/train_clustering.py (batch_graphs=1: the reference's one-optimizer-step-per-graph schedule,
    /root/reference/graph_hscn/train/train_clustering.py:34-69) against the oracle's restatement of that loop:
    8 graphs x 2 epochs = 16 sequential Adam steps, then the assignment pass.  Parameters within 1e-4, cluster ids
    bit-exact wherever the oracle's top-2 margin exceeds 1e-5 (and the number of flips on nearer ties is printed)."""
    import logging
    from graph_hscn.config.config import OPTIM_DICT, HSCNConfig, OptimConfig, TrainingConfig
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.train.train_clustering import train_clustering
    from oracle import models as OM
    K = 8
    graphs = make_dataset("peptides_func", 8, seed=21)
    mc = HSCNConfig("relu", num_clusters=K, cluster_epochs=2)
    oc = OptimConfig(optim, lr=0.01)
    tc = TrainingConfig("hscn", "cross_entropy", "ap")
    torch.manual_seed(3)
    om = OM.SCN(mc.mp_units, "elu", 9, K)
    pm = SCN(mc.mp_units, "elu", 9, K).to("cuda")
    pm.load_state_dict(om.state_dict())
    opt = OPTIM_DICT[oc.optim_type](lr=oc.lr, weight_decay=oc.weight_decay, params=om.parameters())
    ids_o, soft_o = OM.train_clustering_loop(om, graphs, mc.cluster_epochs, opt)
    ids_d = train_clustering(logging.getLogger("t"), graphs, pm, mc, oc, tc, batch_graphs=1)
    assert pm.last_engine == "resident"
    for (n, po), (_, pp) in zip(om.named_parameters(), pm.named_parameters()):
        d = float((pp.detach().cpu() - po.detach()).abs().max())
        assert d <= 1e-4, (n, d)
    flips = sure_nodes = 0
    for want, got, S in zip(ids_o, ids_d, soft_o):
        top = S.topk(2, dim=1).values
        sure = ((top[:, 0] - top[:, 1]) > 1e-5).numpy()
        assert got.dtype == np.int64 and got.shape == want.shape
        assert np.array_equal(want[sure], got[sure])
        flips += int((want != got).sum())
        sure_nodes += int(sure.sum())
    print(f"stage-A trajectory ({optim}): ids equal on all {sure_nodes} nodes with margin > 1e-5; flips on nearer ties: {flips}")


@pytest.mark.parametrize("kind,name,K,act", [("adam", "peptides_func", 16, "elu"), ("adamW", "peptides_func", 16, "elu"),
                                             ("adam", "pcqm_contact", 8, "tanh"), ("adamW", "peptides_func", 32, "relu"),
                                             ("adam", "peptides_struct", 4, "elu")])
def test_stage_a_epoch_kernel_is_the_per_visit_loop_bit_for_bit(kind, name, K, act, monkeypatch):
    """The reference's stage-A loop from ONE call (hscn_scn_resident_train_epoch: the dataset as one batch in HBM, its
    structure built by one forward launch, the visits walked by one persistent workgroup -- or issued back to back
    by the library, HSCN_PERSISTENT_EPOCH=0 --, the assignment pass one more launch) against the same loop issued
    visit by visit from Python (ScnTrainStep.run(opt=...) on per-graph
    objects): identical parameters and cluster ids after 3 epochs over 24 graphs of very different sizes."""
    import numpy as np
    from graph_hscn.config.config import HSCNConfig, OptimConfig, TrainingConfig
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.train.train_clustering import train_clustering
    graphs = make_dataset(name, 24, seed=13)
    mc = HSCNConfig("relu", num_clusters=K, cluster_epochs=3)
    oc = OptimConfig(kind, lr=0.01, weight_decay=0.01 if kind == "adamW" else 0.0)
    tc = TrainingConfig("hscn", "cross_entropy", "ap")

    def run(epoch_kernel):
        torch.manual_seed(5)
        scn = SCN([16], act, graphs[0].x.size(1), K).to("cuda")
        ids = train_clustering(None, graphs, scn, mc, oc, tc, batch_graphs=1, epoch_kernel=epoch_kernel)
        assert scn.last_engine == "resident"
        return [p.detach().clone() for p in scn.parameters()], ids

    pa, ia = run(True)                       # the chain walked by one persistent workgroup (k_scn_epoch)
    pb, ib = run(False)                      # per-graph Python objects, one launch per visit
    monkeypatch.setenv("HSCN_PERSISTENT_EPOCH", "0")
    pc, ic = run(True)                       # the library's loop of per-visit launches
    for other_p, other_i in ((pb, ib), (pc, ic)):
        for x, y in zip(pa, other_p):
            assert torch.equal(x, y), float((x - y).abs().max())
        assert len(ia) == len(other_i) == 24
        for x, y in zip(ia, other_i):
            assert np.array_equal(x, y)
