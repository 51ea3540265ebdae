"""graph_hscn.step: the training steps as direct C-ABI launches == the autograd path, bit for bit; and the
capture of such a step is immune to whatever earlier eager steps left alive (round 1: host SIGSEGV at
capture_end when an eager ``loss`` / ``pred`` of the same model was still referenced)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(B, K, C, seeds, dev=None):
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    rng = np.random.default_rng(0)
    out = []
    for seed in seeds:
        graphs = make_dataset("peptides_func", B, seed=seed)
        for g in graphs:
            g.y = torch.from_numpy((rng.random((1, C)) < 0.3).astype(np.float32))
        out.append(HeteroBatch.from_data_list(
            [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]))
    return out


def _model(C, H=16, L=3, dev="cuda"):
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.model.hscn import HSCN
    torch.manual_seed(0)
    m = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], 9, H, C, L).to(dev)
    m.engine = "resident"
    return m


def _eager(model, d, loss_fn):
    from graph_hscn.loss import criterion
    model.zero_grad(set_to_none=True)
    pred = model(d.x_dict, d.edge_index_dict, d)
    loss, score = criterion(loss_fn, pred, d["local"].y)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    return pred, loss, score, grads


@pytest.mark.parametrize("one_launch", [False, True])
@pytest.mark.parametrize("overlap,compute_virtual,loss_fn", [(True, True, "cross_entropy"), (False, True, "l1"),
                                                             (True, False, "cross_entropy")])
def test_direct_step_is_the_autograd_step_bit_for_bit(overlap, compute_virtual, loss_fn, one_launch):
    """Both issue forms of the direct step -- the forward + backward launch pair and the ONE-launch step
    (csrc/resident_step.h: forward, loss tail and backward of a graph in one workgroup, activations never leave
    LDS, the virtual branch on its own workgroups fed through an in-launch hand-off) -- against the autograd path."""
    if one_launch and compute_virtual and not overlap:
        pytest.skip("the one-launch step carries the virtual branch only as extra workgroups (overlap)")
    import graph_hscn.engine as eng
    from graph_hscn.step import ResidentTrainStep
    dev = torch.device("cuda:0")
    (hb,) = _batches(7, 16, 10, (4,))
    d = hb.to(dev)
    model = _model(10)
    model.overlap_virtual, model.compute_virtual = overlap, compute_virtual
    pred, loss, score, grads = _eager(model, d, loss_fn)
    want_v = eng.last_deferred_virtual.clone() if (overlap and compute_virtual) else None
    rs = ResidentTrainStep(model, d, loss_fn, one_launch=one_launch)
    assert rs.defer == (overlap and compute_virtual) and rs.one_launch == one_launch
    rs.bind_grads()
    rs.run()
    rs.run()                      # idempotent: buffers are rewritten, not accumulated into
    torch.cuda.synchronize()
    rs.check()
    from tests.helpers import grads_close, pool_order_close
    # the launch pair IS the autograd path's launches: bit for bit; the one-launch step pools out of the last layer's
    # accumulators (another grouping of the same row sums): float rounding
    same = pool_order_close if one_launch else torch.equal
    assert same(rs.pred, pred.detach())
    assert same(rs.score, score)
    assert same(rs.loss, loss.detach())
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert got.keys() == grads.keys()
    for n in grads:
        # the launch pair IS the autograd path's launches; the one-launch step groups the weight gradient's partial
        # sums by row tile (same terms, another order)
        assert (grads_close(got[n], grads[n]) if one_launch else torch.equal(got[n], grads[n])), n
    if want_v is not None:
        assert torch.equal(rs.virtual, want_v)
    if compute_virtual and not overlap:
        model.keep_virtual = True
        with torch.no_grad():
            model(d.x_dict, d.edge_index_dict, d)
        assert torch.equal(rs.virtual, model.last_virtual)


def test_capture_is_immune_to_live_eager_tensors_and_earlier_captures():
    """The regression for round 1's capture_end crash: an eager loss / prediction of the same model is ALIVE while
    a CapturedStep is built, then a second CapturedStep is built on the same model; both replay correctly."""
    from graph_hscn.replay import CapturedStep, StaticHeteroBatch
    dev = torch.device("cuda:0")
    batches = _batches(6, 16, 10, (1, 2))
    model = _model(10)
    ref = []
    keep_alive = []
    for hb in batches:
        d = hb.to(dev)
        pred, loss, score, grads = _eager(model, d, "cross_entropy")
        ref.append((pred.detach().clone(), loss.detach().clone(), grads))
        keep_alive.append((pred, loss, loss.detach(), d))       # autograd graphs of eager steps stay referenced
    static = StaticHeteroBatch(batches, dev)
    static.load(batches[0])
    step1 = CapturedStep(model, static, "cross_entropy", one_launch=False)
    step2 = CapturedStep(model, static, "cross_entropy", one_launch=False)   # an earlier capture on the same model is alive too
    for step in (step1, step2, step1):
        for i in (1, 0):
            static.load(batches[i])
            loss = step.replay()
            torch.cuda.synchronize()
            assert torch.equal(step.pred, ref[i][0])
            assert torch.equal(loss, ref[i][1])
            step.bind_grads()
            for n, p in model.named_parameters():
                if n in ref[i][2]:
                    assert torch.equal(p.grad, ref[i][2][n]), (i, n)
    # and an eager step still works afterwards, on the tensors that were kept
    pred, loss, score, grads = _eager(model, batches[1].to(dev), "cross_entropy")
    assert torch.equal(pred.detach(), ref[1][0])


def test_lazy_loss_detach_drops_the_graph():
    """``loss.detach()`` of the lazily valued loss shares its state; the state holds values only, so collecting
    detached losses over an epoch (train/train.py:85) does not keep every step's activations alive."""
    import gc
    import weakref
    dev = torch.device("cuda:0")
    (hb,) = _batches(3, 8, 10, (5,))
    d = hb.to(dev)
    model = _model(10)
    from graph_hscn.loss import LazyLoss, criterion
    pred = model(d.x_dict, d.edge_index_dict, d)
    loss, _ = criterion("cross_entropy", pred, d["local"].y)
    assert isinstance(loss, LazyLoss)
    kept = loss.detach()
    node = weakref.ref(pred.grad_fn)
    loss.backward()
    want = float(loss)
    del pred, loss
    gc.collect()
    assert node() is None, "the detached loss keeps the prediction's autograd graph alive"
    assert float(kept) == want


def test_scn_direct_step_is_the_autograd_step():
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.step import ScnTrainStep
    dev = torch.device("cuda:0")
    graphs = make_dataset("peptides_func", 9, seed=2)
    torch.manual_seed(0)
    scn = SCN([16], "elu", 9, 16).to(dev)
    big = Batch.from_data_list(graphs).to(dev)
    big.x = big.x.float()
    scn.zero_grad(set_to_none=True)
    S, mc, o, total = scn.forward_graphs(big, with_total=True)
    total.backward()
    want = [p.grad.clone() for p in scn.parameters()]
    st = ScnTrainStep(scn, big, one_launch=False)      # the launch pair the autograd Functions issue: bit for bit
    st.bind_grads()
    st.run()
    st.run()
    torch.cuda.synchronize()
    st.check()
    assert torch.equal(st.S, S)
    assert torch.equal(st.losses, torch.stack([mc.detach(), o.detach(), total.detach()]))
    for p, w in zip(scn.parameters(), want):
        assert torch.equal(p.grad, w)
    g = torch.cuda.CUDAGraph()
    keep = (S, mc, o, total)                       # eager outputs alive during the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st.run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        st.run()
    st.grads.zero_()
    g.replay()
    torch.cuda.synchronize()
    for p, w in zip(scn.parameters(), want):
        assert torch.equal(p.grad, w)
    del keep


def _scn_f64_grads(scn, step, graphs, H, act, K):
    """The stage-A gradient of the batch in float64 on the CPU oracle (mean over graphs of mincut + ortho), flat in
    the order of ``ScnTrainStep.grads``."""
    import oracle.models as OM
    import oracle.pyg_ops as P
    F = graphs[0].x.size(1)
    om = OM.SCN([H], act, F, K).double()
    om.load_state_dict({k: v.detach().cpu().double() for k, v in scn.state_dict().items()})
    for g in graphs:
        ei, ew = P.gcn_norm(g.edge_index, None, g.x.size(0), add_self_loops=True, dtype=torch.float64)
        h = om.mp(g.x.double(), ei, ew)             # (oracle.models.SCN.forward with a float64 dense adjacency)
        _, _, mc, o = P.dense_mincut_pool(h, P.to_dense_adj(ei).double(), om._run_mlp(h))
        ((mc + o) / len(graphs)).backward()
    by_name = dict(om.named_parameters())
    names = {id(p): n_ for n_, p in scn.named_parameters()}
    return torch.cat([by_name[names[id(p)]].grad.reshape(-1) for p in step._mp])


@pytest.mark.parametrize("name,B,K,H,act,dtype", [("peptides_func", 1, 16, 16, "elu", torch.float32),
                                                   ("peptides_func", 128, 16, 16, "elu", torch.float32),
                                                   ("peptides_func", 300, 16, 16, "relu", torch.float16),   # > 256 workgroups
                                                   ("peptides_struct", 24, 32, 16, "elu", torch.float32),
                                                   ("peptides_func", 24, 16, 32, "elu", torch.float32),
                                                   ("peptides_func", 9, 4, 16, "tanh", torch.float32),
                                                   ("peptides_func", 5, 28, 16, "elu", torch.float32),
                                                   ("peptides_func", 16, 40, 16, "tanh", torch.float32),     # pair only
                                                   ("peptides_func", 16, 6, 16, "elu", torch.float32),       # pair only
                                                   ("pcqm_contact", 256, 8, 16, "elu", torch.float16),
                                                   ("pascalvoc_sp", 8, 16, 16, "elu", torch.float32)])
def test_scn_one_launch_step_against_the_launch_pair(name, B, K, H, act, dtype):
    """Stage A's forward + losses + backward in one workgroup program (hscn_scn_resident_train_step) against the
    forward / backward pair of launches: the assignments and the three losses bit for bit (the forward half is the
    same code); the parameter gradients -- the one-launch step runs the backward half as one pass over 16-row tiles on
    the matrix cores, the pair element-wise per row -- to float rounding, judged by a float64 referee where the
    oracle can give one (float storage, small batches): |one - f64| <= 2 |pair - f64| + 2e-6 * scale.  Eager and
    replayed 20 times with identical bits (B = 1: the workgroup writes the gradients itself, no fold launch)."""
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.step import ScnTrainStep
    dev = torch.device("cuda:0")
    graphs = make_dataset(name, B, seed=5)
    F = graphs[0].x.size(1)
    torch.manual_seed(1)
    scn = SCN([H], act, F, K).to(dev)
    big = Batch.from_data_list(graphs).to(dev)
    big.x = big.x.to(dtype)
    if not scn.resident_ok(big):
        pytest.skip("graphs do not fit the graph-resident stage A")
    pair = ScnTrainStep(scn, big, one_launch=False)
    if not ScnTrainStep(scn, big).one_launch:       # fits the pair but not the one launch: the default keeps the pair
        with pytest.raises(RuntimeError):
            ScnTrainStep(scn, big, one_launch=True)
        assert K % 4 or K > 32 or H * ((K + 15) // 16) > 32 or name == "pascalvoc_sp"
        pytest.skip("this shape does not fit the one-launch stage-A step: the default keeps the launch pair")
    one = ScnTrainStep(scn, big, one_launch=True)
    pair.run()
    one.run()
    torch.cuda.synchronize()
    pair.check(); one.check()
    assert torch.equal(one.S, pair.S)
    assert torch.equal(one.losses, pair.losses)
    assert bool(torch.isfinite(one.grads).all()) and float(one.grads.abs().max()) > 0
    off = 0
    ref = _scn_f64_grads(scn, one, graphs, H, act, K) if (dtype == torch.float32 and B <= 32) else None
    for p in one._mp:
        sl = slice(off, off + p.numel())
        off += p.numel()
        a, b_ = one.grads[sl].double().cpu(), pair.grads[sl].double().cpu()
        scale = max(float(b_.abs().max()), 1e-6)
        if ref is not None:
            e_one, e_pair = float((a - ref[sl]).abs().max()), float((b_ - ref[sl]).abs().max())
            assert e_pair <= 1e-4 * max(float(ref[sl].abs().max()), 1e-6), (e_pair, scale)
            assert e_one <= 2 * e_pair + 2e-6 * scale, (e_one, e_pair, scale)
        else:
            assert float((a - b_).abs().max()) <= 2e-5 * scale, (float((a - b_).abs().max()), scale)
    want = one.grads.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one.run()
    for _ in range(20):
        one.grads.zero_(); one.losses.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(one.grads, want)
        assert torch.equal(one.losses, pair.losses)
    one.check()


@pytest.mark.parametrize("name,B,K,H,L,C,dtype", [("peptides_func", 128, 16, 16, 3, 10, torch.float32),
                                                  ("peptides_func", 100, 16, 16, 3, 10, torch.float16),
                                                  ("peptides_struct", 32, 32, 16, 2, 11, torch.float32),
                                                  ("pcqm_contact", 120, 16, 32, 3, 1, torch.float32),
                                                  ("pcqm_contact", 256, 16, 16, 3, 1, torch.float16),   # 512 workgroups: CUs shared
                                                  ("pcqm_contact", 400, 16, 16, 2, 1, torch.float32),
                                                  ("peptides_func", 300, 16, 16, 3, 10, torch.float32),   # 600 workgroups of 16 waves:
                                                  ("peptides_func", 256, 16, 16, 2, 10, torch.float16),   # several rounds of the chip
                                                  ("peptides_func", 24, 4, 16, 1, 10, torch.float32)])
def test_one_launch_step_against_the_launch_pair_at_full_occupancy(name, B, K, H, L, C, dtype):
    """The one-launch step with every CU busy (up to 2B = 256 workgroups: B local programs, B virtual-branch
    programs that recompute the local chain they read) and graphs of very different sizes, eager and replayed,
    40 times: the final virtual features equal the launch pair's bit for bit every time; prediction, score and the
    gradients equal them to float rounding (the one-launch step pools out of the last layer's accumulators and, at
    H = 16, groups the weight gradient's partial sums by row tile) and are bitwise identical from run to run."""
    from tests.helpers import grads_close, pool_order_close
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.step import ResidentTrainStep
    dev = torch.device("cuda:0")
    graphs = make_dataset(name, B, seed=B)
    rng = np.random.default_rng(B)
    hs = [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    for h in hs:
        h["local"].y = torch.from_numpy((rng.random((1, C)) < 0.4).astype(np.float32))
    d = HeteroBatch.from_data_list(hs).to(dev).with_feature_dtype(dtype)
    torch.manual_seed(1)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], d["local"].x.size(1), H, C, L).to(dev)
    pair = ResidentTrainStep(model, d, "cross_entropy", one_launch=False)
    pair.run()
    one = ResidentTrainStep(model, d, "cross_entropy", one_launch=True)
    assert one.idle_cus and one.one_launch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        one.run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one.run()
    for it in range(40):
        one.virtual.fill_(-1.0)
        one.grads.zero_()
        if it % 2:
            g.replay()
        else:
            one.run()
        torch.cuda.synchronize()
        assert torch.equal(one.virtual, pair.virtual), it
        if it == 0:
            first, first_pred, first_score = one.grads.clone(), one.pred.clone(), one.score.clone()
            assert pool_order_close(one.pred, pair.pred) and pool_order_close(one.score, pair.score)
            assert grads_close(one.grads[:-1], pair.grads[:-1], rel=1e-5) and pool_order_close(one.grads[-1], pair.grads[-1])
        assert torch.equal(one.pred, first_pred) and torch.equal(one.score, first_score), it
        assert torch.equal(one.grads, first), it          # bitwise reproducible
    one.check()


@pytest.mark.parametrize("kind,wd", [("adam", 0.0), ("adam", 0.01), ("adamW", 0.01)])
def test_flat_adam_is_torch_adam(kind, wd):
    """optim.FlatAdam (one launch on the flat gradient buffer, hscn_adam_step) against torch.optim.Adam / AdamW on
    the CPU (single-tensor implementation, the oracle's optimizer) fed the same gradients for 60 steps, with a
    learning-rate change on the way: parameters agree to float rounding at every step (1e-6 of the parameter scale --
    the two sides differ only in fma contraction and the association of addcdiv)."""
    from graph_hscn.optim import FlatAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(16, 9), (16,), (16, 9), (16, 16), (16,), (7, 3, 5)]
    cpu_p = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    dev_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in cpu_p]
    P = sum(p.numel() for p in cpu_p)
    flat = torch.zeros(P + 1, device=dev)        # (+ 1: the resident steps keep a loss column behind the gradients)
    views, off = [], 0
    for p in dev_p:
        views.append((p, flat[off: off + p.numel()].view_as(p)))
        off += p.numel()
    opt_cls = torch.optim.Adam if kind == "adam" else torch.optim.AdamW
    ref = opt_cls(cpu_p, lr=0.01, weight_decay=wd)
    mine = FlatAdam.from_config(kind, views, flat, 0.01, wd)
    assert FlatAdam.from_config("adagrad", views, flat, 0.01, wd) is None
    g = torch.cuda.CUDAGraph()
    mine.step()                                  # warm-up launch, undone below
    torch.cuda.synchronize()
    with torch.no_grad():
        for p, q in zip(dev_p, cpu_p):
            p.copy_(q)
    mine.reset_state()
    with torch.cuda.graph(g):
        mine.step()
    with torch.no_grad():                        # (the capture does not execute)
        assert all(torch.equal(p.cpu(), q) for p, q in zip(dev_p, cpu_p))
    for it in range(60):
        if it == 30:
            for grp in ref.param_groups:
                grp["lr"] = 0.003
            mine.set_lr(0.003)
        grads = [torch.randn(s) * (0.1 + it % 3) for s in shapes]
        for p, gr in zip(cpu_p, grads):
            p.grad = gr
        flat[:P].copy_(torch.cat([gr.reshape(-1) for gr in grads]))
        ref.step()
        g.replay() if it % 2 else mine.step()
        torch.cuda.synchronize()
        for p, q in zip(dev_p, cpu_p):
            scale = max(1.0, float(q.abs().max()))
            assert float((p.detach().cpu() - q.detach()).abs().max()) <= 1e-6 * scale, (it, kind)
    assert float(mine.step_count) == 60.0
    mine.check()


@pytest.mark.parametrize("kind,wd,dtype", [("adam", 0.0, torch.float32), ("adamW", 0.01, torch.float32),
                                           ("adam", 0.01, torch.float16)])
def test_scn_step_with_the_optimizer_in_its_tail(kind, wd, dtype):
    """One graph per step (the reference's trajectory, train_clustering.py:36-50): ``ScnTrainStep.run(opt=FlatAdam)``
    -- forward, losses, backward AND optimizer.step() in one launch -- against the same step followed by
    ``FlatAdam.step()`` as its own launch, over 12 visits of 3 graphs, with and without the per-graph structure cache:
    parameters, moments, gradients and losses bit for bit."""
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import SCN
    from graph_hscn.optim import FlatAdam
    from graph_hscn.step import ScnTrainStep, ScnWorkspace
    dev = torch.device("cuda:0")
    graphs = [g.to(dev) for g in make_dataset("peptides_func", 3, seed=9)]
    for g in graphs:
        g.x = g.x.to(dtype)

    def run(fuse, cache=False):
        from graph_hscn.step import ScnStructurePool
        torch.manual_seed(4)
        scn = SCN([16], "elu", 9, 16).to(dev)
        ws = ScnWorkspace(dev, max(g.num_nodes for g in graphs), max(g.edge_index.size(1) for g in graphs), 1, 9, 16, 16,
                          dtype)
        pool = ScnStructurePool(dev, sum(g.num_nodes for g in graphs), sum(g.edge_index.size(1) for g in graphs),
                                len(graphs)) if cache else None
        steps = [ScnTrainStep(scn, g, workspace=ws, structure_pool=pool) for g in graphs]
        opt = FlatAdam.from_config(kind, steps[0].param_grads, ws.grads, 0.01, wd)
        assert all(st.fuses_optimizer(opt) for st in steps)
        losses = []
        for _ in range(4):
            for st in steps:
                if fuse:
                    st.run(opt=opt)
                else:
                    st.run()
                    opt.step()
                losses.append(st.losses.clone())
        torch.cuda.synchronize()
        steps[0].check()
        return [p.detach().clone() for p in scn.parameters()], opt.exp_avg.clone(), opt.exp_avg_sq.clone(), \
            float(opt.step_count), ws.grads.clone(), torch.stack(losses)

    b = run(False)
    # ... and with the graphs' CSRs / out-degrees / A_hat x kept in HBM after the first visit (ScnStructurePool)
    for a in (run(True), run(True, cache=True), run(False, cache=True)):
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3] == 12.0
        assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])


def test_one_launch_step_row_records_and_csr_walk_in_one_launch():
    """The one-launch step keeps a graph's ll structure as 16-byte row records when no node has more than six edges and
    takes the general CSR build + walk otherwise -- per workgroup.  A batch that mixes molecule-like graphs with graphs
    that carry a hub (in-degree 7 .. 40, and one node that is the SOURCE of many edges) runs both forms in one launch:
    virtual features equal the launch pair's bit for bit, prediction / score / gradients to float rounding, and the
    prediction equals the CPU oracle's to 1e-5."""
    from oracle import models as OM
    from tests.helpers import ATOL, grads_close, pool_order_close
    from graph_hscn.config.config import ACT_DICT
    from graph_hscn.data import Data, HeteroBatch
    from graph_hscn.loader.hetero_data import hetero_from_clusters
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.model.hscn import HSCN
    from graph_hscn.step import ResidentTrainStep
    dev = torch.device("cuda:0")
    K, C, H, L = 8, 10, 16, 3
    rng = np.random.default_rng(11)
    graphs = make_dataset("peptides_func", 10, seed=21)
    for gi, hub_deg in ((2, 7), (5, 40), (8, 13)):          # add a hub to three of them
        g = graphs[gi]
        n = g.num_nodes
        leaves = torch.from_numpy(rng.choice(np.arange(1, n), size=min(hub_deg, n - 1), replace=False))
        hub = torch.zeros_like(leaves)
        extra = torch.cat([torch.stack([leaves, hub]), torch.stack([hub, leaves])], 1) if gi != 8 else torch.stack([hub, leaves])
        graphs[gi] = Data(x=g.x, edge_index=torch.cat([g.edge_index, extra], 1).contiguous(), y=g.y, num_nodes=n)
    hs = [hetero_from_clusters(g, rng.integers(0, K, g.num_nodes), K) for g in graphs]
    for h in hs:
        h["local"].y = torch.from_numpy((rng.random((1, C)) < 0.4).astype(np.float32))
    host = HeteroBatch.from_data_list(hs)
    d = host.to(dev)
    torch.manual_seed(3)
    model = HSCN("GAT", "GCN", "GCN", ACT_DICT["relu"], d["local"].x.size(1), H, C, L).to(dev)
    pair = ResidentTrainStep(model, d, "cross_entropy", one_launch=False)
    pair.run()
    one = ResidentTrainStep(model, d, "cross_entropy", one_launch=True)
    assert one.one_launch
    one.run()
    torch.cuda.synchronize()
    one.check()
    assert torch.equal(one.virtual, pair.virtual)
    assert pool_order_close(one.pred, pair.pred) and pool_order_close(one.score, pair.score)
    assert grads_close(one.grads[:-1], pair.grads[:-1], rel=1e-5)
    ref = OM.HSCN("GAT", "GCN", "GCN", OM.ACT["relu"], d["local"].x.size(1), H, C, L)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    with torch.no_grad():
        want = ref({k: v.float() for k, v in host.x_dict.items()}, host.edge_index_dict, host["local"].batch, host.num_graphs)
    assert torch.allclose(one.pred.cpu(), want, atol=ATOL, rtol=1e-5), float((one.pred.cpu() - want).abs().max())
