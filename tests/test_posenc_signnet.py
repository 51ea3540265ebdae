"""SignNet positional encoding (reference transform/posenc.py, encoder/signnet.py).

CPU: the Laplacian statistics of graph_hscn.transform against closed forms and against the oracle's restatement.
GPU: the encoder through the HIP operators against the oracle with identical weights, its sign invariance, and the
batched device decomposition against the host one in the quantities that are well defined (eigenvalues, the
eigen-equation, eigenspace projectors)."""
import math

import numpy as np
import pytest
import torch

from oracle import signnet as OS


def _path(n):
    a = torch.arange(n - 1)
    return torch.cat([torch.stack([a, a + 1]), torch.stack([a + 1, a])], 1)


class _G:
    pass


def _graph(n, ei, F=9):
    g = _G()
    g.x = torch.randn(n, F)
    g.edge_index = ei
    g.num_nodes = n
    return g


def test_path_graph_laplacian_spectrum_is_the_closed_form():
    """Unnormalised Laplacian of the path P_n: eigenvalues 2 - 2 cos(pi k / n), k = 0 .. n-1; L2-normalised
    eigenvectors have unit norm; eigenvalues are repeated per node and NaN-padded beyond n."""
    from graph_hscn.config.config import PEConfig
    from graph_hscn.transform import compute_posenc_stats
    n, mf = 7, 10
    cfg = PEConfig(9, 16, 8, eigen_max_freqs=mf, eigen_laplacian_norm="none")
    g = compute_posenc_stats(_graph(n, _path(n)), True, cfg)
    want = np.array([2 - 2 * math.cos(math.pi * k / n) for k in range(n)])
    assert g.eigvals_sn.shape == (n, mf, 1) and g.eigvecs_sn.shape == (n, mf)
    np.testing.assert_allclose(g.eigvals_sn[0, :n, 0].numpy(), want, atol=2e-6)
    assert torch.equal(g.eigvals_sn[0, :n], g.eigvals_sn[n - 1, :n])
    assert torch.isnan(g.eigvals_sn[:, n:, 0]).all() and torch.isnan(g.eigvecs_sn[:, n:]).all()
    np.testing.assert_allclose(g.eigvecs_sn[:, :n].norm(dim=0).numpy(), np.ones(n), atol=1e-6)


@pytest.mark.parametrize("lap", ["sym", "rw", "none"])
@pytest.mark.parametrize("vn", ["L1", "L2", "abs-max"])
def test_posenc_stats_equal_the_oracle(lap, vn):
    from graph_hscn.config.config import PEConfig
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.transform import compute_posenc_stats
    cfg = PEConfig(9, 16, 8, eigen_max_freqs=6, eigen_laplacian_norm=lap, eigvec_norm=vn)
    for g in make_dataset("pcqm_contact", 4, seed=3):
        got = compute_posenc_stats(g, True, cfg)
        ev, vec = OS.posenc_stats(g.edge_index, g.num_nodes, 6, lap, vn)
        np.testing.assert_allclose(got.eigvals_sn.numpy(), ev.numpy(), atol=2e-5, equal_nan=True)
        # eigenvectors up to sign per column (degenerate columns: compare the eigen-equation instead)
        L = OS.laplacian_dense(g.edge_index, g.num_nodes, None if lap == "none" else lap, True).astype(np.float64)
        L = np.tril(L) + np.tril(L, -1).T   # eigh reads the lower triangle: for "rw" (not symmetric) that is the matrix decomposed
        V = got.eigvecs_sn.numpy().astype(np.float64)
        lam = np.linalg.eigvalsh(L)[:6]     # unclamped (the stored ones are clamped at 0, posenc.py:103)
        np.testing.assert_allclose(np.maximum(lam, 0), got.eigvals_sn[0, :, 0].numpy(), atol=2e-5)
        assert np.abs(L @ V - V * lam[None, :]).max() < 5e-5 * max(1.0, np.abs(V).max())


def test_directed_input_is_symmetrised_and_bad_normaliser_is_refused():
    from graph_hscn.config.config import PEConfig
    from graph_hscn.transform import compute_posenc_stats, eigvec_normalizer
    n = 6
    one_way = torch.stack([torch.arange(n - 1), torch.arange(1, n)])
    cfg = PEConfig(9, 16, 8, eigen_max_freqs=4, eigen_laplacian_norm="none")
    a = compute_posenc_stats(_graph(n, one_way), False, cfg)
    b = compute_posenc_stats(_graph(n, _path(n)), True, cfg)
    np.testing.assert_allclose(a.eigvals_sn.numpy(), b.eigvals_sn.numpy(), atol=1e-6)
    with pytest.raises(ValueError):
        eigvec_normalizer(torch.ones(3, 2), torch.ones(2), normalization="L7")


def _enc_pair(model, use_bn, seed):
    from graph_hscn.config.config import PEConfig
    from graph_hscn.encoder import SignNetNodeEncoder
    cfg = PEConfig(9, 16, 6, model=model, layers=3, post_layers=2, eigen_max_freqs=5, phi_hidden_dim=16, phi_out_dim=4, use_bn=use_bn)
    torch.manual_seed(seed)
    oe = OS.SignNetNodeEncoder(cfg, 9, 16)
    pe = SignNetNodeEncoder(cfg, 9, 16).to("cuda")
    assert sorted(pe.state_dict()) == sorted(oe.state_dict())
    pe.load_state_dict(oe.state_dict())
    return cfg, oe, pe


@pytest.mark.gpu
@pytest.mark.parametrize("model,use_bn", [("DeepSet", False), ("MLP", False), ("DeepSet", True)])
def test_signnet_encoder_matches_oracle_and_is_sign_invariant(model, use_bn):
    from graph_hscn.data import Batch
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.transform import compute_posenc_stats
    cfg, oe, pe = _enc_pair(model, use_bn, seed=4)
    graphs = make_dataset("pcqm_contact", 5, seed=9)
    graphs[0] = make_dataset("pcqm_contact", 40, seed=1)[np.argmin([g.num_nodes for g in make_dataset("pcqm_contact", 40, seed=1)])]
    for g in graphs:
        compute_posenc_stats(g, True, cfg)
    b = Batch.from_data_list(graphs)
    vec = torch.cat([g.eigvecs_sn for g in graphs], 0)
    oe.eval(); pe.eval()
    with torch.no_grad():
        want_x, want_pe = oe(b.x.float(), vec, b.edge_index, b.batch)
        d = b.to("cuda")
        d.eigvecs_sn, d.eigvals_sn = vec.to("cuda"), None
        d.x = d.x.float()
        out = pe(d)
        assert out.x.shape == (b.num_nodes, 16)
        assert torch.allclose(out.x.cpu(), want_x, atol=2e-5, rtol=1e-4), float((out.x.cpu() - want_x).abs().max())
        # flipping the sign of every eigenvector changes nothing (the point of SignNet)
        d2 = b.to("cuda")
        d2.eigvecs_sn, d2.eigvals_sn, d2.x = -vec.to("cuda"), None, d2.x.float()
        assert torch.allclose(pe(d2).x, out.x, atol=1e-6)


@pytest.mark.gpu
def test_batched_device_decomposition_agrees_with_the_host_one():
    from graph_hscn.config.config import PEConfig
    from graph_hscn.loader.synthetic import make_dataset
    from graph_hscn.transform.posenc import compute_posenc_stats, compute_posenc_stats_batched
    cfg = PEConfig(9, 16, 8, eigen_max_freqs=8, eigen_laplacian_norm="sym")
    gs_h = make_dataset("pcqm_contact", 12, seed=2)
    gs_d = make_dataset("pcqm_contact", 12, seed=2)
    for g in gs_h:
        compute_posenc_stats(g, True, cfg)
    compute_posenc_stats_batched(gs_d, True, cfg, device="cuda")
    for a, b in zip(gs_h, gs_d):
        np.testing.assert_allclose(a.eigvals_sn.numpy(), b.eigvals_sn.numpy(), atol=2e-5, equal_nan=True)
        L = OS.laplacian_dense(a.edge_index, a.num_nodes, "sym", True).astype(np.float64)
        k = min(8, a.num_nodes)
        V = b.eigvecs_sn[:, :k].numpy().astype(np.float64)
        lam = b.eigvals_sn[0, :k, 0].numpy().astype(np.float64)
        assert np.abs(L @ V - V * lam[None, :]).max() < 1e-4
        # projectors of the eigenspaces agree where the spectrum has a gap after the k-th value
        ev_all = np.linalg.eigvalsh(L)
        if k < a.num_nodes and ev_all[k] - ev_all[k - 1] > 1e-3:
            Va = a.eigvecs_sn[:, :k].numpy().astype(np.float64)
            assert np.abs(Va @ Va.T - V @ V.T).max() < 1e-3
