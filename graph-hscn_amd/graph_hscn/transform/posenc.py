"""Laplacian eigen-decomposition statistics for the SignNet positional encoding
(/root/reference/graph_hscn/transform/posenc.py:14-107), same names and return conventions.

This is dataset PRE-PROCESSING in the reference (loader/loader.py:74-90: once per graph, on the host, numpy ``eigh``),
not part of the training step.  Two back ends for the decomposition:

* ``device=None`` (default): the reference's own path -- dense Laplacian with numpy, ``np.linalg.eigh`` on the host;
* ``device="cuda"``: the batched form -- the graphs' Laplacians are assembled on the GPU as one zero-padded
  ``[B, nmax, nmax]`` tensor (padding rows carry a diagonal entry above the graph's spectrum, so their eigenpairs sort behind every real
  one and cannot mix with them) and decomposed by ONE batched ``torch.linalg.eigh`` (rocSOLVER): SURVEY.md 8(f)4.

Eigenvectors are defined up to sign, and up to a rotation inside a repeated eigenvalue's eigenspace (molecules have
many): the two back ends agree on eigenvalues, on ``L v = lambda v`` and on the projector of every eigenspace, not on
the entries of ``v`` -- which is exactly the ambiguity SignNet (encoder/signnet.py) is built to be invariant to."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


def _undirected(edge_index: Tensor) -> Tensor:
    """PyG ``to_undirected`` without edge attributes: both directions, duplicates merged."""
    ei = torch.cat([edge_index, edge_index.flip(0)], 1)
    return torch.unique(ei, dim=1)


def _dense_laplacian(edge_index: np.ndarray, N: int, normalization: Optional[str]) -> np.ndarray:
    """PyG ``get_laplacian`` densified (``to_scipy_sparse_matrix(...).toarray()``): self loops removed, unit weights,
    duplicate edges summed; None: D - A; "sym": I - D^-1/2 A D^-1/2; "rw": I - D^-1 A."""
    row, col = edge_index
    keep = row != col
    row, col = row[keep], col[keep]
    A = np.zeros((N, N), dtype=np.float32)
    np.add.at(A, (row, col), np.float32(1.0))
    deg = np.zeros(N, dtype=np.float32)
    np.add.at(deg, row, np.float32(1.0))
    if normalization is None:
        return (np.diag(deg) - A).astype(np.float32)
    if normalization == "sym":
        with np.errstate(divide="ignore"):
            dis = deg ** np.float32(-0.5)
        dis[np.isinf(dis)] = 0.0
        return (np.eye(N, dtype=np.float32) - dis[:, None] * A * dis[None, :]).astype(np.float32)
    if normalization == "rw":
        with np.errstate(divide="ignore"):
            di = np.float32(1.0) / deg
        di[np.isinf(di)] = 0.0
        return (np.eye(N, dtype=np.float32) - di[:, None] * A).astype(np.float32)
    raise ValueError(f"unknown Laplacian normalization {normalization!r}")


def compute_posenc_stats(data, is_undirected: bool, cfg):
    """posenc.py:14-48: attaches ``eigvals_sn`` [N, max_freqs, 1] and ``eigvecs_sn`` [N, max_freqs] to ``data``."""
    N = int(data.num_nodes) if hasattr(data, "num_nodes") else int(data.x.shape[0])
    norm_type = cfg.eigen_laplacian_norm.lower()
    if norm_type == "none":
        norm_type = None
    ei = data.edge_index if is_undirected else _undirected(data.edge_index)
    L = _dense_laplacian(ei.cpu().numpy(), N, norm_type)
    evals_sn, evects_sn = np.linalg.eigh(L)
    data.eigvals_sn, data.eigvecs_sn = get_lap_decomp_stats(evals=evals_sn, evects=evects_sn,
                                                            max_freqs=cfg.eigen_max_freqs, eigvec_norm=cfg.eigvec_norm)
    return data


def compute_posenc_stats_batched(graphs: Sequence, is_undirected: bool, cfg, device="cuda") -> List:
    """The same statistics for a list of graphs with ONE batched device ``eigh`` (see the module docstring)."""
    norm_type = cfg.eigen_laplacian_norm.lower()
    if norm_type == "none":
        norm_type = None
    ns = [int(g.num_nodes) if hasattr(g, "num_nodes") else int(g.x.shape[0]) for g in graphs]
    nmax = max(ns)
    dev = torch.device(device)
    Ls = torch.zeros(len(graphs), nmax, nmax, dtype=torch.float32)
    for b, (g, n) in enumerate(zip(graphs, ns)):
        ei = g.edge_index if is_undirected else _undirected(g.edge_index)
        L = torch.from_numpy(_dense_laplacian(ei.cpu().numpy(), n, norm_type))
        Ls[b, :n, :n] = L
        if n < nmax:
            # padding: eigenvector e_i with an eigenvalue just above the graph's own spectrum (Gershgorin bound of the
            # matrix eigh reads, whichever triangle), so the graph's n pairs come first and keep their f32 precision
            bound = float(max(L.abs().sum(0).max(), L.abs().sum(1).max())) + 1.0
            idx = torch.arange(n, nmax)
            Ls[b, idx, idx] = bound
    evals, evects = torch.linalg.eigh(Ls.to(dev))
    evals, evects = evals.cpu().numpy(), evects.cpu().numpy()
    for b, (g, n) in enumerate(zip(graphs, ns)):
        g.eigvals_sn, g.eigvecs_sn = get_lap_decomp_stats(evals=evals[b, :n], evects=evects[b, :n, :n],
                                                        max_freqs=cfg.eigen_max_freqs, eigvec_norm=cfg.eigvec_norm)
    return list(graphs)


def get_lap_decomp_stats(evals, evects, max_freqs: int, eigvec_norm: str = "L2") -> Tuple[Tensor, Tensor]:
    """posenc.py:51-83: the ``max_freqs`` smallest eigenpairs, eigenvalues clamped at 0, eigenvectors normalised,
    both NaN-padded to ``max_freqs`` columns; eigenvalues repeated per node -> [N, max_freqs, 1]."""
    N = len(evals)
    idx = evals.argsort()[:max_freqs]
    evals, evects = evals[idx], np.real(evects[:, idx])
    evals = torch.from_numpy(np.real(evals)).clamp_min(0)
    evects = torch.from_numpy(np.ascontiguousarray(evects)).float()
    evects = eigvec_normalizer(evects, evals, normalization=eigvec_norm)
    if N < max_freqs:
        eig_vecs = F.pad(evects, (0, max_freqs - N), value=float("nan"))
        eig_vals = F.pad(evals, (0, max_freqs - N), value=float("nan")).unsqueeze(0)
    else:
        eig_vecs = evects
        eig_vals = evals.unsqueeze(0)
    eig_vals = eig_vals.repeat(N, 1).unsqueeze(2)
    return eig_vals, eig_vecs


def eigvec_normalizer(eig_vecs: Tensor, eig_vals: Tensor, normalization: str = "L2", eps: float = 1e-12) -> Tensor:
    """posenc.py:86-107."""
    if normalization == "L1":
        denom = eig_vecs.norm(p=1, dim=0, keepdim=True)
    elif normalization == "L2":
        denom = eig_vecs.norm(p=2, dim=0, keepdim=True)
    elif normalization == "abs-max":
        denom = torch.max(eig_vecs.abs(), dim=0, keepdim=True).values
    else:
        raise ValueError(f"Unsupported normalization `{normalization}`")
    denom = denom.clamp_min(eps).expand_as(eig_vecs)
    return eig_vecs / denom
