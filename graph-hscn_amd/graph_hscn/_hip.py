"""ctypes binding of the C ABI in include/hscn.h (libhscn.so, gfx950).

The product path has no CPU or eager-PyTorch fallback: every operator in
``graph_hscn.nn`` goes through this library, and using one without it (or on a
CPU tensor) raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p
from typing import Optional

import torch

# HSCN_LIB: alternative build of the same ABI (the stamped diagnostic build used by tools/diag_resident.py)
_LIB_PATH = os.environ.get("HSCN_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhscn.so")
_lib: Optional[ctypes.CDLL] = None

ABI_VERSION = 18
ACT = {"identity": 0, "relu": 1, "elu": 2, "tanh": 3}

P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "hscn_abi_version": (c_int, []),
    "hscn_strerror": (c_char_p, [c_int]),
    "hscn_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "hscn_csr_build": (c_int, [P, P, c_int64, c_int64, c_int64, P, P, P, P, P, c_size_t, P]),
    "hscn_csr_pair_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "hscn_csr_build_pair": (c_int, [P, P, c_int64, c_int64, c_int64, P, P, P, P, P, P, P, P, c_size_t, P]),
    "hscn_csr_cross_positions": (c_int, [P, P, c_int64, P, P, P]),
    "hscn_gcn_dinv": (c_int, [P, c_int64, P, P]),
    "hscn_gcn_norm_weights": (c_int, [P, P, P, P, c_int64, P, P, P]),
    "hscn_linear_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, P]),
    "hscn_linear_bwd_w_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "hscn_linear_bwd_w": (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int, P, c_size_t, P]),
    "hscn_act_fwd": (c_int, [P, P, c_int64, c_int, P]),
    "hscn_act_bwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "hscn_dropout": (c_int, [P, P, c_int64, c_float, c_uint64, P]),
    "hscn_spmm_csr_gcn": (c_int, [P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    "hscn_spmm_csr_weighted": (c_int, [P, P, P, P, P, P, c_int64, c_int, P]),
    "hscn_gat_segment_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_float, c_int, c_int, P]),
    "hscn_gat_segment_bwd_dst": (c_int, [P, P, P, P, P, P, P, P, P, c_int64, c_int, c_float, P]),
    "hscn_gat_segment_bwd_src": (c_int, [P, P, P, P, P, P, P, P, P, c_int64, c_int, P]),
    "hscn_segment_mean_fwd": (c_int, [P, P, P, P, c_int64, c_int, P]),
    "hscn_segment_mean_bwd": (c_int, [P, P, P, P, c_int64, c_int, P]),
    "hscn_mincut_sparse_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int, c_int, P]),
    "hscn_mincut_sparse_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int, P]),
    "hscn_bgemm_f32": (c_int, [P, P, P, c_int64, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64,
                               c_int64, c_int, P]),
    "hscn_mincut_dense_fwd": (c_int, [P, P, P, c_int64, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "hscn_mincut_dense_bwd": (c_int, [P, P, P, P, P, P, P, c_int64, c_int, c_int, P, P, P, P, P]),
    "hscn_assign_argmax": (c_int, [P, P, c_int64, c_int, P]),
    "hscn_to_dense_adj": (c_int, [P, P, c_int64, c_int64, P, P]),
    "hscn_to_dense_adj_batched": (c_int, [P, P, c_int64, c_int64, c_int64, P, P]),
    "hscn_build_hetero_count": (c_int, [P, c_int, P, P, c_int64, c_int, c_int, P, P, P, P, P]),
    "hscn_build_hetero_scan": (c_int, [P, c_int64, P, P, P, P, P, P, P]),
    "hscn_build_hetero_emit": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_int, c_int64, c_int64, P, P, P, P, P]),
    "hscn_collate_gather": (c_int, [P, P, c_int64, P, P, P, P, P]),
    "hscn_criterion_fwd": (c_int, [P, P, c_int64, c_int, P, P, P, P]),
    "hscn_scale": (c_int, [P, P, P, c_int64, P]),
    "hscn_scn_resident_supported": (c_int, [c_int] * 5),
    "hscn_scn_resident_param_count": (c_int64, [c_int] * 3),
    "hscn_scn_resident_fwd": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int, P, P, P, P, P,
                                      c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "hscn_scn_resident_bwd": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int, P, P, P, P, P,
                                      P, P, P, P, P, P, P, P, c_int, c_int, P, P, P, P]),
    "hscn_scn_resident_train_step_supported": (c_int, [c_int] * 5),
    "hscn_scn_resident_train_step": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int, P, P, P,
                                             P, P, P, P, c_int, c_int, P, P, P, P, P, P, P, P, P, P]),
    "hscn_scn_resident_train_epoch": (c_int, [P, P, P, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, P, P, P, P, P,
                                              P, P, c_int, c_int, P, P, P, P, P, P, P, P]),
    "hscn_adam_step": (c_int, [P, P, c_int, P, P, P, c_int64, P, P, P, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                               ctypes.c_double, c_int, P]),
    "hscn_resident_supported": (c_int, [c_int] * 8),
    "hscn_resident_param_count": (c_int64, [c_int] * 4),
    "hscn_resident_fwd": (c_int, [P, P, P, c_int64, P, c_int64, P, c_int64, P, P, P, P, P, c_int64, c_int64,
                                  c_int64, c_int, c_int, c_int, c_int, c_int, c_float, P, P, P, P, P, c_int,
                                  c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P]),
    "hscn_resident_bwd": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P,
                                  P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, P, P, P, P]),
    "hscn_resident_fwd_with_virtual": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                               c_int, P, P, P, P, P, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P]),
    "hscn_resident_bwd_with_virtual": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                               c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, P, P, P, P, P]),
    "hscn_resident_train_step_supported": (c_int, [c_int] * 8),
    "hscn_resident_train_step_wgs_per_cu": (c_int, [c_int] * 8),
    "hscn_resident_train_step": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P,
                                         P, P, P, P, c_int, c_int, P, c_int, P, P, P, P, P, P, P, P, P, P]),
    "hscn_resident_structure": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, P, P, P, P, c_int64, c_int, c_int, c_int,
                                        c_int, P, P, P]),
    "hscn_collate_gather_structure": (c_int, [P, P, P, c_int64, P, P, P, P, P, P]),
    "hscn_mincut_dense_ragged_fwd": (c_int, [P, P, c_int, P, P, c_int64, c_int64, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "hscn_mincut_dense_ragged_bwd": (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int, c_int, P, P, P, P, P]),
    "hscn_dense_adj_s": (c_int, [P, c_int, P, P, c_int64, c_int, c_int, c_int, P, P, P]),
    "hscn_dense_adj_asymmetry_u8": (c_int, [P, c_int64, c_int, P, P]),
    "hscn_mincut_dense_ragged_bwd_sym": (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int, c_int, P, P, P, P, P, P]),
    "hscn_to_dense_adj_ragged_u8": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int, P, P, P]),
    "hscn_to_dense_adj_ragged": (c_int, [P, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int, P, P]),
    "hscn_gcn_norm_self_loops": (c_int, [P, P, P, c_int64, c_int64, c_float, P, P, P, P]),
    "hscn_norm_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "hscn_layer_norm_fwd": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_float, P]),
    "hscn_layer_norm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, P, c_size_t, P]),
    "hscn_batch_norm_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_float, c_float, c_int, P, c_size_t, P]),
    "hscn_batch_norm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, P, c_size_t, P]),
    "hscn_comm_alloc": (c_int, [c_size_t, c_int, P]),
    "hscn_comm_free": (c_int, [P]),
    "hscn_comm_ipc_export": (c_int, [P, P]),
    "hscn_comm_ipc_open": (c_int, [P, P]),
    "hscn_comm_ipc_close": (c_int, [P]),
    "hscn_allreduce_oneshot_slot_bytes": (c_size_t, [c_int64, c_int]),
    "hscn_allreduce_oneshot_flag_bytes": (c_size_t, [c_int64, c_int]),
    "hscn_allreduce_oneshot_chunks": (c_int64, [c_int64]),
    "hscn_allreduce_oneshot": (c_int, [P, c_int64, P, P, P, P, c_int, c_int, c_float, ctypes.c_uint32, P]),
}
# IEEE-half storage twins (include/hscn.h: hscn_resident_*_f16): same argument lists
for _n in ("hscn_resident_fwd", "hscn_resident_bwd", "hscn_resident_fwd_with_virtual", "hscn_resident_bwd_with_virtual",
           "hscn_scn_resident_fwd", "hscn_scn_resident_bwd", "hscn_resident_train_step",
           "hscn_scn_resident_train_step", "hscn_scn_resident_train_epoch"):
    _SIGNATURES[_n + "_f16"] = _SIGNATURES[_n]


class HipExtensionMissing(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libhscn.so once; fail loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipExtensionMissing(
            f"{_LIB_PATH} not found: the graph_hscn operators run only through the gfx950 HIP "
            "library (build it with `make -C graph-hscn_amd` or __graft_entry__.build()); "
            "there is no CPU/eager fallback.")
    L = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:  # pragma: no cover
            raise HipExtensionMissing(f"{_LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = L.hscn_abi_version()
    if v != ABI_VERSION:
        raise HipExtensionMissing(f"libhscn.so ABI {v} != expected {ABI_VERSION}")
    _lib = L
    return L


def exported_symbols():
    return list(_SIGNATURES)


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "materialize"):      # loss.LazyScaled reaching an operator that wants plain values
        t = t.materialize()
    if not t.is_cuda:
        raise RuntimeError(
            "graph_hscn operators take HIP device tensors only (got a CPU tensor): the hot path "
            "has no CPU fallback -- move the batch and model to 'cuda' (MI355X).")
    if not t.is_contiguous():
        raise RuntimeError("graph_hscn operators need contiguous tensors")
    return t.data_ptr()


def stream() -> int:
    """Raw handle of the current HIP stream of the current device (the private getter is ~20x cheaper than
    building a ``torch.cuda.Stream`` object; the eager path is host-bound)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def check(rc: int, name: str) -> None:
    if rc != 0:
        msg = lib().hscn_strerror(rc)
        raise RuntimeError(f"{name} failed with code {rc}: {msg.decode() if msg else '?'}")


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)
