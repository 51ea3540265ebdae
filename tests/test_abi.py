"""The C-ABI library loads and exports every symbol include/hscn.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hscn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hscn_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("hscn_csr_build", "hscn_spmm_csr_gcn", "hscn_spmm_csr_weighted", "hscn_gat_segment_fwd",
                 "hscn_gat_segment_bwd_dst", "hscn_gat_segment_bwd_src", "hscn_segment_mean_fwd",
                 "hscn_mincut_sparse_fwd", "hscn_mincut_sparse_bwd", "hscn_assign_argmax", "hscn_linear_fwd",
                 "hscn_resident_fwd", "hscn_resident_bwd"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from graph_hscn import _hip
    lib = _hip.lib()                       # raises if the .so is missing: there is no fallback
    for name in _declared():
        assert hasattr(lib, name), f"libhscn.so lacks {name}"
    assert lib.hscn_abi_version() == _hip.ABI_VERSION


def test_python_binding_covers_the_header():
    from graph_hscn import _hip
    assert sorted(_hip.exported_symbols()) == _declared()


def test_error_strings_and_argument_checks_without_a_gpu():
    from graph_hscn import _hip
    lib = _hip.lib()
    assert lib.hscn_strerror(0) == b"ok"
    assert b"bad argument" in lib.hscn_strerror(-1)
    # argument validation happens before any launch
    assert lib.hscn_csr_build(None, None, -1, 4, 4, None, None, None, None, None, 0, None) == -1
    assert lib.hscn_resident_supported(9, 16, 3, 10, 444, 16, 1000, 136) == 1
    assert lib.hscn_resident_supported(9, 24, 3, 10, 444, 16, 1000, 136) == 0     # H must be 16/32/64
    assert lib.hscn_resident_supported(9, 16, 3, 10, 5000, 16, 10000, 136) == 0   # does not fit LDS
    assert lib.hscn_resident_param_count(9, 16, 3, 10) == 9 * 16 + 16 + 2 * (256 + 16) + 256 + 16 + 160 + 10


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from graph_hscn import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_hip.HipExtensionMissing):
        _hip.lib()
