"""The C-ABI library loads and exports exactly what include/tasu_hip.h declares (no compute calls: CPU box)."""
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "tasu_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return {m.group(1): m.group(2) for m in re.finditer(r"\bint(?:64_t)?\s+(tasu_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S)}


def test_every_declared_symbol_is_exported_and_bound():
    from ps_slm_amd import _lib
    decl = header_functions()
    assert len(decl) >= 30
    lib = _lib.load()
    assert lib.tasu_abi_version() == _lib.ABI_VERSION
    assert set(decl) == set(_lib.PROTOTYPES), set(decl) ^ set(_lib.PROTOTYPES)
    for name, args in decl.items():
        assert hasattr(lib, name), name
        n_args = 0 if args.strip() in ("", "void") else args.count(",") + 1
        assert n_args == len(_lib.PROTOTYPES[name]), (name, n_args, len(_lib.PROTOTYPES[name]))


def test_signatures_are_plain_c():
    for name, args in header_functions().items():
        assert "torch" not in args and "at::" not in args and "std::" not in args, name


def test_bad_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised here: rc == 1 (TASU_ERR_ARG)."""
    from ps_slm_amd import _lib
    lib = _lib.load()
    assert lib.tasu_gemm_nt_bf16(None, 64, None, 64, None, 64, None, None, 64, 64, 64, 0, None) == 1
    assert lib.tasu_rmsnorm_fwd(None, None, None, None, 4, 6, 1e-6, None) == 1
    assert lib.tasu_adamw(None, None, None, None, None, 0, 5e-5, 0.9, 0.999, 1e-6, 0.0, 1, 1.0, None) == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ps_slm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TasuLibraryError):
        _lib.load()


def test_hipops_refuses_cpu_box():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ps_slm_amd.ops import HipOps, TasuOpError
    with pytest.raises(TasuOpError):
        HipOps()
