"""The C-ABI library loads and exports every symbol include/sage_hip.h declares, and the
ctypes binding covers exactly that set (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sage_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(sage_[a-z0-9_]+)\s*\(", text))


def test_header_symbols_exported_and_bound():
    from graphsage_amd import _lib
    declared = _declared()
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sage_hip.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    loaded = _lib.load()
    assert loaded.sage_abi_version() == 2
    assert loaded.sage_last_error() is not None


def test_no_cpu_fallback():
    import pytest
    import torch
    from graphsage_amd import ops
    with pytest.raises(RuntimeError):
        ops.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        ops.linear_concat_fwd(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "graphsage_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
