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


def _prototypes():
    """name -> (return kind, [parameter kinds]) parsed from the header's prototypes; a kind is
    'ptr', 'i32', 'i64', 'u32', 'u64', 'size', 'f32' or 'str' (const char* return)."""
    text = open(os.path.join(ROOT, "include", "sage_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)

    def kind(decl):
        decl = decl.strip()
        if "*" in decl:
            return "str" if re.match(r"const\s+char\s*\*", decl) else "ptr"
        base = re.sub(r"\b(const|[A-Za-z_][A-Za-z0-9_]*)$", lambda m: m.group(0) if m.group(0) in
                      ("int", "float", "size_t", "sage_i64", "uint32_t", "uint64_t", "void", "long") else "", decl).strip()
        base = re.sub(r"\bconst\b", "", base).strip()
        return {"int": "i32", "float": "f32", "size_t": "size", "sage_i64": "i64", "uint32_t": "u32",
                "uint64_t": "u64", "unsigned long long": "u64"}[base]

    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(sage_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        params = params.strip()
        plist = [] if params in ("", "void") else [kind(q) for q in params.split(",")]
        out[name] = ("str" if "char" in ret else kind(ret + " x") if "*" not in ret else "ptr", plist)
    return out


def test_binding_matches_every_prototype():
    """A parameter added to a prototype but not to the ctypes argtypes (or an int where the
    header says sage_i64) would pass a names-only check and corrupt the call: compare each
    prototype's parameter count and kinds with graphsage_amd._lib.SIGNATURES."""
    from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64, c_void_p
    from graphsage_amd import _lib
    kinds = {c_void_p: "ptr", c_int: "i32", c_int64: "i64", c_uint32: "u32", c_uint64: "u64", c_size_t: "size",
             c_float: "f32", c_char_p: "str", ctypes.POINTER(c_int): "ptr"}
    protos = _prototypes()
    same = {"size": "u64"}      # (c_size_t IS c_uint64 on this platform: one ctypes class)
    protos = {k: (same.get(r, r), [same.get(q, q) for q in ps]) for k, (r, ps) in protos.items()}
    kinds = {k: same.get(v, v) for k, v in kinds.items()}
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        ret, params = protos[name]
        assert kinds[restype] == ret, (name, "return", kinds[restype], ret)
        got = [kinds[a] for a in argtypes]
        assert got == params, (name, got, params)


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
