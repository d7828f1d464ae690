"""The C-ABI library loads on a CPU-only box and exports every symbol include/dxmi_hip.h declares
(no compute calls here: there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dxmi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dxmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from dxmi_hip import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dxmi_hip.h but not exported"


def test_ctypes_table_matches_header():
    from dxmi_hip import _lib
    assert sorted(_lib.SIGNATURES.keys()) == header_symbols()
    lib = _lib.load()
    assert lib.dxmi_version() >= 100
    assert isinstance(lib.dxmi_last_error(), bytes)


def test_conv_desc_layout_matches_header():
    """ctypes mirror has the same field order as struct dxmi_conv_desc."""
    from dxmi_hip._lib import ConvDesc
    text = open(os.path.join(ROOT, "include", "dxmi_hip.h")).read()
    body = re.search(r"typedef struct dxmi_conv_desc \{(.*?)\} dxmi_conv_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split()
        # "int32_t N, IH, IW" style multi-declarations
        first = parts.index(next(p for p in parts if p not in ("const", "void", "float", "int32_t")))
        names += [n.strip(", ") for n in " ".join(parts[first:]).split(",")]
    assert names == [f[0] for f in ConvDesc._fields_]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dxmi_hip import DxmiError, ops
    with pytest.raises(DxmiError):
        ops.device_check()
    with pytest.raises(DxmiError):
        ops.pool_act(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16), False, 0)
