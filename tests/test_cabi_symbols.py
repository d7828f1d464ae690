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


def test_host_side_under_address_and_ub_sanitizers():
    """the host C-ABI shim under -fsanitize=address,undefined.  `make asan` compiles the HOST half of every
    source (no device code) with both sanitizers and links tests/host/cabi_malformed.c against it and a no-device HIP
    runtime stub: ~70 malformed calls (null pointers, Cout % 64 != 0, negative / zero sizes, over-long tensor lists)
    must each come back as DXMI_EINVAL with a message — no crash, no sanitizer report.  (Round 4: this found an integer
    division by zero in dxmi_conv2d_wgrad for C0 = 0 and four unchecked shapes.)  CPU only: never run on the GPU box."""
    import shutil
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("host-only sanitizer build: CPU boxes only")
    if not (shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no toolchain")
    csrc = os.path.join(ROOT, "diffusion-by-maxentirl_amd", "csrc")
    r = subprocess.run(["make", "-j8", "asan"], cwd=csrc, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run([os.path.join(csrc, "build_asan", "cabi_malformed")], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "0 failure(s)" in out, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert out.count("\nok ") + out.startswith("ok ") >= 60


def test_wgrad_workspace_bound_covers_every_split_choice():
    """Round-3 ADVICE (high): sizing and launch now share one split bound.  Host-only function: callable without a GPU."""
    from dxmi_hip import _lib
    lib = _lib.load()
    # 1x1, 256 -> 256 at N=16 16x16: the 128 x 128 kernel runs 64 splits of 64-pixel tiles (was sized for 48)
    assert lib.dxmi_conv2d_wgrad_workspace_bytes(16, 16, 16, 256, 256, 1) >= 64 * 256 * 256 * 4 + 64 * 4 * 256 * 4
    for (N, H, Cin, Cout, k) in [(1, 4, 64, 64, 1), (3, 8, 128, 128, 1), (24, 16, 128, 128, 1), (256, 32, 128, 128, 3), (16, 64, 192, 192, 3),
                                 (40, 8, 256, 128, 1), (16, 16, 768, 768, 1)]:
        npix = N * H * H
        tiles = (npix + 63) // 64 if k == 1 else (npix + 127) // 128
        s_max = min(max(1, (1024 if k == 1 else 512) // ((Cin // 64) * (Cout // 64))), tiles + 16)
        assert lib.dxmi_conv2d_wgrad_workspace_bytes(N, H, H, Cin, Cout, k) >= s_max * k * k * Cout * Cin * 4 + s_max * 4 * Cout * 4
    assert lib.dxmi_conv2d_wgrad_workspace_bytes(-1, 16, 16, 256, 256, 1) == 0
