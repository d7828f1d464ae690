"""FID statistics path (the half that needs no Inception weights).
CPU: the oracle restatement against the golden the REFERENCE's calculate_frechet_distance produced (tests/golden/make_golden.py
gen_fid), host-side pieces of the product module.  GPU: dxmi_fid_stats (f32 MFMA Gram matrix) against np.mean / np.cov and the
whole flow (extractor -> activations -> statistics on the device -> distance) against the reference's FID values at rel 1e-4."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import fid as ofid  # noqa: E402


def _case(name):
    return next(c for c in ofid.CASES if c[0] == name)


def _acts(name):
    _, dims, n1, n2, s1, s2, scale2, shift2 = _case(name)
    return ofid.synthetic_activations(s1, n1, dims), ofid.synthetic_activations(s2, n2, dims, scale=scale2, shift=shift2)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "fid_stats.npz"))


# ------------------------------------------------------------------------------------------------ CPU: oracle vs golden
@pytest.mark.parametrize("name", [c[0] for c in ofid.CASES])
def test_fid_oracle_matches_reference_golden(golden, name):
    a1, a2 = _acts(name)
    m1, c1 = ofid.activation_statistics(a1)
    m2, c2 = ofid.activation_statistics(a2)
    assert m1.dtype == np.float32 and c1.dtype == np.float64
    assert np.array_equal(m1, golden[f"{name}_mu1"]) and np.array_equal(m2, golden[f"{name}_mu2"])       # same expressions: bit-exact
    assert np.array_equal(c1[:8, :8], golden[f"{name}_corner1"])
    idx = golden[f"{name}_idx"]
    assert np.array_equal(c1[idx[:, 0], idx[:, 1]], golden[f"{name}_entries1"])
    assert np.isclose(np.trace(c1), golden[f"{name}_trace1"], rtol=1e-13) and np.isclose(np.linalg.norm(c2), golden[f"{name}_fro2"], rtol=1e-13)
    fid = ofid.frechet_distance(m1, c1, m2, c2)
    assert abs(fid - float(golden[f"{name}_fid"])) <= 1e-9 * abs(float(golden[f"{name}_fid"])), (fid, float(golden[f"{name}_fid"]))


def test_product_frechet_distance_is_the_reference_algorithm(golden):
    """Host function of the product module (no device needed): reference statistics in -> the reference's value out."""
    from pytorch_fid.fid_score import calculate_frechet_distance, load_statistics
    g = golden
    m1, m2 = g["d64_mu1"], g["d64_mu2"]
    fid = calculate_frechet_distance(m1, g["d64_sigma1"], m2, g["d64_sigma2"])
    assert abs(fid - float(g["d64_fid"])) <= 1e-10 * float(g["d64_fid"])
    with pytest.raises(AssertionError):
        calculate_frechet_distance(m1, g["d64_sigma1"], m2[:32], g["d64_sigma2"])
    # identical distributions: 0 up to the square root's rounding
    assert abs(calculate_frechet_distance(m1, g["d64_sigma1"], m1, g["d64_sigma1"])) < 1e-6


def test_statistics_file_and_extractor_loading(tmp_path, golden):
    from pytorch_fid.fid_score import load_extractor, load_statistics
    p = str(tmp_path / "VIRTUAL_test.npz")
    np.savez(p, mu=golden["d64_mu1"], sigma=golden["d64_sigma1"])
    m, s = load_statistics(p)
    assert m.dtype == np.float64 and s.shape == (64, 64) and np.array_equal(s, golden["d64_sigma1"])
    assert load_extractor("torch.nn:Identity").__class__.__name__ == "Identity"       # a class is instantiated
    with pytest.raises(ValueError):
        load_extractor("torch.nn")


def test_no_gpu_means_loud_failure_for_statistics():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dxmi_hip import DxmiError
    from pytorch_fid.fid_score import activation_statistics
    with pytest.raises(DxmiError):
        activation_statistics(torch.zeros(8, 64))


def test_fid_stats_workspace_plan_is_host_side():
    from dxmi_hip import _lib
    lib = _lib.load()
    nb = 16
    tiles = nb * (nb + 1) // 2
    need = lib.dxmi_fid_stats_workspace_bytes(50000, 2048)
    assert need >= 4 * tiles * 128 * 128 * 4                 # >= 4 row splits of the upper-triangular tiles
    assert lib.dxmi_fid_stats_workspace_bytes(1, 2048) == 0 and lib.dxmi_fid_stats_workspace_bytes(100, 0) == 0


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", [c[0] for c in ofid.CASES])
def test_fid_stats_kernel_vs_numpy(golden, name):
    from dxmi_hip import ops
    a1, _ = _acts(name)
    mu, sigma = ops.fid_stats(torch.from_numpy(a1).cuda())
    mu, sigma = mu.cpu().numpy(), sigma.cpu().numpy()
    m_ref, c_ref = np.mean(a1.astype(np.float64), axis=0), np.cov(a1, rowvar=False)
    assert mu.dtype == np.float64 and sigma.shape == c_ref.shape
    assert np.max(np.abs(mu - m_ref)) <= 1e-12 * max(1.0, np.max(np.abs(m_ref)))          # fp64 column means
    assert np.max(np.abs(mu - golden[f"{name}_mu1"])) < 1e-6                                  # the reference's float32 means
    assert np.array_equal(sigma, sigma.T)                                                     # mirrored, not recomputed
    rel = np.linalg.norm(sigma - c_ref) / np.linalg.norm(c_ref)
    assert rel < 5e-6, rel                                                                    # fp32 products, fp32 sums over <= 2048 rows per split (measured 2e-6)
    assert abs(np.trace(sigma) - float(golden[f"{name}_trace1"])) < 1e-6 * float(golden[f"{name}_trace1"])
    idx = golden[f"{name}_idx"]
    assert np.allclose(sigma[idx[:, 0], idx[:, 1]], golden[f"{name}_entries1"], rtol=0, atol=2e-6 * np.abs(c_ref).max())
    # bitwise reproducible (fixed-order fold) and independent of what the workspace held
    mu2, sigma2 = ops.fid_stats(torch.from_numpy(a1).cuda())
    assert torch.equal(torch.from_numpy(sigma), sigma2.cpu()) and torch.equal(torch.from_numpy(mu), mu2.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("name", [c[0] for c in ofid.CASES])
def test_fid_flow_matches_reference_value(golden, name):
    """extractor -> activations -> (gather) -> statistics on the device -> distance: the reference's FID at rel 1e-4 (measured
    ~1e-7).  The 'extractor' replays the synthetic activations (any callable with model(batch)[0] -> [B, dims, 1, 1])."""
    from pytorch_fid.fid_score import (activation_statistics, calculate_frechet_distance, gather_activations,
                                       get_activations_from_tensor)
    a1, a2 = _acts(name)
    dims = a1.shape[1]

    class Replay(torch.nn.Module):
        def __init__(self, table):
            super().__init__()
            self.table, self.pos = torch.from_numpy(table).cuda(), 0

        def forward(self, batch):
            rows = self.table[self.pos:self.pos + len(batch)]
            self.pos += len(batch)
            # a 2x2 feature map whose spatial mean is the row: exercises the pooling branch (fid_score.py:210-213)
            d = torch.tensor([[1.0, -1.0], [0.5, -0.5]], device=rows.device)
            return [rows[:, :, None, None] + d * rows[:, :, None, None]]

    stats = []
    for a in (a1, a2):
        images = torch.zeros(len(a), 3, 8, 8)
        act = get_activations_from_tensor(images, Replay(a), batch_size=50, dims=dims, device="cuda")
        assert act.shape == (len(a), dims) and act.is_cuda
        assert torch.allclose(act.cpu(), torch.from_numpy(a), rtol=1e-6, atol=1e-7)
        stats.append(activation_statistics(gather_activations(act)))
    fid = calculate_frechet_distance(stats[0][0], stats[0][1], stats[1][0], stats[1][1])
    want = float(golden[f"{name}_fid"])
    assert abs(fid - want) <= 1e-4 * want, (fid, want)
    print(f"FID {name}: device statistics {fid:.9f} vs reference {want:.9f} (rel {abs(fid - want) / want:.1e})")


@pytest.mark.gpu
def test_fid_stats_rejects_malformed_arguments():
    from dxmi_hip import DxmiError, ops
    with pytest.raises(DxmiError):
        ops.fid_stats(torch.zeros(1, 64, device="cuda"))          # N < 2: np.cov would divide by zero
    with pytest.raises(DxmiError):
        ops.fid_stats(torch.zeros(16, 66, device="cuda"))         # D % 4 != 0
