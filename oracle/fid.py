"""TEST INFRASTRUCTURE (oracle): CPU restatement of the FID statistics path, plus the seeded synthetic activations the
fixtures and the GPU parity tests share.  Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product (diffusion-by-maxentirl_amd/pytorch_fid) never does.

Restates (reference file:line):
  * activation statistics  — train_image_large.py:68, pytorch_fid/fid_score.py:300-302: np.mean(act, axis=0) (float32 in,
    float32 out), np.cov(act, rowvar=False) (float64).
  * Frechet distance       — pytorch_fid/fid_score.py:224-281.
Pinned by tests/golden/fid_stats.npz, which tests/golden/make_golden.py writes by calling the reference's own
`calculate_frechet_distance` on the reference's own statistics expressions (tests/test_oracle_golden.py::test_fid_*).
"""
import numpy as np
from scipy import linalg


def synthetic_activations(seed, n, dims, scale=1.0, shift=0.0, mix=0.5):
    """Deterministic pool3-like features: non-negative, correlated across neighbouring features, per-feature scales.
    np.random.RandomState is NumPy's frozen legacy stream, so every NumPy build draws the same numbers."""
    rs = np.random.RandomState(seed)
    z = rs.standard_normal((n, dims))
    per = 0.5 + rs.random_sample(dims)                     # per-feature scale in [0.5, 1.5)
    a = z * per * scale + shift + mix * np.roll(z, 1, axis=1)
    return np.maximum(a, 0.0).astype(np.float32)


def activation_statistics(act):
    """(mu float32 [D], sigma float64 [D, D]) exactly as the reference forms them (train_image_large.py:68)."""
    act = np.asarray(act)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:224-281: ||mu1 - mu2||^2 + Tr(s1) + Tr(s2) - 2 Tr(sqrtm(s1 s2)), eps retry on a singular product."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean = linalg.sqrtm(sigma1.dot(sigma2))
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


# the fixture's cases: (name, dims, n1, n2, seed1, seed2, scale2, shift2)
CASES = [
    ("d2048", 2048, 2341, 2500, 11, 12, 1.15, 0.05),       # pool3 size, N > dims, ragged row counts
    ("d64", 64, 301, 257, 21, 22, 0.9, 0.1),               # small: the full covariance is stored
    ("d192_singular", 192, 100, 120, 31, 32, 1.1, 0.0),    # N < dims: rank-deficient covariances (fid_score.py:258-264)
]
