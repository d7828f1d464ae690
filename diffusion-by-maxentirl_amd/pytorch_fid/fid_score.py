"""`pytorch_fid.fid_score` for the DxMI scripts — the FID evaluation flow of the reference
(train_image_large.py:56-87, generate_large.py:57-74, train_cifar10.py fid()) with its heavy host step on the device:

    act = get_activations_from_tensor(images, extractor, batch_size, dims, device)     # extractor: supplied by the user
    act = gather_activations(act)                                                      # all_gather over ranks (RCCL)
    m1, s1 = activation_statistics(act)                                                # np.mean / np.cov -> HIP (f32 MFMA)
    fid = calculate_frechet_distance(m1, s1, m2, s2)                                   # float64 host algorithm, as the reference

The feature extractor is an argument — any callable `extractor(batch) -> [features [B, dims, h, w]]` with the reference
extractor's call convention (`model(batch)[0]`, fid_score.py:208); `load_extractor("module:attr")` resolves one from the command
line.  Round 6: `pytorch_fid.inception.InceptionV3` is the reference's extractor on the HIP kernels (the published torchvision
Inception3 + the FID patches, one generic MFMA conv launch per BasicConv2d); what is still NOT here is its WEIGHT FILE
(pt_inception-2015-12-05: the reference downloads it, this image has no network) — `--fid_extractor
pytorch_fid.inception:FIDInceptionV3` loads it from DXMI_FID_WEIGHTS and fails loudly without it.

Reference statistics files (`datasets/VIRTUAL_*.npz`, `mu` / `sigma` keys, train_image_large.py:225-232) load through
`load_statistics`.  The names and argument meaning follow pytorch_fid/fid_score.py:170-281.
"""
import importlib

import numpy as np
import torch
from scipy import linalg

from dxmi_hip import ops
from dxmi_hip._lib import DxmiError


def load_extractor(spec):
    """'package.module:attribute' -> the object (a callable, or a zero-argument factory returning one)."""
    mod, _, name = spec.partition(":")
    if not name:
        raise ValueError(f"--fid_extractor expects 'module:attribute', got {spec!r}")
    obj = getattr(importlib.import_module(mod), name)
    return obj() if isinstance(obj, type) else obj


def load_statistics(path):
    """(mu, sigma) float64 numpy arrays of a reference statistics file (`mu` / `sigma` keys: fid_score.py:305-313)."""
    f = np.load(path, allow_pickle=True)
    try:
        m, s = f["mu"][:], f["sigma"][:]
    except (KeyError, IndexError, TypeError):
        m, s = f.item()["mu"][:], f.item()["sigma"][:]
    return np.asarray(m, dtype=np.float64), np.asarray(s, dtype=np.float64)


@torch.no_grad()
def get_activations_from_tensor(data, model, batch_size=50, dims=2048, device="cuda", resize=0):
    """Activations of `model` for every image of `data` [N, 3, H, W] in [0, 1] (reference: fid_score.py:170-221): batches of
    `batch_size` in order, `model(batch)[0]`, global average pooling when the feature map is not 1x1, rows kept on `device`.
    Returns fp32 [N, dims] on the device."""
    if hasattr(model, "eval"):
        model.eval()
    pred_arr = torch.zeros((len(data), dims), dtype=torch.float32, device=device)
    start = 0
    for i in range(0, len(data), batch_size):
        batch = data[i:i + batch_size].to(device)
        pred = model(batch)[0]
        if pred.dim() == 4:
            if pred.size(2) != 1 or pred.size(3) != 1:
                pred = pred.mean(dim=(2, 3), keepdim=True)          # adaptive_avg_pool2d(pred, (1, 1))
            pred = pred.squeeze(3).squeeze(2)
        if pred.shape[1] != dims:
            raise ValueError(f"the extractor returned {pred.shape[1]} features, dims = {dims}")
        pred_arr[start:start + pred.shape[0]] = pred.float()
        start += pred.shape[0]
    return pred_arr


def gather_activations(act):
    """All ranks' activation rows, rank-major (reference train_image_large.py:63-66: all_gather of equal-sized blocks, then
    torch.cat).  Over RCCL on the device; the identity on a single process."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return act
    parts = [torch.zeros_like(act) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, act.contiguous())
    return torch.cat(parts)


def activation_statistics(act):
    """(mu, sigma) = (np.mean(act, axis=0), np.cov(act, rowvar=False)) of fp32 activations [N, dims] (reference
    train_image_large.py:68, fid_score.py:300-302), computed on the device by `dxmi_fid_stats` (f32-input MFMA Gram matrix, fp64
    fold); returned as float64 numpy arrays, which is what `calculate_frechet_distance` takes.  Device tensors only: there is
    no CPU path in this package."""
    if not (torch.is_tensor(act) and act.is_cuda):
        raise DxmiError("pytorch_fid.activation_statistics needs a device tensor (the HIP path is the only one)")
    mu, sigma = ops.fid_stats(act.float().contiguous())
    return mu.cpu().numpy(), sigma.cpu().numpy()


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = ||mu1 - mu2||^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)) in float64 on the host, the algorithm of the reference
    (fid_score.py:224-281, D. Sutherland's stable form): scipy's Schur-based square root of the product, the eps-regularised
    retry when the product is singular, the imaginary-component check.  A 2048 x 2048 sqrtm is ~4 s on the host and runs once
    per evaluation; the N x 2048 x 2048 statistics in front of it are what the device kernel removes."""
    mu1, mu2 = np.atleast_1d(np.asarray(mu1)), np.atleast_1d(np.asarray(mu2))
    sigma1, sigma2 = np.atleast_2d(np.asarray(sigma1)), np.atleast_2d(np.asarray(sigma2))
    if mu1.shape != mu2.shape:
        raise AssertionError("Training and test mean vectors have different lengths")
    if sigma1.shape != sigma2.shape:
        raise AssertionError("Training and test covariances have different dimensions")
    diff = mu1 - mu2
    covmean = linalg.sqrtm(sigma1.dot(sigma2))
    if not np.isfinite(covmean).all():
        print(f"fid calculation produces singular product; adding {eps} to diagonal of cov estimates")
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


IMAGE_EXTENSIONS = {"bmp", "jpg", "jpeg", "pgm", "png", "ppm", "tif", "tiff", "webp"}


def compute_statistics_of_path(path, model, batch_size, dims, device, resize=0):
    """(mu, sigma) of an npz / npy statistics file, or of every image file of a folder (reference: fid_score.py:305-322;
    images are read with PIL as the reference's ImagePathDataset does, scaled to [0, 1], `resize` > 0 resizes bilinearly)."""
    import os
    import pathlib
    if path.endswith(".npz") or path.endswith(".npy"):
        return load_statistics(path)
    from PIL import Image
    files = sorted(f for ext in IMAGE_EXTENSIONS for f in pathlib.Path(path).glob(f"*.{ext}"))
    print(f"Found {len(files)} files")
    if not files:
        raise RuntimeError(f"no images under {path}")
    acts = []
    for i in range(0, len(files), batch_size):
        imgs = []
        for f in files[i:i + batch_size]:
            im = Image.open(f).convert("RGB")
            if resize > 0:
                im = im.resize((resize, resize), Image.BILINEAR)
            imgs.append(torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1))
        batch = torch.stack(imgs).to(device).float() / 255.0
        acts.append(get_activations_from_tensor(batch, model, batch_size=batch_size, dims=dims, device=device))
    return activation_statistics(torch.cat(acts))


def calculate_fid_given_paths(paths, batch_size, device, dims, resize=0, extractor=None):
    """FID of two paths (folders of images or statistics files), reference fid_score.py:325-343 — with the feature extractor
    as an ARGUMENT where the reference constructs InceptionV3 (weights unavailable here)."""
    import os
    for p in paths:
        if not os.path.exists(p):
            raise RuntimeError("Invalid path: %s" % p)
    if extractor is None:
        raise DxmiError("calculate_fid_given_paths: pass extractor= (an InceptionV3 pool3 feature extractor; this image cannot "
                        "download pytorch_fid's weights)")
    m1, s1 = compute_statistics_of_path(paths[0], extractor, batch_size, dims, device, resize)
    m2, s2 = compute_statistics_of_path(paths[1], extractor, batch_size, dims, device, resize)
    return calculate_frechet_distance(m1, s1, m2, s2)


def fid_from_images(images_u8, extractor, m2, s2, batch_size=50, dims=2048, device="cuda"):
    """The body of the reference's fid() (train_image_large.py:56-70) for this rank's uint8 images [n, 3, H, W]: scale to
    [0, 1], extract, gather over ranks, statistics on the device, distance on rank 0's host.  Every rank returns the value."""
    x = (images_u8.to(device) / 255.0).float()
    act = gather_activations(get_activations_from_tensor(x, extractor, batch_size=batch_size, dims=dims, device=device))
    m1, s1 = activation_statistics(act)
    return float(calculate_frechet_distance(m1, s1, np.asarray(m2), np.asarray(s2)))
