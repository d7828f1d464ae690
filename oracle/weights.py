"""Formula-generated weights keyed by state-dict name (test infrastructure).

Full-size networks need no checkpoint files: every tensor is a deterministic function of its name
and shape through torch's CPU generator (identical in the build container and on the GPU box,
same torch build).  Scale keeps activations O(1): variance 1/fan_in for matrices/filters,
GroupNorm gains near 1, small biases.  zero_module-style layers get non-zero weights on purpose so
no work is skipped.
"""
import zlib

import torch


def formula_tensor(name, shape):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode("utf-8")))
    shape = tuple(shape)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return u * (3.0 / fan_in) ** 0.5
    if "norm" in name and name.endswith("weight"):
        return 1.0 + 0.2 * u
    return 0.1 * u


def formula_state_dict(reference_state_dict, skip=("log_betas", "std")):
    """Same keys/shapes as `reference_state_dict`; entries whose last name component is in `skip`
    (schedule-derived buffers/parameters) are copied through unchanged."""
    out = {}
    for k, v in reference_state_dict.items():
        if k.split(".")[-1] in skip or not torch.is_floating_point(v):
            out[k] = v.clone()
        else:
            out[k] = formula_tensor(k, v.shape).to(v.dtype)
    return out
