"""Small helpers of `models.cm.nn` that the DxMI scripts import (reference: models/cm/nn.py).

Layer factories return torch.nn modules used as PARAMETER CONTAINERS by models.cm.unet.UNetModel; the
compute runs through dxmi_hip.  timestep_embedding is the device kernel (order 1 = [cos | sin] with
freq = exp(-ln(P) i / half), reference :119-137).
"""
import torch
import torch.nn as nn

from dxmi_hip import ops


class GroupNorm32(nn.GroupNorm):
    """32-group norm container (reference :18-20); statistics are always fp32 in the HIP kernels."""


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the DxMI image models are 2-D)")
    return nn.Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def normalization(channels):
    return GroupNorm32(32, channels)


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def update_ema(target_params, source_params, rate=0.99):
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)


def timestep_embedding(timesteps, dim, max_period=10000):
    return ops.timestep_embedding(timesteps, dim, order=1, max_period=float(max_period))
