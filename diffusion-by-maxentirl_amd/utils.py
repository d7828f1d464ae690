"""Small helpers of the CLI surface (reference utils.py:21, :140, :251-273)."""
import errno
import os
from collections import OrderedDict

import torch


def get_rank():
    return torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0


def print0(*args, **kwargs):
    if get_rank() == 0:
        print(*args, **kwargs)


def mkdir_p(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if not (e.errno == errno.EEXIST and os.path.isdir(path)):
            raise


def remove_module(d):
    return OrderedDict((k[len("module."):], v) for k, v in d.items())


def fix_legacy_dict(d):
    """Unwrap {'model': ...} / {'state_dict': ...} checkpoints and strip a DDP 'module.' prefix
    (reference utils.py:263-273) — needed to load the pretrained DDPM checkpoint."""
    keys = list(d.keys())
    if "model" in keys:
        d = d["model"]
    if "state_dict" in keys:
        d = d["state_dict"]
    keys = list(d.keys())
    if len(keys) > 1 and "module." in keys[1]:
        d = remove_module(d)
    return d


def weight_norm(net):
    """sqrt(sum ||p||^2) over parameters (reference utils.py:140-147)."""
    total = 0.0
    for p in net.parameters():
        total += p.detach().float().pow(2).sum().item()
    return total ** 0.5
