"""q-process helpers used by the trainer (reference: models/diffusion.py:5-22)."""
import torch


def make_beta_schedule(schedule="linear", n_timesteps=1000, start=1e-5, end=1e-2):
    """models/diffusion.py:5-15."""
    if schedule == "linear":
        return torch.linspace(start, end, n_timesteps)
    if schedule == "quad":
        return torch.linspace(start ** 0.5, end ** 0.5, n_timesteps) ** 2
    if schedule == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, n_timesteps)) * (end - start) + start
    if schedule == "constant":
        return torch.ones(n_timesteps) * start
    raise ValueError(f"unknown beta schedule {schedule!r}")


def extract(input, t, x):
    """Gather `input[t]` and reshape to broadcast over x (models/diffusion.py:18-22).
    Integer index path: the gather is exact."""
    out = torch.gather(input, 0, t.to(input.device))
    return out.reshape(t.shape[0], *([1] * (x.dim() - 1)))
