"""Differentiable VAR transition (`VARSampler.sample_step` under autograd; reference
models/DxMI/var_sampler.py:357-408).  The U-Net runs through its HIP autograd function; the per-sample
schedule scalars are integer gathers; the remaining elementwise algebra on the [B,3,32,32] fp32 state
(12 KB per sample) is left to torch autograd so `log_betas` receives its gradient exactly as in the
reference (sigma = exp(log_betas_all[t]))."""
import math

import torch


def sample_step_with_grad(sampler, x, t, noise=None):
    T = sampler.n_timesteps
    tau = sampler.continuous_steps[t]
    xm = sampler.x_prev_multiplier[t][:, None, None, None]
    cm = sampler.theta_multiplier[t][:, None, None, None]
    eps = sampler.net(x.contiguous().float(), tau)
    xs = x * xm
    control = cm * eps * sampler.adhoc_scale1
    mean = xs + control
    lba = sampler._log_betas_all()          # differentiable w.r.t. net.log_betas (fix_last pins the last entry)
    sigma = torch.exp(lba[t])[:, None, None, None]
    z = torch.randn_like(x) if noise is None else noise
    xn = mean + sigma * z
    lp = -((xn.detach() - mean) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))
    logp = lp.mean(-1).mean(-1).mean(-1)
    return {"sample": xn, "logp": logp, "logp_terminal": torch.zeros(len(x), device=x.device), "mean": mean,
            "sigma": sigma, "entropy": torch.log(sigma), "control": control}
