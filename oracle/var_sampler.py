"""Oracle: VAR few-step sampler transitions (TEST INFRASTRUCTURE).

Restates models/DxMI/var_sampler.py: VAR_sampling :204-297 (T-step loop), VARSampler.sample_step
:357-408 (per-sample integer t), VAR_log_prob :189-200.  The network is passed as a callable
net_fn(x, t_float) -> eps so any backbone (oracle U-Net, HIP U-Net, stub) can be driven.
Noise is INJECTED (x_T and one z per step) — CPU and GPU generators differ, so "identical seeds"
is only definable through the noise tensors themselves (RNG parity).
"""
import math

import torch


def gaussian_logp_mean(x, mean, sigma):
    """Normal(mean, sigma).log_prob(x) averaged over (C,H,W); torch.distributions.Normal
    formula, reductions in the reference's order (var_sampler.py:288-289)."""
    var = sigma ** 2
    lp = -((x - mean) ** 2) / (2 * var) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))
    return lp.mean(dim=-1).mean(dim=-1).mean(dim=-1)


def log_betas_all(log_betas, std, trainable_beta):
    """var_sampler.py:268-280 / :383-394 — which log-sigma each step uses."""
    if trainable_beta == "fix_last":
        return torch.cat([log_betas[:-1], std[-1].log().unsqueeze(0)])
    if trainable_beta:
        return log_betas
    return None


def sample(net_fn, sched, log_betas, noise, trainable_beta="fix_last", adhoc_scale1=1.0):
    """VAR_sampling, var_sampler.py:204-297.
    sched: dict of float32 torch tensors (continuous_steps, Gamma_bar, std); noise: [x_T, z_0..z_{T-1}].
    Returns the dict VARSampler.sample builds (:418-428)."""
    cont, gbar, std = sched["continuous_steps"], sched["Gamma_bar"], sched["std"]
    S = len(cont)
    x = noise[0].clone()
    B = x.shape[0]
    lba = log_betas_all(log_betas, std, trainable_beta)
    x_seq, logps, controls, means, sigmas = [x.clone()], [], [], [], []
    for i in range(S):
        tau = cont[i]
        eps = net_fn(x, tau * torch.ones(B))
        if i == S - 1:
            assert abs(float(tau)) < 0.1
            alpha_next, sigma = torch.tensor(1.0), torch.tensor(0.0)
        else:
            alpha_next = gbar[S - 1 - i - 1]
            sigma = 1.0 * torch.sqrt((1 - alpha_next) / (1 - gbar[S - 1 - i]) * (1 - gbar[S - 1 - i] / alpha_next))
        x = x * torch.sqrt(alpha_next / gbar[S - 1 - i])
        c = torch.sqrt(1 - alpha_next - sigma ** 2) - torch.sqrt(1 - gbar[S - 1 - i]) * torch.sqrt(alpha_next / gbar[S - 1 - i])
        control = c * eps * adhoc_scale1
        mean = x + control
        if lba is not None:
            sigma = torch.exp(lba[i])
        elif i == S - 1:
            sigma = torch.tensor(0.001)
        x = x + (control + sigma * noise[i + 1])
        pred_std = sigma.repeat(B)[:, None, None, None]
        logps.append(gaussian_logp_mean(x, mean, pred_std))
        x_seq.append(x.clone())
        controls.append(control.clone())
        means.append(mean.clone())
        sigmas.append(pred_std.clone())
    return {"sample": x_seq[-1], "l_sample": x_seq, "logp": logps, "logp_terminal": torch.zeros(B),
            "mean": means, "sigma": sigmas, "control": controls}


def step_tables(sched):
    """Per-integer-t multipliers of sample_step (var_sampler.py:363-376), vectorised over t=0..S-1
    in float32 exactly as the reference evaluates them per sample."""
    gbar = sched["Gamma_bar"]
    S = len(gbar)
    t = torch.arange(S)
    is_last = t == S - 1
    alpha_next = gbar[S - 1 - t - 1]          # index -1 wraps for t = S-1, masked below (:367-368)
    alpha_next = alpha_next * (~is_last) + is_last * 1.0
    gcur = gbar[S - 1 - t]
    sigma = 1.0 * torch.sqrt((1 - alpha_next) / (1 - gcur) * (1 - gcur / alpha_next))
    sigma = sigma * (~is_last) + is_last * 0
    x_mult = torch.sqrt(alpha_next / gcur)
    c = torch.sqrt(1 - alpha_next - sigma ** 2) - torch.sqrt(1 - gcur) * torch.sqrt(alpha_next / gcur)
    return x_mult, c, sigma


def sample_step(net_fn, sched, log_betas, x, t, z, trainable_beta="fix_last", adhoc_scale1=1.0):
    """VARSampler.sample_step, var_sampler.py:357-408; t int64 [B]; z = the randn_like draw."""
    S = len(sched["continuous_steps"])
    if isinstance(t, int) or t.dim() == 0 or len(t) == 1:
        t = torch.ones(x.shape[0], dtype=torch.long) * t          # models/modules.py:183-186
    tau = sched["continuous_steps"][t]
    is_last = t == S - 1
    eps = net_fn(x, tau)
    x_mult_tab, c_tab, sig_tab = step_tables(sched)
    xs = x * x_mult_tab[t][:, None, None, None]
    control = c_tab[t][:, None, None, None] * eps * adhoc_scale1
    mean = xs + control
    lba = log_betas_all(log_betas, sched["std"], trainable_beta)
    if lba is not None:
        sigma = torch.exp(lba[t])
    else:
        sigma = sig_tab[t] * (~is_last) + is_last * 0.001
    sigma = sigma[:, None, None, None]
    xn = mean + sigma * z
    return {"sample": xn, "logp": gaussian_logp_mean(xn.detach(), mean, sigma),
            "logp_terminal": torch.zeros(len(x)), "mean": mean, "sigma": sigma,
            "entropy": torch.log(sigma), "control": control}


def log_prob_step(net_fn, sched, x_prev, x_next, t):
    """VARSampler.log_prob_step -> VAR_log_prob, var_sampler.py:431-444, :189-200: log N(x_next; mean_t(x_prev), std_t)
    averaged over (C,H,W) under the FIXED schedule (std table with its last entry 1e-3, tau from diffusion_steps_list =
    continuous_steps); differentiable through net_fn, x_next detached."""
    x_mult_tab, c_tab = sched["x_prev_multiplier"], sched["theta_multiplier"]     # VAR_get_params tables (:146-186)
    eps = net_fn(x_prev, sched["continuous_steps"][t])
    mean = x_prev * x_mult_tab[t][:, None, None, None] + c_tab[t][:, None, None, None] * eps
    return gaussian_logp_mean(x_next.detach(), mean, sched["std"][t][:, None, None, None])
