"""Oracle: VAR / EDM noise-schedule construction (TEST INFRASTRUCTURE; host-side float math).

Restates, in scalar numpy (float32 where the reference holds float32 torch tensors, float64
where it runs Python/NumPy-1 float64), the one-off table construction of
  models/DxMI/var_sampler.py:19-45   calc_diffusion_hyperparams
  models/DxMI/var_sampler.py:47-70   bisearch
  models/DxMI/var_sampler.py:73-97   get_VAR_noise
  models/DxMI/var_sampler.py:100-111 _log_gamma / _log_cont_noise
  models/DxMI/var_sampler.py:115-143 _precompute_VAR_steps
  models/DxMI/var_sampler.py:146-186 VAR_get_params
  models/DxMI/var_sampler.py:326-355 VARSampler.init_schedule (log_betas init)

NumPy-2 note (SURVEY 7 "hard parts"): the reference's _log_cont_noise receives float32 0-d arrays
and only works under NumPy-1 promotion (0-d float32 (+) python float -> float64).  This file
spells that promotion out explicitly, which reproduces the eta table printed in
models/DxMI/trainer.py:148-149.
"""
import numpy as np
import torch

BETA_0, BETA_T, T_DDPM = 0.0001, 0.02, 1000
f32 = np.float32


def ddpm_tables():
    """var_sampler.py:19-45 -> Beta, Alpha, Alpha_bar (float32, sequential products)."""
    # the one float32 op taken from torch itself: torch.linspace's two-ended evaluation is not
    # reproducible to the last ulp from numpy
    beta = torch.linspace(BETA_0, BETA_T, T_DDPM).numpy().copy()
    alpha = (f32(1) - beta).astype(np.float32)
    alpha_bar = alpha.copy()
    for t in range(1, T_DDPM):
        alpha_bar[t] = f32(alpha_bar[t] * alpha_bar[t - 1])
    return beta, alpha, alpha_bar


def bisearch(f, domain, target, eps=1e-8):
    """var_sampler.py:47-70 — bisection for a decreasing f."""
    sign = -1 if target < 0 else 1
    left, right = domain
    x = None
    for _ in range(1000):
        x = (left + right) / 2
        if f(x) < target:
            right = x
        elif f(x) > (1 + sign * eps) * target:
            left = x
        else:
            break
    return x


def var_noise(S, schedule="quadratic"):
    """var_sampler.py:73-97 — eta_i = beta_0 (1 + i x)^2 with prod(1-eta) = alpha_bar_1000 (float64)."""
    target = np.prod(1 - np.linspace(BETA_0, BETA_T, T_DDPM))
    if schedule == "quadratic":
        g = lambda x: np.array([BETA_0 * (1 + i * x) ** 2 for i in range(S)])
        domain = (0.0, 0.95 / np.sqrt(BETA_0) / S)
    elif schedule == "linear":
        g = lambda x: np.linspace(BETA_0, x, S)
        domain = (BETA_0, 0.99)
    else:
        raise NotImplementedError(schedule)
    f = lambda x: np.prod(1 - g(x))
    return g(bisearch(f, domain, target, eps=1e-4))


def _log_gamma(x):
    """var_sampler.py:100-103 — Stirling with the 1/(12x) term."""
    y = x - 1
    return np.log(2 * np.pi * y) / 2 + y * (np.log(y) - 1) + np.log(1 + 1 / (12 * y))


def log_cont_noise(t, beta_0_f32, beta_T_f32, T):
    """var_sampler.py:106-111 under NumPy-1 promotion: (bT-b0) in float32, everything after in float64."""
    delta_beta = np.float64(f32(f32(beta_T_f32) - f32(beta_0_f32))) / (T - 1)
    c = (1.0 - np.float64(f32(beta_0_f32))) / delta_beta
    t_1 = np.float64(t) + 1
    return t_1 * np.log(delta_beta) + _log_gamma(c + 1) - _log_gamma(c - t_1 + 1)


def gamma_bar_f32(eta):
    """cumulative product of (1 - eta) in float32 (var_sampler.py:122-126)."""
    g = (f32(1) - eta.astype(np.float32)).astype(np.float32)
    for t in range(1, len(g)):
        g[t] = f32(g[t] * g[t - 1])
    return g


def continuous_steps(eta):
    """var_sampler.py:115-143 — continuous DDPM time matching each Gamma_bar level, noisiest first."""
    beta, _, alpha_bar = ddpm_tables()
    gbar = gamma_bar_f32(eta)
    T = T_DDPM
    assert gbar[0] <= alpha_bar[0] and gbar[-1] >= alpha_bar[-1]
    steps = []
    for t in range(len(eta) - 1, -1, -1):
        t_adapted = None
        for i in range(T - 1):
            if alpha_bar[i] >= gbar[t] > alpha_bar[i + 1]:
                # np.log of a float32 0-d array stays float32 (var_sampler.py:138)
                target = np.log(f32(gbar[t]))
                t_adapted = bisearch(lambda _t: log_cont_noise(_t, beta[0], beta[-1], T),
                                     domain=(i - 0.01, i + 1.01), target=target)
                break
        if t_adapted is None:
            t_adapted = T - 1
        steps.append(t_adapted)
    return np.asarray(steps, dtype=np.float64)


def var_params(eta, cont_steps, kappa=1.0):
    """var_sampler.py:146-186 (and the per-t expressions of sample_step :367-376): float32 tables
    x_prev_multiplier, theta_multiplier, std; plus alpha_next per step."""
    gbar = gamma_bar_f32(eta)
    S = len(eta)
    xm = np.zeros(S, np.float32)
    cm = np.zeros(S, np.float32)
    std = np.zeros(S, np.float32)
    for i in range(S):
        gcur = gbar[S - 1 - i]
        if i == S - 1:
            assert abs(cont_steps[i]) < 0.1
            alpha_next, sigma = f32(1.0), f32(0.0)
        else:
            alpha_next = gbar[S - 1 - i - 1]
            sigma = f32(f32(kappa) * np.sqrt(f32(f32(f32(1) - alpha_next) / f32(f32(1) - gcur)) * f32(f32(1) - f32(gcur / alpha_next)), dtype=np.float32))
        ratio = f32(alpha_next / gcur)
        xm[i] = np.sqrt(ratio, dtype=np.float32)
        cm[i] = f32(np.sqrt(f32(f32(f32(1) - alpha_next) - f32(sigma * sigma)), dtype=np.float32)
                    - f32(np.sqrt(f32(f32(1) - gcur), dtype=np.float32) * np.sqrt(ratio, dtype=np.float32)))
        std[i] = f32(0.001) if i == S - 1 else sigma
    return xm, cm, std


def var_schedule(S):
    """Everything VARSampler.__init__ derives for S steps (var_sampler.py:301-355)."""
    eta = var_noise(S, "quadratic")
    cont = continuous_steps(eta)
    xm, cm, std = var_params(eta, cont)
    return {
        "user_defined_eta": eta,
        "continuous_steps": cont.astype(np.float32),  # torch.tensor(list of python floats) -> float32
        "Gamma_bar": gamma_bar_f32(eta),
        "x_prev_multiplier": xm,
        "theta_multiplier": cm,
        "std": std,
        "log_betas": np.log(std).astype(np.float32),
    }
