"""VAR few-step DDPM sampler (`models.DxMI.var_sampler.VARSampler`) on the gfx950 kernel library.

Plugin-compatible with the reference class (reference: models/DxMI/var_sampler.py:300-444): same
constructor, attributes (.net, .n_timesteps, .sample_shape, .trainable_beta, .user_defined_eta,
buffers continuous_steps / Gamma_bar / x_prev_multiplier / theta_multiplier / std /
diffusion_steps_list, `net.log_betas` parameter, `net.std` buffer), same return dictionaries from
.sample() and .sample_step().

Host side (one-off): the schedule tables, built with explicit float64 where the reference relied
on NumPy-1 scalar promotion (numpy>=2 breaks the reference there, README.md:29).
Device side: per step ONE fused HIP kernel does the whole transition (scale x, control, mean,
x' = mean + sigma z, Gaussian log-prob reduced over CHW) instead of ~10 eager ops + 5 clones
(reference :262-295); per-sample schedule scalars come from an integer-index gather kernel.
Extension: `noise=` injects the Gaussian draws (x_T, z_1..z_T) — the only way to define
"identical seeds" across CPU and GPU generators.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from dxmi_hip import graph as _graph
from dxmi_hip import ops
from dxmi_hip._lib import DxmiError
from ..modules import process_single_t

diffusion_config = {"beta_0": 0.0001, "beta_T": 0.02, "T": 1000}


# ----------------------------------------------------------------------------- host schedule (a1)
def calc_diffusion_hyperparams(T, beta_0, beta_T):
    """1000-step linear-beta DDPM tables in float32 (reference :19-45)."""
    beta = torch.linspace(beta_0, beta_T, T)
    alpha = 1 - beta
    alpha_bar = alpha.clone()
    beta_tilde = beta.clone()
    for t in range(1, T):  # sequential float32 products, as the reference accumulates them
        alpha_bar[t] *= alpha_bar[t - 1]
        beta_tilde[t] *= (1 - alpha_bar[t - 1]) / (1 - alpha_bar[t])
    return {"T": T, "Beta": beta, "Alpha": alpha, "Alpha_bar": alpha_bar, "Sigma": torch.sqrt(beta_tilde)}


def bisearch(f, domain, target, eps=1e-8):
    """Bisection on a decreasing f: the x where f(x) enters [target, (1+-eps) target] (reference :47-70)."""
    sign = -1 if target < 0 else 1
    lo, hi = domain
    x = None
    for _ in range(1000):
        x = (lo + hi) / 2
        fx = f(x)
        if fx < target:
            hi = x
        elif fx > (1 + sign * eps) * target:
            lo = x
        else:
            break
    return x


def get_VAR_noise(S, schedule="linear"):
    """eta_0..eta_{S-1} with prod(1 - eta) = alpha_bar_1000, float64 (reference :73-97)."""
    b0, bT, T = diffusion_config["beta_0"], diffusion_config["beta_T"], diffusion_config["T"]
    target = np.prod(1 - np.linspace(b0, bT, T))
    if schedule == "linear":
        g = lambda x: np.linspace(b0, x, S)
        domain = (b0, 0.99)
    elif schedule == "quadratic":
        g = lambda x: np.array([b0 * (1 + i * x) ** 2 for i in range(S)])
        domain = (0.0, 0.95 / np.sqrt(b0) / S)
    else:
        raise NotImplementedError(schedule)
    return g(bisearch(lambda x: np.prod(1 - g(x)), domain, target, eps=1e-4))


def _log_gamma(x):
    y = x - 1  # Stirling, reference :100-103
    return np.log(2 * np.pi * y) / 2 + y * (np.log(y) - 1) + np.log(1 + 1 / (12 * y))


def _log_cont_noise(t, beta_0, beta_T, T):
    """log alpha_bar at continuous time t (reference :106-111), in explicit float64: beta_0/beta_T
    arrive as float32 scalars; their difference is formed in float32 (as NumPy-1 did), the rest
    in float64."""
    b0, bT = np.float32(beta_0), np.float32(beta_T)
    delta_beta = np.float64(np.float32(bT - b0)) / (T - 1)
    c = (1.0 - np.float64(b0)) / delta_beta
    t_1 = np.float64(t) + 1
    return t_1 * np.log(delta_beta) + _log_gamma(c + 1) - _log_gamma(c - t_1 + 1)


def _gamma_bar(user_defined_eta):
    g = 1 - torch.from_numpy(np.asarray(user_defined_eta)).to(torch.float32)
    for t in range(1, len(g)):
        g[t] *= g[t - 1]
    return g


def _precompute_VAR_steps(diffusion_hyperparams, user_defined_eta, device=None):
    """Continuous DDPM time of each user step, noisiest first (reference :115-143)."""
    T, alpha_bar, beta = diffusion_hyperparams["T"], diffusion_hyperparams["Alpha_bar"], diffusion_hyperparams["Beta"]
    gbar = _gamma_bar(user_defined_eta)
    assert gbar[0] <= alpha_bar[0] and gbar[-1] >= alpha_bar[-1]
    abar = alpha_bar.numpy()
    b0, bT = beta[0].numpy(), beta[-1].numpy()
    steps = []
    for t in range(len(gbar) - 1, -1, -1):
        level = gbar[t].numpy()
        t_adapted = None
        # first i with abar[i] >= level > abar[i+1]; abar is decreasing
        hits = np.nonzero((abar[:-1] >= level) & (level > abar[1:]))[0]
        if len(hits):
            i = int(hits[0])
            t_adapted = bisearch(lambda _t: _log_cont_noise(_t, b0, bT, T), (i - 0.01, i + 1.01), np.log(level))
        steps.append(T - 1 if t_adapted is None else t_adapted)
    return steps


def _step_tables(gbar, kappa):
    """Per-integer-t multipliers (x multiplier, theta multiplier, sigma) in float32, vectorised over
    t = 0..S-1 with the reference's expressions (:146-186 and :367-376; the t = S-1 row takes
    alpha_next = 1, sigma = 0)."""
    # Scalar float32 arithmetic, one IEEE operation at a time: torch's VECTORISED float32 kernels
    # round sqrt/div differently from its scalar path on some hosts (1 ulp, seen on the GPU box's
    # CPU), and the reference evaluates these per element on 0-d tensors.
    f = np.float32
    g = gbar.numpy()
    S = len(g)
    x_mult, c, sigma = (np.zeros(S, np.float32) for _ in range(3))
    one = f(1)
    for t in range(S):
        gcur = g[S - 1 - t]
        if t == S - 1:
            alpha_next, sig = one, f(0)
        else:
            alpha_next = g[S - 2 - t]
            sig = f(f(kappa) * np.sqrt(f(f(f(one - alpha_next) / f(one - gcur)) * f(one - f(gcur / alpha_next)))))
        ratio = np.sqrt(f(alpha_next / gcur))
        x_mult[t] = ratio
        c[t] = f(np.sqrt(f(f(one - alpha_next) - f(sig * sig))) - f(np.sqrt(f(one - gcur)) * ratio))
        sigma[t] = sig
    return torch.from_numpy(x_mult), torch.from_numpy(c), torch.from_numpy(sigma)


def VAR_get_params(diffusion_hyperparams, user_defined_eta, kappa, continuous_steps):
    """x_prev_multiplier, theta_multiplier, std, diffusion_steps_list (reference :146-186)."""
    assert 0.0 <= kappa <= 1.0
    gbar = _gamma_bar(user_defined_eta)
    assert abs(float(continuous_steps[-1])) < 0.1
    x_mult, c, sigma = _step_tables(gbar, kappa)
    std = sigma.clone()
    std[-1] = 0.001
    return x_mult, c, std, torch.as_tensor(continuous_steps, dtype=torch.float32).clone()


# ----------------------------------------------------------------------------- sampler plugin
class VARSampler(nn.Module):
    def __init__(self, net, n_timesteps, sample_shape, trainable_beta=True, adhoc_scale1=1.0, adhoc_scale2=1.0):
        super().__init__()
        assert trainable_beta in {True, False, "fix_last"}
        self.net = net
        self.n_timesteps = n_timesteps
        self.temb_table = True      # sample(): evaluate the U-Net's timestep branch once per call on the T distinct timesteps
        self.sample_shape = sample_shape
        self.adhoc_scale1, self.adhoc_scale2 = adhoc_scale1, adhoc_scale2
        self.trainable_beta = trainable_beta
        self.init_schedule()
        x_mult, c, std, dsl = VAR_get_params(self.diffusion_hyperparams, self.user_defined_eta, self.kappa, self.continuous_steps)
        self.register_buffer("x_prev_multiplier", x_mult)
        self.register_buffer("theta_multiplier", c)
        self.register_buffer("std", std)
        self.register_buffer("diffusion_steps_list", dsl)
        if self.trainable_beta == "fix_last":
            self.net.register_buffer("std", std)
        self._tcache = {}
        self.use_graph = False      # sample(): replay the T-step loop of a fixed (batch, device, destination) as one hipGraph
        self._graphs = {}

    def init_schedule(self):
        """reference :326-355."""
        self.diffusion_hyperparams = calc_diffusion_hyperparams(**diffusion_config)
        self.kappa = 1.0
        self.user_defined_eta = get_VAR_noise(self.n_timesteps, "quadratic")
        self.register_buffer("continuous_steps", torch.tensor(_precompute_VAR_steps(self.diffusion_hyperparams, self.user_defined_eta)))
        gbar = _gamma_bar(self.user_defined_eta)
        abar = self.diffusion_hyperparams["Alpha_bar"]
        assert gbar[0] <= abar[0] and gbar[-1] >= abar[-1]
        self.register_buffer("Gamma_bar", gbar)
        _, _, sigma = _step_tables(gbar, self.kappa)
        sigma[-1] = 0.001
        if self.trainable_beta:
            self.net.log_betas = nn.Parameter(torch.log(sigma * self.adhoc_scale2))  # log sigma, despite the name

    # ---- helpers
    def _bare_net(self):
        return self.net.module if hasattr(self.net, "module") else self.net

    def _log_betas_all(self):
        """log sigma per step (reference :268-280, :383-394); None when sigma is not learnable."""
        net = self._bare_net()
        if self.trainable_beta == "fix_last":
            return torch.cat([net.log_betas[:-1], net.std[-1].log().unsqueeze(0)])
        if self.trainable_beta:
            return net.log_betas
        return torch.log(self.std)  # fixed schedule: sigma_t, last step 1e-3 (:282-283, :396)

    def _t_const(self, B, i, device):
        key = (B, str(device))
        tab = self._tcache.get(key)
        if tab is None:
            tab = torch.arange(self.n_timesteps, device=device)[:, None].repeat(1, B).contiguous()
            self._tcache = {key: tab}
        return tab[i]

    def _check_t(self, t):
        """The reference's torch indexing raises on t outside [-T, T).  Host-resident t (python int, CPU tensor) is
        checked here; for device tensors a check would be a host sync per transition, so the gather kernel poisons
        out-of-range samples with NaN instead of reading out of bounds."""
        if torch.is_tensor(t):
            if t.is_cuda or t.numel() == 0:
                return
            lo, hi = int(t.min()), int(t.max())
        else:
            lo = hi = int(t)
        if hi >= self.n_timesteps or lo < -self.n_timesteps:
            raise IndexError(f"timestep out of range for a {self.n_timesteps}-step sampler: [{lo}, {hi}]")

    def _transition(self, x, t, z, assoc, outs=None, sigma_out=None, temb_rows=None, log_betas=None):
        """One fused transition for integer timesteps t [B] (device int64).  outs = (x_next, mean, control, logp) and
        sigma_out: preallocated destinations (rows of the trajectory block / replay ring).  temb_rows: this step's row of the
        U-Net's timestep-branch table when the whole batch shares t (sample())."""
        if log_betas is None:
            log_betas = self._log_betas_all().detach().float().contiguous()
        tau, xm, cm, sg = ops.var_gather_sched(t, self.continuous_steps, self.x_prev_multiplier, self.theta_multiplier,
                                               log_betas, sigma_out=sigma_out)
        eps = self.net(x, tau) if temb_rows is None else self.net(x, tau, temb_rows=temb_rows)
        if self.adhoc_scale1 != 1.0:
            cm = cm * self.adhoc_scale1
        x_next, mean, control, logp = ops.var_step(x, eps, z, xm, cm, sg, assoc=assoc, outs=outs)
        return x_next, mean, control, logp, sg

    # ---- plugin API
    def sample_step(self, x, t, y=None, noise=None):
        """One transition with per-sample integer t (reference :357-408)."""
        if not x.is_cuda:
            raise DxmiError("VARSampler.sample_step runs only on the HIP device path")
        self._check_t(t)
        t = process_single_t(x, t)
        if torch.is_grad_enabled() and any(p.requires_grad for p in ops.fast_parameters(self.net)):
            from .var_sampler_train import sample_step_with_grad
            return sample_step_with_grad(self, x, t, noise)
        z = torch.randn_like(x) if noise is None else noise
        with torch.no_grad():
            xn, mean, control, logp, sg = self._transition(x.contiguous().float(), t, z.contiguous(), assoc=0)
        sigma = sg.view(-1, 1, 1, 1)
        return {"sample": xn, "logp": logp, "logp_terminal": torch.zeros(len(x), device=x.device), "mean": mean,
                "sigma": sigma, "entropy": torch.log(sigma), "control": control}

    def sample(self, n_sample, device="cpu", enable_grad=False, noise=None, out=None):
        """T-step generation (reference :411-428 -> VAR_sampling :204-297).  The whole trajectory lives in ONE block
        [T+1, B, C, H, W] (+ [T, B, ...] mean / control, [T, B] logp / sigma) that the fused transition kernel writes
        directly; the returned lists are views of it.  out: a slot of a models.DxMI.replay.TransitionRing — the
        trajectory is then generated in place in the replay buffer."""
        device = torch.device(device)
        if device.type != "cuda":
            raise DxmiError("VARSampler.sample runs only on the HIP device path (device must be cuda:N)")
        if enable_grad:
            raise NotImplementedError("enable_grad=True (fresh_sample_grad) is not used by the DxMI configs")
        if self.use_graph and noise is None and not _graph.capturing():
            # hipGraph replay (dxmi_hip/graph.py): one graph per (batch, destination); the first call of a key runs eagerly, the
            # second is captured.  Without `out=` the returned tensors are static: the next call with the same key overwrites them.
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            key = (n_sample, device.index, None if out is None else (id(out["ring"]), out["slot"]))
            g = self._graphs.get(key)
            if g is None:
                g = self._graphs[key] = _graph.StepGraph(lambda: self._sample(n_sample, device, None, out), device,
                                                         modules=[self._bare_net()], name=f"VARSampler.sample{key}")
            return g()
        return self._sample(n_sample, device, noise, out)

    def _sample(self, n_sample, device, noise, out):
        shape = tuple(self.sample_shape)
        size = (n_sample,) + shape
        T = self.n_timesteps
        if out is None:
            f32 = dict(dtype=torch.float32, device=device)
            traj = torch.empty((T + 1,) + size, **f32)
            mean_b, control_b = torch.empty((T,) + size, **f32), torch.empty((T,) + size, **f32)
            logp_b, sigma_b = torch.empty((T, n_sample), **f32), torch.empty((T, n_sample), **f32)
        else:
            traj, mean_b, control_b, logp_b, sigma_b = (out[k] for k in ("traj", "mean", "control", "logp", "sigma"))
            assert traj.shape == (T + 1,) + size and traj.device == device, "ring slot does not match this sampler / batch"
        if noise is None:
            traj[0].normal_()
        else:
            traj[0].copy_(noise[0])
        with torch.no_grad():
            # every image of a generation step shares its timestep: the U-Net's timestep branch (embedding, two dense layers,
            # all temb_proj) is evaluated once per sample() call on the T distinct timesteps instead of on T x B identical rows
            table = None
            if self.temb_table and hasattr(self._bare_net(), "temb_table"):
                table = self._bare_net().temb_table(self.continuous_steps[:T].to(device).float().contiguous())
            lb = self._log_betas_all().detach().float().contiguous()     # the sigma table does not change inside a sample() call
            for i in range(T):
                z = torch.randn(size, device=device) if noise is None else noise[i + 1].to(device).float().contiguous()
                self._transition(traj[i], self._t_const(n_sample, i, device), z, assoc=1,
                                 outs=(traj[i + 1], mean_b[i], control_b[i], logp_b[i]), sigma_out=sigma_b[i],
                                 temb_rows=None if table is None else table[i:i + 1], log_betas=lb)
        d = {"sample": traj[T], "l_sample": list(traj.unbind(0)), "logp": list(logp_b.unbind(0)),
             "logp_terminal": torch.zeros(n_sample, device=device), "mean": list(mean_b.unbind(0)),
             "sigma": [s_.view(-1, 1, 1, 1) for s_ in sigma_b.unbind(0)], "control": list(control_b.unbind(0))}
        if out is not None:
            d["_ring_slot"] = (id(out["ring"]), out["slot"])
        return d

    def log_prob_step(self, x_prev, x_next, t):
        """log-prob of a stored transition under the fixed-sigma policy (reference :431-444 -> VAR_log_prob :189-200).
        As in the reference the network call is NOT detached: under autograd the result back-propagates into the
        U-Net parameters and x_prev through the HIP backward (unet_small_train.py); x_next is detached."""
        if not x_prev.is_cuda:
            raise DxmiError("VARSampler.log_prob_step runs only on the HIP device path")
        self._check_t(t)
        t = process_single_t(x_prev, t)
        if t.numel() and (int(t.max()) >= self.n_timesteps or int(t.min()) < -self.n_timesteps):   # torch gather below: no silent wrap
            raise IndexError("timestep out of range")
        eps = self.net(x_prev.contiguous().float(), self.diffusion_steps_list[t])
        mean = x_prev * self.x_prev_multiplier[t][:, None, None, None] + self.theta_multiplier[t][:, None, None, None] * eps
        sigma = self.std[t][:, None, None, None]
        lp = -((x_next.detach() - mean) ** 2) / (2 * sigma ** 2) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))
        return lp.mean(dim=-1).mean(dim=-1).mean(dim=-1)
