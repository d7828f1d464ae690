"""Differentiable VAR transition (`VARSampler.sample_step` under autograd; reference models/DxMI/var_sampler.py:357-408).
The U-Net runs through its HIP autograd function; the per-sample schedule scalars are integer gathers; the transition itself is
ONE fused kernel forward (dxmi_var_step_fwd: x' = mean + sigma z, mean, control, the Gaussian log-prob reduced over CHW) and ONE
backward (dxmi_var_step_bwd: the gradient of the network output and the per-sample sigma gradient, reduced on the device) — round
6; it was ~12 torch elementwise launches forward and ~20 backward.  `sigma = exp(log_betas_all[t])` stays two torch ops so that
`log_betas` receives its gradient through torch's own index / exp backward, exactly as in the reference ('fix_last' pins the last
entry through `_log_betas_all`)."""
import torch

from dxmi_hip import ops


class _VarStepFn(torch.autograd.Function):
    """(x, eps, z, xmul [B], cmul [B], sigma [B]) -> (x', mean, control, logp); gradients flow to eps and sigma (x: the buffered state,
    z: the draw — neither asks for one in the trainers; a gradient w.r.t. x is returned as xmul (g_x' + g_mean) when it does)."""

    @staticmethod
    def forward(ctx, x, eps, z, xm, cm, sigma):
        x, eps, z = x.contiguous().float(), eps.contiguous().float(), z.contiguous().float()
        xm, cm, sg = xm.contiguous().float(), cm.contiguous().float(), sigma.detach().contiguous().float()
        xn, mean, control, logp = ops.var_step(x, eps, z, xm, cm, sg, assoc=0)
        ctx.save_for_backward(z, xm, cm, sg)
        ctx.need_dx = x.requires_grad
        return xn, mean, control, logp

    @staticmethod
    def backward(ctx, g_next, g_mean, g_control, g_logp):
        z, xm, cm, sg = ctx.saved_tensors
        c = lambda g: None if g is None else g.contiguous().float()
        g_next, g_mean, g_control, g_logp = c(g_next), c(g_mean), c(g_control), c(g_logp)
        d_eps, d_sigma = ops.var_step_bwd(g_next, g_mean, g_control, g_logp, z, cm, sg)
        dx = None
        if ctx.need_dx:
            gsum = sum(g for g in (g_next, g_mean) if g is not None)
            dx = gsum * xm.view(-1, *([1] * (z.dim() - 1))) if torch.is_tensor(gsum) else None
        return dx, d_eps, None, None, None, d_sigma


def sample_step_with_grad(sampler, x, t, noise=None):
    tau = sampler.continuous_steps[t]
    xm = sampler.x_prev_multiplier[t]
    cm = sampler.theta_multiplier[t]
    if sampler.adhoc_scale1 != 1.0:
        cm = cm * sampler.adhoc_scale1
    x = x.contiguous().float()
    eps = sampler.net(x, tau)
    lba = sampler._log_betas_all()          # differentiable w.r.t. net.log_betas (fix_last pins the last entry)
    sigma_b = torch.exp(lba[t])             # [B]
    z = torch.randn_like(x) if noise is None else noise
    xn, mean, control, logp = _VarStepFn.apply(x, eps, z, xm, cm, sigma_b)
    sigma = sigma_b[:, None, None, None]
    return {"sample": xn, "logp": logp, "logp_terminal": torch.zeros(len(x), device=x.device), "mean": mean,
            "sigma": sigma, "entropy": torch.log(sigma), "control": control}
