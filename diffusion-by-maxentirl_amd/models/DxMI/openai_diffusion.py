"""Few-step EDM sampler (`models.DxMI.openai_diffusion.OpenAIDiffusion`) on the gfx950 kernel library.

Plugin-compatible with the reference wrapper (reference: models/DxMI/openai_diffusion.py:10-129): same
constructor, attributes (.net, .diffusion, .sigmas, .sigma_down, .sigma_up, `net.log_betas` parameter or
buffer), same dictionaries from .sample_step() / .sample().

Host side: the Karras sigma ladder and its ancestral split are computed once on the CPU in fp32 with
the reference's expressions (tables are bit-identical to the golden fixture).
Device side, per step: ONE preconditioning kernel (c_in * x and the 250 ln(sigma) time input), the
U-Net, then ONE fused transition kernel (denoised, d, mean, x' = mean + sigma_up z) instead of the
~12 eager elementwise ops of :71-94.
Extension: `noise=` injects the Gaussian draws (x_T and z_1..z_T) so CPU and GPU runs share "seeds".
"""
import torch
import torch.nn as nn

from dxmi_hip import graph as _graph
from dxmi_hip import ops
from dxmi_hip._lib import DxmiError
from models.cm.karras_diffusion import get_sigmas_karras


class _EdmStepFn(torch.autograd.Function):
    """(x, model_output, z, sigma, sigma_down, sigma_up) -> (x', mu); gradients flow to model_output and sigma_up (x is the buffered
    state, z the draw, the sigma ladder a constant table)."""

    @staticmethod
    def forward(ctx, x, model_output, z, sigma, sigma_down, sigma_up, sigma_data):
        z, sg, sdn = z.contiguous().float(), sigma.contiguous().float(), sigma_down.contiguous().float()
        samples, mu = ops.edm_step(x.contiguous().float(), model_output.contiguous().float(), z, sg, sdn,
                                   sigma_up.detach().float().contiguous(), sigma_data)
        ctx.save_for_backward(z, sg, sdn)
        ctx.sigma_data = sigma_data
        return samples, mu

    @staticmethod
    def backward(ctx, g_sample, g_mu):
        z, sg, sdn = ctx.saved_tensors
        c = lambda g: None if g is None else g.contiguous().float()
        d_out, d_up = ops.edm_step_bwd(c(g_sample), c(g_mu), z, sg, sdn, ctx.sigma_data)
        return None, d_out, None, None, None, d_up, None


class OpenAIDiffusion:
    def __init__(self, model, diffusion, n_timesteps, sample_shape, class_cond=False, num_classes=0, trainable_beta=False,
                 sigma_min=0.002, sigma_max=80., stochastic_last=False, rho=7.0):
        self.net, self.diffusion = model, diffusion
        self.class_cond, self.num_classes = class_cond, num_classes
        self.sample_shape, self.n_timesteps, self.sigma_max = tuple(sample_shape), n_timesteps, sigma_max
        if stochastic_last:
            self.sigmas = get_sigmas_karras(n_timesteps + 1, sigma_min, sigma_max, rho=rho, device="cpu")[:-1]
        else:
            self.sigmas = get_sigmas_karras(n_timesteps, sigma_min, sigma_max, rho=rho, device="cpu")
        self.sigma_down, self.sigma_up = self.get_ancestral_step(self.sigmas)
        self.trainable_beta = trainable_beta
        if trainable_beta:
            self.net.register_parameter("log_betas", nn.Parameter(torch.log(self.sigma_up.clamp(1e-3))))
        else:
            self.net.register_buffer("log_betas", torch.log(self.sigma_up))
        self._dev_tabs = {}
        self.use_graph = False      # sample(): replay the T-step loop of a fixed (batch, destination) as one hipGraph
        self._graphs = {}

    def get_ancestral_step(self, sigmas):
        sigma_from, sigma_to = sigmas[:-1], sigmas[1:]
        sigma_up = (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        return sigma_down, sigma_up

    def train(self):
        self.net.train()

    def eval(self):
        self.net.eval()

    def parameters(self):
        return self.net.parameters()

    def _tabs(self, device):
        key = str(device)
        if key not in self._dev_tabs:
            self._dev_tabs[key] = tuple(t.to(device) for t in (self.sigmas, self.sigma_down, self.sigma_up))
        return self._dev_tabs[key]

    def _log_betas(self):
        net = self.net.module if hasattr(self.net, "module") else self.net
        return net.log_betas

    def sample_step(self, x, indices, noise=None, _outs=None, **model_kwargs):
        if not x.is_cuda:
            raise DxmiError("OpenAIDiffusion.sample_step runs only on the HIP device path (no CPU fallback)")
        sig_t, down_t, up_t = self._tabs(x.device)
        idx = indices.to(x.device)
        sigma, sigma_down, sigma_up = sig_t[idx], down_t[idx], up_t[idx]
        if self.trainable_beta:
            s = torch.exp(self._log_betas()[idx])
            if self.trainable_beta == "fix_last":
                terminal = idx == self.n_timesteps - 1
                s = s * ~terminal + sigma_up * terminal
            elif self.trainable_beta == "fix_last3":
                non_terminal = idx < self.n_timesteps - 3
                s = s * non_terminal + sigma_up * (~non_terminal)
            sigma_up = s
        x = x.contiguous().float()
        z = torch.randn_like(x) if noise is None else noise
        if torch.is_grad_enabled() and any(p.requires_grad for p in ops.fast_parameters(self.net)):
            return self._sample_step_grad(x, z, sigma, sigma_down, sigma_up, model_kwargs)
        x_in, rescaled_t = ops.edm_precond(x, sigma.contiguous(), self.diffusion.sigma_data)
        model_output = self.net(x_in, rescaled_t, **model_kwargs)
        samples, mu = ops.edm_step(x, model_output, z.contiguous(), sigma.contiguous(), sigma_down.contiguous(),
                                   sigma_up.detach().float().contiguous(), self.diffusion.sigma_data, outs=_outs)
        return {"sample": samples, "mean": mu, "sigma": sigma_up.clamp(1e-4, None)}

    def _sample_step_grad(self, x, z, sigma, sigma_down, sigma_up, model_kwargs):
        """Training path (policy update, trainer.py:693-746): the U-Net runs its HIP forward/backward through
        models/cm/unet_train.py; the transition around it is the fused forward kernels of the inference path (dxmi_edm_precond,
        dxmi_edm_step_fwd) and ONE backward kernel (dxmi_edm_step_bwd: the gradient of the network output and the per-sample
        sigma_up gradient, reduced on the device) — round 6; it was ~15 torch elementwise launches forward and as many backward.
        The loss reaches the network output and the learnable `log_betas` (through `sigma_up = exp(log_betas[idx])`, torch ops) exactly
        as in the reference (:71-94)."""
        x_in, rescaled_t = ops.edm_precond(x, sigma.contiguous(), self.diffusion.sigma_data)
        model_output = self.net(x_in, rescaled_t, **model_kwargs)
        samples, mu = _EdmStepFn.apply(x, model_output, z, sigma, sigma_down, sigma_up, self.diffusion.sigma_data)
        return {"sample": samples, "mean": mu, "sigma": sigma_up.clamp(1e-4, None)}

    def sample(self, n_sample, device, i_class=None, enable_grad=False, x0=None, noise=None, out=None):
        """noise: optional [T+1, n, C, H, W]; noise[0] * sigma_max is x_T, noise[1 + i] the draw of step i.
        The trajectory is ONE block [T+1, B, C, H, W] (+ [T, B, ...] mean, [T, B] sigma) written by the fused transition
        kernel; out: a slot of a models.DxMI.replay.TransitionRing to generate in place in the replay buffer."""
        device = torch.device(device)
        if self.use_graph and not enable_grad and x0 is None and noise is None and not isinstance(i_class, int) \
                and device.type == "cuda" and not _graph.capturing():
            # hipGraph replay (dxmi_hip/graph.py): one graph per (batch, destination, labels given or drawn); the first call of a key
            # runs eagerly, the second is captured.  Without `out=` the returned tensors are static (overwritten by the next call).
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            key = (n_sample, device.index, None if out is None else (id(out["ring"]), out["slot"]), i_class is not None)
            g = self._graphs.get(key)
            if g is None:
                from .trainer import _pack_modules
                fn = (lambda y: self._sample(n_sample, device, y, False, None, None, out)) if i_class is not None else \
                     (lambda: self._sample(n_sample, device, None, False, None, None, out))
                g = self._graphs[key] = _graph.StepGraph(fn, device, modules=_pack_modules(self), name=f"OpenAIDiffusion.sample{key}")
            return g(i_class) if i_class is not None else g()
        return self._sample(n_sample, device, i_class, enable_grad, x0, noise, out)

    def _sample(self, n_sample, device, i_class, enable_grad, x0, noise, out):
        if self.class_cond:
            if i_class is None:
                i_class = torch.randint(0, self.num_classes, (n_sample,), device=device)
            elif isinstance(i_class, int):
                i_class = torch.tensor([i_class] * n_sample, device=device, dtype=torch.long)
            model_kwargs = {"y": i_class}
        else:
            i_class, model_kwargs = None, {}
        T = self.n_timesteps
        size = (n_sample,) + tuple(self.sample_shape)
        in_place = not enable_grad
        if out is None:
            f32 = dict(dtype=torch.float32, device=device)
            traj, mean_b, sigma_b = torch.empty((T + 1,) + size, **f32), torch.empty((T,) + size, **f32), torch.empty((T, n_sample), **f32)
        else:
            assert in_place, "a ring slot cannot hold a differentiable trajectory"
            traj, mean_b, sigma_b = out["traj"], out["mean"], out["sigma"]
            assert traj.shape == (T + 1,) + size and traj.device == device, "ring slot does not match this sampler / batch"
            if i_class is not None and out["y"] is not None:
                out["y"].copy_(i_class)
        if x0 is not None:
            traj[0].copy_(x0)
        elif noise is not None:
            traj[0].copy_(noise[0])
            traj[0].mul_(self.sigma_max)
        else:
            traj[0].normal_().mul_(self.sigma_max)
        x = traj[0]
        l_x, l_mean, l_sigma = [x], [], []
        for i in range(T):
            with torch.set_grad_enabled(enable_grad):
                d_step = self.sample_step(x, torch.full((len(x),), i, dtype=torch.long, device=device),
                                          noise=None if noise is None else noise[1 + i].to(device),
                                          _outs=(traj[i + 1], mean_b[i]) if in_place else None, **model_kwargs)
            x = d_step["sample"]
            l_x.append(x)
            l_mean.append(d_step["mean"])
            if in_place:
                sigma_b[i].copy_(d_step["sigma"])
                l_sigma.append(sigma_b[i])
            else:
                l_sigma.append(d_step["sigma"])
        d = {"sample": l_x[-1], "l_sample": l_x, "y": i_class, "mean": l_mean, "sigma": l_sigma}
        if out is not None:
            d["_ring_slot"] = (id(out["ring"]), out["slot"])
        return d
