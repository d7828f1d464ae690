"""EDM preconditioning (`models.cm.karras_diffusion`) — the parts the DxMI few-step path uses.

Reference: models/cm/karras_diffusion.py — KarrasDenoiser.__init__/get_snr/get_scalings (:33-68),
denoise (:337-351), get_sigmas_karras (:423-429), get_ancestral_step (:437-444).  The consistency-
distillation losses and the 40-step Heun/ODE samplers of that file are outside SURVEY section 8.

denoise() keeps the reference signature for any callable `model`; OpenAIDiffusion does not call it on
the hot path — it uses the fused dxmi_edm_precond / dxmi_edm_step_fwd kernels, which evaluate the same
fp32 formulas per element.
"""
import torch

from .nn import append_dims, append_zero


class KarrasDenoiser:
    def __init__(self, sigma_data: float = 0.5, sigma_max=80.0, sigma_min=0.002, rho=7.0, weight_schedule="karras",
                 distillation=False, loss_norm="l2"):
        self.sigma_data, self.sigma_max, self.sigma_min = sigma_data, sigma_max, sigma_min
        self.weight_schedule, self.distillation, self.loss_norm, self.rho = weight_schedule, distillation, loss_norm, rho
        if loss_norm == "lpips":
            raise NotImplementedError("LPIPS loss belongs to consistency distillation, not to the DxMI path")

    def get_snr(self, sigmas):
        return sigmas ** -2

    def get_sigmas(self, sigmas):
        return sigmas

    def get_scalings(self, sigma):
        c_skip = self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2)
        c_out = sigma * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in

    def get_scalings_for_boundary_condition(self, sigma):
        c_skip = self.sigma_data ** 2 / ((sigma - self.sigma_min) ** 2 + self.sigma_data ** 2)
        c_out = (sigma - self.sigma_min) * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in

    def denoise(self, model, x_t, sigmas, **model_kwargs):
        scal = self.get_scalings_for_boundary_condition(sigmas) if self.distillation else self.get_scalings(sigmas)
        c_skip, c_out, c_in = [append_dims(s, x_t.ndim) for s in scal]
        rescaled_t = 1000 * 0.25 * torch.log(sigmas + 1e-44)
        model_output = model(c_in * x_t, rescaled_t, **model_kwargs)
        return model_output, c_out * model_output + c_skip * x_t


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas).to(device)


def get_ancestral_step(sigma_from, sigma_to):
    sigma_up = (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up
