"""Oracle: one DxMI training step on an EDM backbone, on the CPU (TEST INFRASTRUCTURE).

Restates models/DxMI/trainer.py:412-746 (DxMI_Trainer_Cond: set_models :496-525, get_running_cost :527-534 with
beta_ordering 'forward', update_adaptive_vel_reg :536-551, update_f_v :553-691, update_sampler_mixed_precision :693-746)
with models/cm/fp16_util.py:161-248 (MixedPrecisionTrainer: loss scaled by 2**lg_loss_scale, un-scaled before the step,
scale grown by 1e-3 per successful step; the flat master tensors hold the same numbers as the per-tensor parameters
used here, and RAdam is element-wise, so the update is the same) over the oracle's functional networks with torch
autograd, keeping the reference's literal index expressions.
"""
import torch
import torch.nn.functional as F

from . import edm
from . import value as ovalue
from .precision import Precision


class OracleDxMICond:
    def __init__(self, net_sd, value_sd, cfg, sch, B, T, prec=None, tau1=0.1, tau2=0.01, gamma=1.0, adavelreg=0.99, time_cost=0,
                 time_cost_sig=1.0, skip_sampler_tau=0, value_grad_clip=True, lr=1e-6, beta_lr=1e-4, v_lr=1e-5, lg_loss_scale=20.0):
        self.prec = prec or Precision("fp32")
        self.cfg, self.sch, self.B, self.T = cfg, sch, B, T
        self.net = {k: v.clone().requires_grad_(True) for k, v in net_sd.items()}
        self.val = {k: v.clone().requires_grad_(True) for k, v in value_sd.items()}
        self.tau1, self.tau2, self.gamma, self.adavelreg = tau1, tau2, gamma, adavelreg
        self.time_cost, self.time_cost_sig, self.skip_sampler_tau, self.value_grad_clip = time_cost, time_cost_sig, skip_sampler_tau, value_grad_clip
        not_beta = [v for k, v in self.net.items() if k != "log_betas"]
        self.opt = torch.optim.RAdam([{"params": not_beta, "lr": lr}, {"params": [self.net["log_betas"]], "lr": beta_lr}], weight_decay=0.0)
        self.opt_v = torch.optim.Adam(list(self.val.values()), lr=v_lr)
        self.betas_for_q = sch.sigmas[:-1] ** 2          # :516-517
        self.lg_loss_scale = lg_loss_scale

    def unet(self, x, t, **kw):
        return edm.unet_forward(self.net, self.cfg, x, t, prec=self.prec, **kw)

    def v(self, x):
        return ovalue.forward(self.val, x, self.prec)

    def sample(self, x0, noises, y):
        with torch.no_grad():
            d = edm.sample(self.unet, self.sch, x0, noises, log_betas=self.net["log_betas"].detach(), y=y)
        d["y"] = y
        return d

    @staticmethod
    def reset_buffer():
        b = {k: torch.FloatTensor() for k in ("state", "next_state", "mean", "sigma")}
        b["timestep"], b["y"] = torch.LongTensor(), torch.LongTensor()
        return b

    @staticmethod
    def append_buffer(buf, d):      # trainer.py:23-55
        x_seq = d["l_sample"]
        n, T = len(x_seq[0]), len(x_seq) - 1
        for t in range(T):
            buf["state"] = torch.cat((buf["state"], x_seq[t].detach()))
            buf["next_state"] = torch.cat((buf["next_state"], x_seq[t + 1].detach()))
            buf["timestep"] = torch.cat((buf["timestep"], torch.tensor([t] * n)))
            buf["mean"] = torch.cat((buf["mean"], d["mean"][t].detach()))
            buf["sigma"] = torch.cat((buf["sigma"], d["sigma"][t].detach()))
            buf["y"] = torch.cat((buf["y"], d["y"].detach()))
        return buf

    def running_cost(self, state, next_state, t):
        beta = torch.gather(self.betas_for_q, 0, t).reshape(len(t), 1, 1, 1)      # forward ordering, diffusion.py:18-22
        return (((next_state - state) ** 2) / (2 * beta)).view(len(state), -1).mean(dim=1)

    def update_f_v(self, img, d, buf):
        samples = torch.stack(d["l_sample"])
        diff = ((samples[1:] - samples[:-1]) ** 2).view(samples.shape[0] - 1, -1).mean(dim=1)
        self.betas_for_q = (self.betas_for_q * self.adavelreg + (1 - self.adavelreg) * diff).detach()
        self.opt_v.zero_grad()
        x0 = d["l_sample"][-1]
        out = self.v(torch.cat((img.detach(), x0.detach()), 0))
        pos_e, neg_e = out[:x0.shape[0]], out[x0.shape[0]:]
        reg = pos_e.pow(2).mean() + neg_e.pow(2).mean()
        d_loss = pos_e.mean() - neg_e.mean() + self.gamma * reg
        d_loss.backward()
        self.opt_v.step()
        self.opt_v.zero_grad()
        T, B = self.T, self.B
        permutation = torch.randperm(B * T)
        indices = permutation + (buf["state"].shape[0] - B * T)
        d_rc, d_val = {}, {}
        for i in range(T):
            update_t = T - i - 1
            train_indices = torch.nonzero(buf["timestep"][indices] == update_t).flatten()
            state = buf["state"][indices][train_indices]
            timestep = buf["timestep"][indices][train_indices]
            next_state = buf["next_state"][indices][train_indices]
            rc = self.running_cost(state, next_state, timestep)
            with torch.no_grad():
                target = self.v(next_state).squeeze()
            target = target + self.time_cost
            center = T // 2
            target = target + self.time_cost_sig * torch.sigmoid(-timestep + center) - self.time_cost_sig * torch.sigmoid(-timestep - 1 + center)
            v_xt = self.v(state).squeeze()
            v_loss = F.mse_loss(v_xt, target.detach())
            v_loss.backward()
            if self.value_grad_clip:
                torch.nn.utils.clip_grad_norm_(list(self.val.values()), 0.1)
            self.opt_v.step()
            self.opt_v.zero_grad()
            d_rc[f"running_cost/step_{update_t}_"] = rc.mean().item()
            d_val[f"value/step_{update_t}_"] = v_xt.mean().item()
        logs = {"ebm/d_loss_": d_loss.item(), "ebm/v_loss_": v_loss.item(), "ebm/pos_e_": pos_e.mean().item(),
                "ebm/neg_e_": neg_e.mean().item(), "ebm/running_cost_": rc.mean().item(), "ebm/reg_": reg.item()}
        logs.update(d_rc)
        logs.update(d_val)
        for t, b in enumerate(self.betas_for_q):
            logs[f"adavelreg/beta{t}_"] = b.item()
        return logs

    def update_sampler_mixed_precision(self, buf, record=None):
        """record: optional list receiving, per optimiser step, {name: unscaled gradient} (what opt.step() sees after
        fp16_util.py:204-223 has divided by 2^lg_loss_scale)."""
        permutation = torch.randperm(buf["state"].shape[0])
        B = self.B
        for m in range(0, len(permutation), B):
            for p in self.net.values():
                p.grad = None
            idx = permutation[m:m + B]
            state, t, y = buf["state"][idx], buf["timestep"][idx], buf["y"][idx]
            d = edm.sample_step(self.unet, self.sch, state, t, torch.randn_like(state), log_betas=self.net["log_betas"], y=y)
            rc = self.running_cost(state, d["sample"], t)
            ent = torch.log(d["sigma"].squeeze())
            sv = self.v(d["sample"]).squeeze()
            non_terminal = (t < self.T - self.skip_sampler_tau).float()
            loss = (sv + (rc * self.tau2 - ent * self.tau1) * non_terminal).mean()
            (loss * 2 ** self.lg_loss_scale).backward()
            for p in self.net.values():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad.mul_(1.0 / (2 ** self.lg_loss_scale))
            if record is not None:
                record.append({k: p.grad.detach().clone() for k, p in self.net.items()})
            self.opt.step()
            self.lg_loss_scale += 1e-3
            for p in self.val.values():
                p.grad = None
        logs = {"sampler/sampler_loss_": loss.item(), "sampler/sampler_value_loss_": sv.mean().item(),
                "sampler/running_cost_": rc.mean().item(), "sampler/causal_entropy_": ent.mean().item()}
        for k, s in enumerate(torch.exp(self.net["log_betas"].detach())):
            logs[f"sigma/sigma_{k}_"] = s.item()
        return logs
