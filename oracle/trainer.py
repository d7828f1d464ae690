"""Oracle: one DxMI training step on the CPU (TEST INFRASTRUCTURE).

Restates models/DxMI/trainer.py of the reference for the CIFAR-10 configuration (f is None, no
value_resample): append_buffer :23-55, update_adaptive_vel_reg :218-228, get_running_cost :163-169 with
models/diffusion.py:18-22 (extract), update_f_v :230-346, update_sampler :348-408 — over the oracle's
functional networks with torch autograd and torch.optim.Adam, keeping the reference's literal index
expressions (`buf[key][indices][train_indices]`) so the integer path is the reference's own.
`value_resample=True` (configs/cifar10/T4_ddgan.yaml:39) re-draws every TD step's transition with
`sample_step` (trainer.py:281-285); its `randn_like` comes from torch's CPU generator at the point the
reference draws it (after that step's gather, before the value forwards).
`record` (optional dict) receives the gradients the parity tests compare: the value net's at every
optimizer_v step, the U-Net's (clipped) after the policy step.
"""
import torch
import torch.nn.functional as F

from . import unet_small as ounet
from . import value as ovalue
from . import var_sampler as ovs
from .precision import Precision


class OracleDxMI:
    def __init__(self, net_sd, value_sd, sched, B, T, prec=None, tau1=0.1, tau2=0.01, gamma=1.0, adavelreg=0.99,
                 time_cost=0, time_cost_sig=1.0, lr=1e-7, beta_lr=1e-5, v_lr=1e-5, eta=None, value_resample=False, record=None):
        self.prec = prec or Precision("fp32")
        self.B, self.T = B, T
        self.cfg = ounet.UNetSmallConfig()
        self.net = {k: v.clone().requires_grad_(k not in ("std",)) for k, v in net_sd.items()}
        self.val = {k: v.clone().requires_grad_(True) for k, v in value_sd.items()}
        self.sched = sched
        self.tau1, self.tau2, self.gamma, self.adavelreg = tau1, tau2, gamma, adavelreg
        self.time_cost, self.time_cost_sig = time_cost, time_cost_sig
        self.value_resample, self.record = value_resample, record
        not_beta = [v for k, v in self.net.items() if k not in ("log_betas", "std")]
        self.opt = torch.optim.Adam([{"params": [self.net["log_betas"]], "lr": beta_lr}, {"params": not_beta, "lr": lr}])
        self.opt_v = torch.optim.Adam(list(self.val.values()), lr=v_lr)
        self.betas_for_q = torch.tensor(eta, dtype=torch.float32)  # use_sampler_beta: user_defined_eta (:145-146)

    # ---- networks
    def unet(self, x, t):
        return ounet.forward(self.net, self.cfg, x, t, self.prec)

    def v(self, x):
        return ovalue.forward(self.val, x, self.prec)

    def sample(self, noise):
        with torch.no_grad():
            return ovs.sample(self.unet, self.sched, self.net["log_betas"].detach(), noise)

    # ---- trainer.py:23-55
    @staticmethod
    def append_buffer(buf, d):
        x_seq = d["l_sample"]
        n, T = len(x_seq[0]), len(x_seq) - 1
        for t in range(T):
            buf["state"] = torch.cat((buf["state"], x_seq[t].detach()))
            buf["next_state"] = torch.cat((buf["next_state"], x_seq[t + 1].detach()))
            buf["timestep"] = torch.cat((buf["timestep"], torch.tensor([t] * n)))
            for k in ("logp", "control", "mean", "sigma"):
                buf[k] = torch.cat((buf[k], d[k][t].detach()))
        return buf

    @staticmethod
    def reset_buffer():
        b = {k: torch.FloatTensor() for k in ("state", "next_state", "logp", "control", "mean", "sigma")}
        b["timestep"] = torch.LongTensor()
        return b

    def running_cost(self, state, next_state, t):
        t_rev = self.T - t - 1
        beta = torch.gather(self.betas_for_q, 0, t_rev).reshape(len(t), 1, 1, 1)     # diffusion.py:18-22
        return (((next_state - state) ** 2) / (2 * beta)).view(len(state), -1).mean(dim=1)

    # ---- trainer.py:230-346
    def update_f_v(self, img, d, buf):
        samples = torch.stack(d["l_sample"])
        diff = ((samples[1:] - samples[:-1]) ** 2).view(samples.shape[0] - 1, -1).mean(dim=1).flip(0)
        self.betas_for_q = (self.betas_for_q * self.adavelreg + (1 - self.adavelreg) * diff).detach()
        self.opt_v.zero_grad()
        x0 = d["l_sample"][-1]
        out = self.v(torch.cat((img.detach(), x0.detach()), 0))
        pos_e, neg_e = out[:x0.shape[0]], out[x0.shape[0]:]
        reg = pos_e.pow(2).mean() + neg_e.pow(2).mean()
        d_loss = pos_e.mean() - neg_e.mean() + self.gamma * reg
        d_loss.backward()
        self._rec_v()
        self.opt_v.step()
        self.opt_v.zero_grad()
        d_rc, d_val = {}, {}
        T, B = self.T, self.B
        permutation = torch.randperm(B * T)
        indices = permutation + (buf["state"].shape[0] - B * T)
        for i in range(T):
            update_t = T - i - 1
            train_indices = torch.nonzero(buf["timestep"][indices] == update_t).flatten()
            state = buf["state"][indices][train_indices]
            timestep = buf["timestep"][indices][train_indices]
            if self.value_resample:
                with torch.no_grad():
                    next_state = ovs.sample_step(self.unet, self.sched, self.net["log_betas"].detach(), state, timestep,
                                                 torch.randn_like(state))["sample"]
            else:
                next_state = buf["next_state"][indices][train_indices]
            rc = self.running_cost(state, next_state, timestep)
            with torch.no_grad():
                target = self.v(next_state).squeeze()
            center = T // 2
            target = target + self.time_cost_sig * torch.sigmoid(-timestep + center) - self.time_cost_sig * torch.sigmoid(-timestep - 1 + center)
            target = target + self.time_cost
            v_xt = self.v(state).squeeze()
            v_loss = F.mse_loss(v_xt, target.detach())
            v_loss.backward()
            self._rec_v()
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

    def _rec_v(self):
        if self.record is not None:
            self.record.setdefault("value_grads", []).append({k: p.grad.detach().clone() for k, p in self.val.items()})

    # ---- trainer.py:171-216
    def sample_guidance(self, x0, noises, guidance_scale):
        x = x0
        l_x, l_g, l_on = [x0.clone()], [], []
        for t in range(self.T):
            tt = torch.full((len(x),), t, dtype=torch.long)
            with torch.no_grad():
                d = ovs.sample_step(self.unet, self.sched, self.net["log_betas"].detach(), x, tt, noises[t])
            nx = d["sample"].detach().requires_grad_(True)
            grad = torch.autograd.grad(self.v(nx).squeeze().sum(), nx)[0]
            guidance = grad * guidance_scale * d["sigma"]
            x = (nx + guidance).detach()
            l_on.append(torch.distributions.Normal(d["mean"], d["sigma"]).log_prob(x).mean(-1).mean(-1).mean(-1))
            l_g.append(guidance)
            l_x.append(x.clone())
        return {"sample": x, "l_sample": l_x, "guidance": l_g, "logp_on": l_on}

    # ---- trainer.py:348-408 (n_generator = 1)
    def update_sampler(self, buf, z):
        permutation = torch.randperm(buf["state"].shape[0])
        self.opt.zero_grad()
        idx = permutation[: self.B]
        state, t = buf["state"][idx], buf["timestep"][idx]
        d = ovs.sample_step(self.unet, self.sched, self.net["log_betas"], state, t, z)
        rc = self.running_cost(state, d["sample"], t)
        ent = torch.log(d["sigma"].squeeze())
        sv = self.v(d["sample"]).squeeze()
        loss = (sv + (rc * self.tau2 - ent * self.tau1) * (t < self.T).float()).mean()
        loss.backward()
        for p in self.val.values():   # value grads are discarded by the next zero_grad (trainer.py:235)
            p.grad = None
        torch.nn.utils.clip_grad_norm_([p for k, p in self.net.items() if p.requires_grad], 0.1)
        if self.record is not None:
            self.record["net_grads"] = {k: p.grad.detach().clone() for k, p in self.net.items() if p.grad is not None}
        self.opt.step()
        logs = {"sampler/sampler_loss_": loss.item(), "sampler/sampler_value_loss_": sv.mean().item(),
                "sampler/running_cost_": rc.mean().item(), "sampler/causal_entropy_": ent.mean().item()}
        for k, s in enumerate(torch.exp(self.net["log_betas"].detach())):
            logs[f"sigma/sigma_{k}_"] = s.item()
        return logs


class OracleDxMI_EV(OracleDxMI):
    """models/DxMI/trainer.py:865-1078 (DxMI_Trainer_EV): a separate energy f(x) next to the value v(x, t).  `energy_sd` holds
    the IGEBMEncoderV2 weights of f without the value wrapper's "net." prefix; v is the time-independent wrapper of the same
    architecture (t is ignored, models/value.py:8-12).  Differences from the parent are the reference's own: the contrastive
    step runs on f with its gradient clipped at 0.1 and no regulariser (:956-973); every TD step re-draws the transition with
    `sample_step` (:976-978, gradient enabled but cut by the detached target) and its target is v(x', t+1) for the non-terminal
    steps, f(x') for the last, plus tau2 * running_cost - tau1 * log sigma (:983-987); the policy step uses the same mixed
    terminal value (:1047-1056); log keys as in :1003-1017 and :1062-1076."""

    def __init__(self, net_sd, value_sd, energy_sd, sched, B, T, f_lr=1e-5, **kw):
        super().__init__(net_sd, value_sd, sched, B, T, **kw)
        self.fsd = {k: v.clone().requires_grad_(True) for k, v in energy_sd.items()}
        self.opt_f = torch.optim.Adam(list(self.fsd.values()), lr=f_lr)
        self.betas_for_q = torch.exp(self.net["log_betas"]).detach()              # :912-914 (use_sampler_beta)

    def f(self, x):
        return ovalue.forward(self.fsd, x, self.prec, prefix="")

    def _mix(self, next_state, t):
        non_terminal = (t < self.T - 1).float()
        return self.v(next_state).squeeze() * non_terminal + self.f(next_state).squeeze() * (1 - non_terminal)

    def update_f_v(self, img, d, buf):
        samples = torch.stack(d["l_sample"])
        diff = ((samples[1:] - samples[:-1]) ** 2).view(samples.shape[0] - 1, -1).mean(dim=1).flip(0)
        self.betas_for_q = (self.betas_for_q * self.adavelreg + (1 - self.adavelreg) * diff).detach()
        x0 = d["l_sample"][-1]
        self.opt_f.zero_grad()
        out = self.f(torch.cat((img.detach(), x0.detach()), 0))
        pos_e, neg_e = out[:x0.shape[0]], out[x0.shape[0]:]
        d_loss = pos_e.mean() - neg_e.mean()
        d_loss.backward()
        torch.nn.utils.clip_grad_norm_(list(self.fsd.values()), 0.1)
        if self.record is not None:
            self.record["f_grads"] = {k: p.grad.detach().clone() for k, p in self.fsd.items()}
        self.opt_f.step()
        self.opt_f.zero_grad()
        self.opt_v.zero_grad()
        T, B = self.T, self.B
        permutation = torch.randperm(B * T)
        indices = permutation + (buf["state"].shape[0] - B * T)
        d_rc = {}
        for i in range(T):
            update_t = T - i - 1
            train_indices = torch.nonzero(buf["timestep"][indices] == update_t).flatten()
            state = buf["state"][indices][train_indices]
            timestep = buf["timestep"][indices][train_indices]
            ds = ovs.sample_step(self.unet, self.sched, self.net["log_betas"], state, timestep, torch.randn_like(state))
            next_state = ds["sample"]
            rc = self.running_cost(state, next_state, timestep)
            ent = torch.log(ds["sigma"].squeeze())
            target = self._mix(next_state, timestep) + rc * self.tau2 - ent * self.tau1
            v_xt = self.v(state).squeeze()
            v_loss = F.mse_loss(v_xt, target.detach())
            for p in self.val.values():
                p.grad = None
            v_loss.backward(inputs=list(self.val.values()))
            self._rec_v()
            self.opt_v.step()
            self.opt_v.zero_grad()
            d_rc[f"running_cost/step_{update_t}_"] = rc.mean().item()
        logs = {"ebm/d_loss_": d_loss.item(), "ebm/v_loss_": v_loss.item(), "ebm/pos_e_": pos_e.mean().item(),
                "ebm/neg_e_": neg_e.mean().item(), "ebm/running_cost_": rc.mean().item()}
        logs.update(d_rc)
        for t, b in enumerate(self.betas_for_q):
            logs[f"adavelreg/beta_for_q_{t}_"] = b.item()
        return logs

    def update_sampler(self, buf, z):
        permutation = torch.randperm(buf["state"].shape[0])
        self.opt.zero_grad()
        idx = permutation[: self.B]
        state, t = buf["state"][idx], buf["timestep"][idx]
        d = ovs.sample_step(self.unet, self.sched, self.net["log_betas"], state, t, z)
        rc = self.running_cost(state, d["sample"], t).mean()
        ent = torch.log(d["sigma"].squeeze()).mean()
        sv = self._mix(d["sample"], t).mean()
        loss = sv + rc * self.tau2 - ent * self.tau1
        loss.backward(inputs=[p for p in self.net.values() if p.requires_grad])
        torch.nn.utils.clip_grad_norm_([p for k, p in self.net.items() if p.requires_grad], 0.1)
        if self.record is not None:
            self.record["net_grads"] = {k: p.grad.detach().clone() for k, p in self.net.items() if p.grad is not None}
        self.opt.step()
        logs = {"sampler/sampler_loss_": loss.item(), "sampler/sampler_value_loss_": sv.item(),
                "sampler/running_cost_": rc.item(), "sampler/causal_entropy_": ent.item()}
        for k, s in enumerate(torch.exp(self.net["log_betas"].detach())):
            logs[f"beta/beta_{k}_"] = s.item()
        return logs
