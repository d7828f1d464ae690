"""DxMI trainer (`models.DxMI.trainer.DxMI_Trainer`, `append_buffer`, `reset_buffer`) for the HIP path.

Behaviour follows the reference algorithm (models/DxMI/trainer.py:23-70, 73-408) — contrastive energy
step, T backward-ordered TD value steps on the replayed transitions, policy step through V with the
velocity / entropy regularisers, adaptive velocity regulariser — including its order-dependent details
(SURVEY 7): `betas_for_q` EMA before the value update, `target = V(x')` (the velocity-augmented target
only survives on the energy branch), `randperm` drawn from torch's CPU generator, the
len(img)==len(x0) slicing, clip-then-step.  What changes is how the work runs on the GPU:
  * transitions live in a preallocated device ring (models/DxMI/replay.py) that the sampler's fused transition
    kernel writes directly: `append_buffer` moves no data (the reference re-copies the growing buffer T times per
    field: O(T^2) traffic).  A reference-style dict buffer is still accepted (one cat per field);
  * the TD loop gathers each step's rows once with the INT gather kernel (dxmi_gather_rows) instead of
    materialising the permuted T*B buffer five times per step, and finds each step's rows with ONE stable sort of
    the permuted timesteps (same rows, same order as the reference's per-step `nonzero`, no host sync per step);
  * gradient clipping computes the norm and the clip coefficient on the device (dxmi_gradnorm_clip: no `.item()`);
  * value targets are evaluated under no_grad (the reference builds and discards an autograd graph);
  * the logged scalars are collected on the device and synchronised ONCE per call instead of ~2T+10
    `.item()` round trips.
Networks run through their HIP autograd functions (models/value_train.py, unet_small_train.py).
"""
import torch
import torch.nn.functional as F

from dxmi_hip import graph as _graph
from dxmi_hip import ops
from ..diffusion import extract, make_beta_schedule
from .replay import TransitionRing, buffer_gather, buffer_rows


def _set_mode(module, training):
    """module.train() / module.eval() when the mode actually changes: the recursive flag walk over the module tree ran 24 times per
    train step (the reference toggles the value net around every TD target), ~2 ms of host time.  "Changes" is judged against the
    mode THIS function applied last (kept on the module) and the root flag: whoever toggled a child alone, or the root behind the
    trainer's back, gets the full recursive walk again — the reference re-applies the mode unconditionally
    (models/DxMI/trainer.py:261-322)."""
    if getattr(module, "training", None) is None:          # OpenAIDiffusion: a plain object forwarding train() / eval() to its net
        (module.train if training else module.eval)()
        return
    last = module.__dict__.get("_dxmi_mode_applied")
    if last is None or last[0] != training or module.training != training or _children_disagree(module, training, last):
        module.train(training)
        module.__dict__["_dxmi_mode_applied"] = (training, _mode_probe(module))


def _mode_probe(module):
    """The (few) direct children whose flags stand for the subtree: toggling any of them alone is noticed by _set_mode."""
    return [m for m in module._modules.values() if m is not None]


def _children_disagree(module, training, last):
    probe = last[1]
    if len(probe) != len(module._modules):
        return True
    for m in probe:
        if m.training != training:
            return True
    return False


def _pack_modules(net):
    """Sub-modules of `net` that keep packed bf16 weight sets (objects with refresh_packs / prepare_capture)."""
    mods = net.modules() if hasattr(net, "modules") else [net]
    out = [m for m in mods if hasattr(m, "refresh_packs") and hasattr(m, "prepare_capture")]
    inner = getattr(net, "net", None)          # OpenAIDiffusion: a plain object holding the U-Net
    if not out and inner is not None and hasattr(inner, "modules"):
        out = [m for m in inner.modules() if hasattr(m, "refresh_packs") and hasattr(m, "prepare_capture")]
    return out


def reset_buffer(device, ring=None):
    """Empty transition buffer (reference :58-70).  ring: a TransitionRing to recycle (its rows are dropped)."""
    if ring is not None:
        return ring.reset()
    d = {k: torch.FloatTensor().to(device) for k in ("state", "next_state", "final", "logp", "control", "entropy", "mean", "sigma")}
    d["timestep"] = torch.LongTensor().to(device)
    d["y"] = torch.LongTensor().to(device)
    return d


def append_buffer(state_buffer, d_sample):
    """Flatten a sampled trajectory into (state, next_state, timestep, ...) rows, time-major blocks of
    n_sample (reference :23-55).  Same resulting tensors; one cat per field."""
    if isinstance(state_buffer, TransitionRing):
        return state_buffer.append(d_sample)
    x_seq = d_sample["l_sample"]
    n_sample, n_seq = len(x_seq[0]), len(x_seq) - 1
    device = x_seq[0].device

    def put(key, pieces):
        state_buffer[key] = torch.cat([state_buffer[key]] + [p.detach() for p in pieces])

    put("state", x_seq[:-1])
    put("next_state", x_seq[1:])
    state_buffer["timestep"] = torch.cat([state_buffer["timestep"],
                                          torch.arange(n_seq, device=device).repeat_interleave(n_sample)])
    put("final", [x_seq[-1]] * n_seq)
    for key in ("logp", "control", "entropy", "mean", "sigma"):
        if key in d_sample:
            put(key, d_sample[key][:n_seq])
    if d_sample.get("y") is not None:   # unconditional OpenAIDiffusion hands back y=None
        put("y", [d_sample["y"]] * n_seq)
    return state_buffer


class DxMI_Trainer:
    def __init__(self, batchsize, tau1=0., tau2=0., gamma=None, q_beta_schedule="constant", q_beta_start=1., q_beta_end=1.,
                 adavelreg=None, n_timesteps=10, value_update_order="backward", entropy_in_value=None,
                 velocity_in_value=None, use_sampler_beta=False, time_cost=None, time_cost_sig=None,
                 repeat_value_update=1, value_resample=False, value_grad_clip=False, skip_sampler_tau=0):
        """Same keyword surface as the reference (:74-133)."""
        self.batchsize, self.n_timesteps = batchsize, n_timesteps
        self.gamma, self.tau1, self.tau2 = gamma, tau1, tau2
        self.value_update_order = value_update_order
        self.entropy_in_value, self.velocity_in_value = entropy_in_value, velocity_in_value
        self.q_beta_schedule, self.q_beta_start, self.q_beta_end = q_beta_schedule, q_beta_start, q_beta_end
        self.adavelreg = adavelreg
        self.use_sampler_beta = use_sampler_beta
        self.time_cost, self.time_cost_sig = time_cost, time_cost_sig
        self.repeat_value_update = repeat_value_update
        self.value_resample = value_resample
        self.value_grad_clip = value_grad_clip
        self.skip_sampler_tau = skip_sampler_tau

    # ------------------------------------------------------------------ hipGraph replay of the two updates (dxmi_hip/graph.py)
    use_graphs = False      # set True (train_cifar10.py does): update_f_v / update_sampler on a TransitionRing replay as hipGraphs

    def _step_graph(self, name, state_dict, extra_key, fn, nets):
        """The StepGraph of one update for one (ring, fill level, shapes) — or None when the call cannot be replayed: graphs off,
        a reference-style dict buffer (row counts grow with every append), or a capture already running (a caller capturing a
        larger step)."""
        if not self.use_graphs or not isinstance(state_dict, TransitionRing) or _graph.capturing():
            return None
        graphs = self.__dict__.setdefault("_graphs", {})
        key = (name, id(state_dict), state_dict.filled) + tuple(extra_key)
        g = graphs.get(key)
        if g is None:
            device = state_dict.device
            if torch.is_tensor(getattr(self, "betas_for_q", None)) and self.betas_for_q.device != device:
                self.betas_for_q = self.betas_for_q.to(device)       # q-betas are indexed on the device inside the step
            mods = [m for net in nets if net is not None for m in _pack_modules(net)]
            g = graphs[key] = _graph.StepGraph(fn, device, modules=mods, name=f"{type(self).__name__}.{name}")
        return g

    @staticmethod
    def _randperm(n, device):
        """torch.randperm(n) from the CPU generator, as the reference draws it (:271, :352), on the device.  Inside a StepGraph
        capture it is a host input of the graph: drawn per replay from the same generator, uploaded with the other host inputs."""
        cap = _graph.current()
        if cap is None:
            return torch.randperm(n).to(device)
        return cap.host_input(torch.int64, n, lambda: torch.randperm(n))

    def _set_betas_for_q(self, new):
        """In place where possible: a captured step reads and writes ONE persistent tensor (a re-bound attribute would leave the
        graph updating a buffer nobody reads)."""
        cur = self.betas_for_q
        if torch.is_tensor(cur) and cur.device == new.device and cur.shape == new.shape and cur.dtype == new.dtype:
            cur.copy_(new)
        else:
            if _graph.capturing():
                raise RuntimeError("betas_for_q changes device / shape inside a StepGraph capture")
            self.betas_for_q = new.detach()

    def set_models(self, f, v, sampler, optimizer, optimizer_fstar, optimizer_v):
        """reference :136-161."""
        self.f, self.v, self.sampler = f, v, sampler
        self.optimizer, self.optimizer_fstar, self.optimizer_v = optimizer, optimizer_fstar, optimizer_v
        if self.f is not None:
            raise NotImplementedError("separate energy network f: used only by the 2-D toy configs, not on the HIP path")
        # data-parallel gradient exchange (replaces the reference's DDP wrappers, train_cifar10.py:298-309):
        # no-ops on a single process
        from dxmi_hip.dist import FlatGradSync
        self.sync_v = FlatGradSync(v) if v is not None else (lambda: None)
        self.sync_sampler = FlatGradSync(sampler) if sampler is not None else (lambda: None)
        if self.use_sampler_beta:
            if hasattr(self.sampler, "user_defined_eta"):
                self.betas_for_q = torch.tensor(self.sampler.user_defined_eta, dtype=torch.float32)
            elif hasattr(self.sampler, "log_betas"):
                self.betas_for_q = torch.exp(self.sampler.log_betas).detach()
            elif hasattr(self.sampler.net, "log_betas"):
                self.betas_for_q = torch.exp(self.sampler.net.log_betas).detach()
        else:
            self.betas_for_q = make_beta_schedule(schedule=self.q_beta_schedule, n_timesteps=self.n_timesteps,
                                                  start=self.q_beta_start, end=self.q_beta_end)

    # ------------------------------------------------------------------ pieces of the objective
    def get_running_cost(self, state, next_state, pred_mean, pred_std, t):
        """||x' - x||^2 / (2 beta_{T-1-t}) averaged over CHW (reference :163-169); integer gather exact."""
        beta_next = extract(self.betas_for_q, self.n_timesteps - t - 1, state).to(state.device)
        return (((next_state - state) ** 2) / (2 * beta_next)).view(len(state), -1).mean(dim=1)

    def update_adaptive_vel_reg(self, d_sample):
        """EMA of the per-step mean squared displacement, reversed to q-time (reference :218-228)."""
        device = d_sample["sample"].device
        samples = torch.stack(d_sample["l_sample"])
        diff = ((samples[1:] - samples[:-1]) ** 2).view(samples.shape[0] - 1, -1).mean(dim=1).flip(0).to(device)
        if self.betas_for_q.device != device:
            self.betas_for_q = self.betas_for_q.to(device)
        self._set_betas_for_q((self.betas_for_q * self.adavelreg + (1 - self.adavelreg) * diff).detach())

    def _time_cost_terms(self, timestep):
        extra = 0.
        if self.time_cost_sig is not None:
            center = self.n_timesteps // 2
            extra = self.time_cost_sig * torch.sigmoid(-timestep + center) - self.time_cost_sig * torch.sigmoid(-timestep - 1 + center)
        return extra

    PAIR_TD_FORWARD = True        # one value-net forward per TD step for target + prediction (A/B switch; results agree to bf16 noise)

    # ------------------------------------------------------------------ value / energy update
    def update_f_v(self, img, d_sample, state_dict):
        """reference :230-346."""
        g = None
        if isinstance(state_dict, TransitionRing) and d_sample.get("_ring_slot") == (id(state_dict), state_dict.filled - 1) \
                and len(img) == self.batchsize:
            # the trajectory IS the ring slot: every tensor of the step but `img` has a fixed address -> replayable
            nets = [self.v] + ([self.sampler] if self.value_resample else [])
            g = self._step_graph("update_f_v", state_dict, tuple(img.shape),
                                 lambda im, d=d_sample, sd=state_dict: self._update_f_v(im, d, sd), nets)
        keys, vals = self._update_f_v(img, d_sample, state_dict) if g is None else g(img.detach())
        return dict(zip(keys, vals.tolist()))          # the one device -> host synchronisation of the call

    def _update_f_v(self, img, d_sample, state_dict):
        x_seq = d_sample["l_sample"]
        if self.adavelreg is not None:
            self.update_adaptive_vel_reg(d_sample)
        self.optimizer_v.zero_grad()
        x0 = x_seq[-1]
        n_steps, batchsize, device = self.n_timesteps, self.batchsize, img.device
        _set_mode(self.v, True)
        # energy step: the last value step is the energy
        Tt = n_steps * torch.ones(len(img) + len(x0), dtype=torch.long, device=device)
        output = self.v(torch.cat((img.detach(), x0.detach()), 0), Tt)
        pos_e, neg_e = output[:x0.shape[0]], output[x0.shape[0]:]
        d_loss = pos_e.mean() - neg_e.mean()
        if self.gamma is not None:
            reg = pos_e.pow(2).mean() + neg_e.pow(2).mean()
            d_loss = d_loss + self.gamma * reg
        else:
            reg = torch.zeros((), device=device)
        d_loss.backward()
        self.sync_v()
        self.optimizer_v.step()
        self.optimizer_v.zero_grad()
        d_running_cost, d_value = {}, {}

        # TD value estimation over the last T*B buffered transitions
        permutation = self._randperm(batchsize * n_steps, device)            # CPU generator, as the reference
        indices = permutation + (buffer_rows(state_dict) - batchsize * n_steps)
        rows_all, ts_all = self._td_rows(state_dict, indices, n_steps, batchsize)
        need_entropy = bool(self.entropy_in_value or self.entropy_in_value == 0)
        running_cost = v_loss = None
        fused = (self.FUSED_TD_STEP and self.PAIR_TD_FORWARD and isinstance(state_dict, TransitionRing) and not self.value_resample
                 and not need_entropy and self.velocity_in_value is None and getattr(self.v, "forward_pair_packed", None) is not None
                 and torch.is_tensor(self.betas_for_q) and self.betas_for_q.device == device and self.betas_for_q.dtype == torch.float32)
        if fused:
            v_loss_t, cost_mean_t = self._td_steps_fused(state_dict, rows_all, n_steps, batchsize, device, d_running_cost, d_value)
            if v_loss_t is None:
                fused = False
        if fused:
            v_loss, running_cost_mean = v_loss_t, cost_mean_t
        else:
            running_cost_mean = None
        for i in range(0 if not fused else n_steps, n_steps):
            update_t = n_steps - i - 1
            rows, timestep = rows_all[update_t], ts_all[update_t]             # == [indices][train_indices], gathered once
            state = buffer_gather(state_dict, "state", rows)
            pred_mean = pred_std = entropy = None
            if self.value_resample:
                with torch.no_grad():
                    d_step = self.sampler.sample_step(state, timestep)
                next_state, pred_mean, pred_std = d_step["sample"], d_step["mean"], d_step["sigma"]
            else:
                next_state = buffer_gather(state_dict, "next_state", rows)
                if need_entropy:       # mean / sigma rows are read only by the entropy term (result-neutral skip)
                    pred_std = buffer_gather(state_dict, "sigma", rows)
            running_cost = self.get_running_cost(state, next_state, pred_mean, pred_std, timestep)
            if need_entropy:
                entropy = torch.log(pred_std.squeeze())
            # TD target v(next_state) and TD prediction v(state): same parameters, so ONE forward over [next_state | state] where the
            # value plugin offers it (models.value.TimeIndependentValue.forward_pair; the reference evaluates the target under eval()
            # and the prediction under train(), :288-300 — no layer of the HIP value nets reads the flag); only the prediction's half
            # is back-propagated.  Halves the launches of the value net's small maps (one tile per CU at 256 images).
            pair = getattr(self.v, "forward_pair", None) if self.PAIR_TD_FORWARD else None
            if pair is not None:
                _set_mode(self.v, True)
                target, v_xt = pair(next_state, timestep + 1, state, timestep)
                target, v_xt = target.squeeze(), v_xt.squeeze()
            else:
                _set_mode(self.v, False)
                with torch.no_grad():
                    target = self.v(next_state, timestep + 1).squeeze()
            target = target + self._time_cost_terms(timestep)
            if self.time_cost is not None:
                target = target + self.time_cost
            if self.velocity_in_value is not None:
                target = target + running_cost * self.tau2 * (timestep < n_steps - self.velocity_in_value).float()
            if self.entropy_in_value or self.entropy_in_value == 0:
                assert isinstance(self.entropy_in_value, int), "self.entropy_in_value should be interger"
                target = target - entropy * self.tau1 * (timestep < n_steps - self.entropy_in_value).float()
            if pair is None:
                _set_mode(self.v, True)
                v_xt = self.v(state, timestep).squeeze()
            v_loss = F.mse_loss(v_xt, target.detach())
            v_loss.backward()
            self.sync_v()
            if self.value_grad_clip:
                self._clip(ops.fast_parameters(self.v), 0.1)
            self.optimizer_v.step()
            self.optimizer_v.zero_grad()
            d_running_cost[f"running_cost/step_{update_t}_"] = running_cost.detach().mean()
            d_value[f"value/step_{update_t}_"] = v_xt.detach().mean()
        if running_cost_mean is None:
            running_cost_mean = running_cost.detach().mean()
        logs = {"ebm/d_loss_": d_loss.detach(), "ebm/v_loss_": v_loss.detach(), "ebm/pos_e_": pos_e.detach().mean(),
                "ebm/neg_e_": neg_e.detach().mean(), "ebm/running_cost_": running_cost_mean, "ebm/reg_": reg.detach()}
        logs.update(d_running_cost)
        logs.update(d_value)
        if self.adavelreg is not None:
            for t, beta in enumerate(self.betas_for_q):
                logs[f"adavelreg/beta{t}_"] = beta
        return self._stack_logs(logs)

    FUSED_TD_STEP = True          # ring buffer: one gather + cost launch and one loss launch per TD step (A/B switch)

    def _td_steps_fused(self, ring, rows_all, n_steps, batchsize, device, d_running_cost, d_value):
        """The T TD steps of update_f_v on a TransitionRing with the elementwise work of a step in TWO launches (round 6: it was ~40
        torch launches per step — index arithmetic, two gathers, the cat of the paired forward, running cost, time-cost terms,
        mse forward / backward, the logged means): dxmi_td_gather_cost writes [next_state | state] of the step's rows straight into
        the batch the value net evaluates and reduces the running cost; dxmi_td_loss forms the TD target (v(x') + the step's
        time-cost terms, detached), the loss, its gradient and the logged means.  Same arithmetic per element as the generic loop
        (reference trainer.py:271-325); the means are summed in a fixed order of their own.  Every row of a step shares its
        timestep on the ring, so the per-row gathers of beta and of the time-cost terms are one scalar per step."""
        from dxmi_hip import ops
        T, B = n_steps, batchsize
        shape = ring.shape
        srows = ring.storage_rows(rows_all.reshape(-1), "state").view(T, B)       # index arithmetic of all T steps at once
        nrows = srows + B                                                           # next_state of a row: B rows further in the block
        traj2d = ring.traj.view(-1, ring.traj[0, 0, 0].numel())
        extra = None
        if self.time_cost_sig is not None or self.time_cost is not None:
            tt = torch.arange(T, device=device)
            extra = self._time_cost_terms(tt) if self.time_cost_sig is not None else torch.zeros(T, device=device)
            if self.time_cost is not None:
                extra = extra + self.time_cost
            extra = extra.float().contiguous()
        v_loss = cost_mean = None
        for i in range(T):
            t = T - i - 1
            pair_x = torch.empty((2 * B,) + tuple(shape), dtype=torch.float32, device=device)
            cost = ops.td_gather_cost(traj2d, srows[t], self.betas_for_q[T - 1 - t:T - t], next_rows=nrows[t], out_pair=pair_x)
            _set_mode(self.v, True)
            res = self.v.forward_pair_packed(pair_x, B)
            if res is None:
                return None, None
            grad, lg = ops.td_loss(res.detach().reshape(-1), cost, None if extra is None else extra[t:t + 1])
            res.backward(gradient=grad.view_as(res))
            self.sync_v()
            if self.value_grad_clip:
                self._clip(ops.fast_parameters(self.v), 0.1)
            self.optimizer_v.step()
            self.optimizer_v.zero_grad()
            d_running_cost[f"running_cost/step_{t}_"] = lg[2]
            d_value[f"value/step_{t}_"] = lg[1]
            v_loss, cost_mean = lg[0], lg[2]
        return v_loss, cost_mean

    # ------------------------------------------------------------------ policy update
    def update_sampler(self, state_dict, n_generator, d_sample=None):
        """reference :348-408."""
        g = self._step_graph("update_sampler", state_dict, (n_generator,),
                             lambda sd=state_dict: self._update_sampler(sd, n_generator), [self.sampler, self.v])
        keys, vals = self._update_sampler(state_dict, n_generator) if g is None else g()
        return dict(zip(keys, vals.tolist()))

    def _update_sampler(self, state_dict, n_generator):
        _set_mode(self.v, False)
        _set_mode(self.sampler, True)
        device = state_dict.device if isinstance(state_dict, TransitionRing) else state_dict["state"].device
        permutation = self._randperm(buffer_rows(state_dict), device)
        batchsize = self.batchsize
        n_data = min(len(permutation), batchsize * n_generator)
        for m in range(0, n_data, batchsize):
            self.optimizer.zero_grad()
            indices = permutation[m:m + batchsize]
            state = buffer_gather(state_dict, "state", indices)
            t = buffer_gather(state_dict, "timestep", indices)
            d_step = self.sampler.sample_step(state, t)
            next_state, pred_mean, pred_std = d_step["sample"], d_step["mean"], d_step["sigma"]
            running_cost = self.get_running_cost(state, next_state, pred_mean, pred_std, t)
            causal_entropy = torch.log(pred_std.squeeze())
            with self._frozen(self.v):
                sampler_value_loss = self.v(next_state, t + 1).squeeze()
            non_terminal = (t < self.n_timesteps - self.skip_sampler_tau).float()
            sampler_loss = (sampler_value_loss + (running_cost * self.tau2 - causal_entropy * self.tau1) * non_terminal).mean()
            sampler_loss.backward()
            self.sync_sampler()   # the value net's side-effect gradients of this backward are discarded, not reduced
            self._clip(ops.fast_parameters(self.sampler), 0.1)
            self.optimizer.step()
        logs = {"sampler/sampler_loss_": sampler_loss.detach(), "sampler/sampler_value_loss_": sampler_value_loss.detach().mean(),
                "sampler/running_cost_": running_cost.detach().mean(), "sampler/causal_entropy_": causal_entropy.detach().mean()}
        if self.sampler.trainable_beta:
            net = self.sampler.net.module if hasattr(self.sampler.net, "module") else self.sampler.net
            sigma = torch.exp(net.log_betas.detach())
            for t in range(len(sigma)):
                logs[f"sigma/sigma_{t}_"] = sigma[t]
        return self._stack_logs(logs)

    # ------------------------------------------------------------------ value-guided sampling
    def sample_guidance(self, n_sample, device, x0=None, guidance_scale=None, t_select=None, noise=None):
        """reference :171-216: after every sampler transition, move the sample along the gradient of the value net at
        the next step, scaled by guidance_scale * sigma.  The transition is the fused HIP step; the input gradient of the
        value net is its HIP backward (models/value_train.py).  noise: optional per-step draws (extension, as in sample)."""
        from torch.distributions import Normal
        from ..modules import process_single_t
        assert guidance_scale is not None, "guidance_scale must be given"
        _set_mode(self.v, False)
        if x0 is None:
            x0 = self._guidance_x0_scale(torch.randn(n_sample, *self.sampler.sample_shape, device=device))
        x0 = x0.to(device)
        l_x, l_guidance, l_logp, l_logp_orig = [x0.detach().clone()], [], [], []
        x = x0
        kw = self._guidance_model_kwargs(n_sample, device)
        for t in range(self.n_timesteps):
            tt = process_single_t(x, t)
            with torch.no_grad():
                d_step = self.sampler.sample_step(x, tt, **kw) if noise is None else self.sampler.sample_step(x, tt, noise=noise[t], **kw)
            next_x = d_step["sample"].detach()
            with torch.enable_grad():
                next_x = next_x.requires_grad_(True)
                value = self.v(next_x, tt + 1).squeeze()
                grad = torch.autograd.grad(value.sum(), next_x)[0]
            sigma = d_step["sigma"]
            sigma = sigma.reshape(-1, *([1] * (x.ndim - 1))) if sigma.dim() == 1 else sigma
            guidance = grad * guidance_scale * sigma
            x = next_x + guidance if (t_select is None or t in t_select) else next_x
            x = x.detach()
            l_logp.append(d_step.get("logp", torch.tensor(0.)))
            l_logp_orig.append(Normal(d_step["mean"], sigma).log_prob(x).mean(-1).mean(-1).mean(-1))
            l_guidance.append(guidance)
            l_x.append(x.clone())
        return {"sample": x, "l_sample": l_x, "logp": l_logp, "logp_on": l_logp_orig,
                "logp_traj": torch.stack([torch.as_tensor(v).to(x.device) * torch.ones(len(x), device=x.device) for v in l_logp]).sum(dim=0),
                "logp_on_traj": torch.stack(l_logp_orig).sum(dim=0), "guidance": l_guidance}

    def _guidance_x0_scale(self, x0):
        return x0

    def _guidance_model_kwargs(self, n_sample, device):
        return {}

    class _frozen:
        """Context: the module's parameters do not ask for gradients (the value net inside the policy step: the reference
        accumulates and then discards them, trainer.py:235 / :387 — skipping the weight-gradient kernels changes nothing)."""

        def __init__(self, module):
            self.ps = [p for p in ops.fast_parameters(module) if p.requires_grad]

        def __enter__(self):
            for p in self.ps:
                p.requires_grad_(False)

        def __exit__(self, *exc):
            for p in self.ps:
                p.requires_grad_(True)

    @staticmethod
    def _clip(parameters, max_norm):
        """torch.nn.utils.clip_grad_norm_ with the norm and the coefficient kept on the device."""
        ops.clip_grad_norm_(list(parameters), max_norm)

    @staticmethod
    def _td_rows(state_dict, indices, n_steps, batchsize):
        """Rows (and their timesteps) of every TD step, indexed by t: the reference takes, per step,
        `indices[nonzero(timestep[indices] == t)]` (trainer.py:278-280).
        Ring buffer: every timestep owns exactly `batchsize` of the last T*B rows (whole trajectories of B samples), so
        ONE stable sort of the permuted timesteps yields the same rows in the same order for all t at once, with no host
        sync.  Reference-style dict buffer: the reference's own per-step nonzero (row counts may be ragged there)."""
        if isinstance(state_dict, TransitionRing):
            assert state_dict.B == batchsize and state_dict.T == n_steps, "ring geometry differs from the trainer's"
            ts_perm = state_dict.timestep_of(indices)
            order = torch.sort(ts_perm, stable=True).indices
            return indices[order].view(n_steps, batchsize), ts_perm[order].view(n_steps, batchsize)
        ts_perm = state_dict["timestep"][indices]
        rows = [indices[torch.nonzero(ts_perm == t).flatten()] for t in range(n_steps)]
        return rows, [state_dict["timestep"][r] for r in rows]

    @staticmethod
    def _stack_logs(logs):
        """(keys, fp32 device vector of the values): no synchronisation — the caller reads it back once."""
        keys = list(logs.keys())
        dev = next(v.device for v in logs.values() if torch.is_tensor(v))
        return keys, torch.stack([torch.as_tensor(logs[k], dtype=torch.float32, device=dev).reshape(()) for k in keys])

    @classmethod
    def _to_floats(cls, logs):
        """One device->host synchronisation for the whole dictionary."""
        keys, vals = cls._stack_logs(logs)
        return dict(zip(keys, vals.tolist()))


class DxMI_Trainer_Cond(DxMI_Trainer):
    """DxMI trainer of the EDM backbones (ImageNet-64, LSUN): time-independent value net, optional class labels,
    sampler update over ALL T*B buffered transitions through a MixedPrecisionTrainer (reference :412-809).
    Differences from DxMI_Trainer are the reference's own: `betas_for_q = sigmas[:-1]**2` for an OpenAIDiffusion
    sampler (:516-517), forward beta ordering by default (:527-534), extra value-target options, `y` plumbing."""

    def __init__(self, batchsize, tau1=0.0, tau2=0.0, gamma=None, q_beta_schedule="constant", q_beta_start=1.0, q_beta_end=1.0,
                 adavelreg=None, n_timesteps=10, value_update_order="backward", entropy_in_value=None, velocity_in_value=None,
                 entropy_value_scale=1, skip_sampler_tau=0, sigma_scale=None, use_sampler_beta=False, aug=None, time_cost=None,
                 time_cost_sig=None, time_cost_sig_center=None, repeat_value_update=1, skip_running_last=False,
                 value_resample=False, beta_ordering="forward", value_grad_clip=False):
        super().__init__(batchsize, tau1=tau1, tau2=tau2, gamma=gamma, q_beta_schedule=q_beta_schedule, q_beta_start=q_beta_start,
                         q_beta_end=q_beta_end, adavelreg=adavelreg, n_timesteps=n_timesteps, value_update_order=value_update_order,
                         entropy_in_value=entropy_in_value, velocity_in_value=velocity_in_value, use_sampler_beta=use_sampler_beta,
                         time_cost=time_cost, time_cost_sig=time_cost_sig, repeat_value_update=repeat_value_update,
                         value_resample=value_resample, value_grad_clip=value_grad_clip, skip_sampler_tau=skip_sampler_tau)
        if aug is not None:
            raise NotImplementedError("aug (StyleGAN2-ADA augmentation) is not part of any shipped DxMI config")
        assert beta_ordering in {"reversed", "forward"}
        self.entropy_value_scale, self.sigma_scale = entropy_value_scale, sigma_scale
        self.time_cost_sig_center, self.skip_running_last, self.beta_ordering = time_cost_sig_center, skip_running_last, beta_ordering

    def set_models(self, v, sampler, optimizer, optimizer_v, f=None, optimizer_fstar=None):
        """reference :496-525."""
        self.f, self.v, self.sampler = f, v, sampler
        self.optimizer, self.optimizer_fstar, self.optimizer_v = optimizer, optimizer_fstar, optimizer_v
        from dxmi_hip.dist import FlatGradSync
        self.sync_v = FlatGradSync(v) if v is not None else (lambda: None)
        self.sync_sampler = lambda: None   # the MixedPrecisionTrainer reduces the sampler gradients in optimize()
        if self.use_sampler_beta:
            if hasattr(self.sampler, "user_defined_eta"):
                self.betas_for_q = torch.tensor(self.sampler.user_defined_eta, dtype=torch.float32)
            elif hasattr(self.sampler, "log_betas"):
                self.betas_for_q = torch.exp(self.sampler.log_betas * 0.5)
            else:   # OpenAIDiffusion
                self.betas_for_q = self.sampler.sigmas[:-1] ** 2
        else:
            self.betas_for_q = make_beta_schedule(schedule=self.q_beta_schedule, n_timesteps=self.n_timesteps,
                                                  start=self.q_beta_start, end=self.q_beta_end)

    def get_running_cost(self, state, next_state, pred_mean, pred_std, t):
        if self.beta_ordering == "reversed":
            t = self.n_timesteps - t - 1
        beta_next = extract(self.betas_for_q, t, state).to(state.device)
        return (((next_state - state) ** 2) / (2 * beta_next)).view(len(state), -1).mean(dim=1)

    def update_adaptive_vel_reg(self, d_sample):
        device = d_sample["sample"].device
        samples = torch.stack(d_sample["l_sample"])
        diff = ((samples[1:] - samples[:-1]) ** 2).view(samples.shape[0] - 1, -1).mean(dim=1)
        if self.beta_ordering == "reversed":
            diff = diff.flip(0)
        if self.betas_for_q.device != device:
            self.betas_for_q = self.betas_for_q.to(device)
        self._set_betas_for_q((self.betas_for_q * self.adavelreg + (1 - self.adavelreg) * diff.to(device)).detach())

    def update_f_v(self, img, d_sample, state_dict, y=None):
        """reference :553-691."""
        g = None
        if isinstance(state_dict, TransitionRing) and d_sample.get("_ring_slot") == (id(state_dict), state_dict.filled - 1) \
                and len(img) == self.batchsize and self.value_update_order != "random":
            nets = [self.v] + ([self.sampler] if self.value_resample else [])
            if y is None:
                fn = lambda im, d=d_sample, sd=state_dict: self._update_f_v(im, d, sd, None)
            else:
                fn = lambda im, yy, d=d_sample, sd=state_dict: self._update_f_v(im, d, sd, yy)
            g = self._step_graph("update_f_v", state_dict, tuple(img.shape) + (y is not None,), fn, nets)
        if g is None:
            keys, vals = self._update_f_v(img, d_sample, state_dict, y)
        else:
            keys, vals = g(img.detach()) if y is None else g(img.detach(), y)
        return dict(zip(keys, vals.tolist()))

    def _update_f_v(self, img, d_sample, state_dict, y=None):
        if self.adavelreg is not None:
            self.update_adaptive_vel_reg(d_sample)
        x0 = d_sample["l_sample"][-1]
        n_steps, batchsize, device = self.n_timesteps, self.batchsize, img.device
        self.optimizer_v.zero_grad()
        _set_mode(self.v, True)
        Tt = n_steps * torch.ones(len(img) + len(x0), dtype=torch.long, device=device)
        ys = torch.cat((y, y), 0) if y is not None else None
        output = self.v(torch.cat((img.detach(), x0.detach()), 0), Tt, y=ys)
        pos_e, neg_e = output[:x0.shape[0]], output[x0.shape[0]:]
        d_loss = pos_e.mean() - neg_e.mean()
        if self.gamma is not None:
            reg = pos_e.pow(2).mean() + neg_e.pow(2).mean()
            d_loss = d_loss + self.gamma * reg
        else:
            reg = torch.zeros((), device=device)
        d_loss.backward()
        self.sync_v()
        self.optimizer_v.step()
        self.optimizer_v.zero_grad()

        permutation = self._randperm(batchsize * n_steps, device)
        indices = permutation + (buffer_rows(state_dict) - batchsize * n_steps)
        rows_all, ts_all = self._td_rows(state_dict, indices, n_steps, batchsize)
        has_y = y is not None and (("y" in state_dict.has) if isinstance(state_dict, TransitionRing)
                                   else len(state_dict["y"]) == len(state_dict["state"]))
        d_running_cost, d_value = {}, {}
        running_cost = v_loss = None
        for _ in range(self.repeat_value_update):
            if self.value_update_order == "random":
                update_order = torch.randperm(n_steps)
            for i in range(n_steps):
                update_t = int(update_order[i]) if self.value_update_order == "random" else n_steps - i - 1
                if self.value_update_order == "shuffle":
                    rows = indices[torch.arange(batchsize, device=device) + i * batchsize]
                    timestep = buffer_gather(state_dict, "timestep", rows)
                else:
                    rows, timestep = rows_all[update_t], ts_all[update_t]
                state = buffer_gather(state_dict, "state", rows)
                yb = buffer_gather(state_dict, "y", rows) if has_y else y
                pred_mean = pred_std = entropy = None
                if self.value_resample:
                    with torch.no_grad():
                        d_step = self.sampler.sample_step(state, timestep, **({"y": yb} if yb is not None else {}))
                    next_state, pred_mean, pred_std = d_step["sample"], d_step["mean"], d_step["sigma"]
                else:
                    next_state = buffer_gather(state_dict, "next_state", rows)
                    if self.entropy_in_value is not None:    # sigma rows are read only by the entropy term
                        pred_std = buffer_gather(state_dict, "sigma", rows)
                running_cost = self.get_running_cost(state, next_state, pred_mean, pred_std, timestep)
                if self.entropy_in_value is not None:
                    entropy = torch.log(pred_std.squeeze() / self.sigma_scale) if self.sigma_scale is not None else torch.log(pred_std.squeeze())
                _set_mode(self.v, False)
                with torch.no_grad():
                    target = self.v(next_state, timestep + 1, y=y).squeeze()
                if self.time_cost is not None:
                    target = target + self.time_cost
                if self.time_cost_sig is not None:
                    center = self.n_timesteps // 2 if self.time_cost_sig_center is None else self.time_cost_sig_center
                    target = target + self.time_cost_sig * torch.sigmoid(-timestep + center) \
                        - self.time_cost_sig * torch.sigmoid(-timestep - 1 + center)
                if self.velocity_in_value is not None:
                    target = target + running_cost * self.tau2 * (timestep < n_steps - self.velocity_in_value).float()
                if self.entropy_in_value is not None:
                    assert isinstance(self.entropy_in_value, int), "self.entropy_in_value should be interger"
                    target = target - entropy * self.tau1 * (timestep < n_steps - self.entropy_in_value).float() * self.entropy_value_scale
                _set_mode(self.v, True)
                v_xt = self.v(state, timestep, y=y).squeeze()
                v_loss = F.mse_loss(v_xt, target.detach())
                v_loss.backward()
                self.sync_v()
                if self.value_grad_clip:
                    self._clip(ops.fast_parameters(self.v), 0.1)
                self.optimizer_v.step()
                self.optimizer_v.zero_grad()
                d_running_cost[f"running_cost/step_{update_t}_"] = running_cost.detach().mean()
                d_value[f"value/step_{update_t}_"] = v_xt.detach().mean()
        logs = {"ebm/d_loss_": d_loss.detach(), "ebm/v_loss_": v_loss.detach(), "ebm/pos_e_": pos_e.detach().mean(),
                "ebm/neg_e_": neg_e.detach().mean(), "ebm/running_cost_": running_cost.detach().mean(), "ebm/reg_": reg.detach()}
        logs.update(d_running_cost)
        logs.update(d_value)
        if self.adavelreg is not None:
            for t, beta in enumerate(self.betas_for_q):
                logs[f"adavelreg/beta{t}_"] = beta
        return self._stack_logs(logs)

    def _guidance_x0_scale(self, x0):          # reference :818-820: EDM trajectories start at sigma_max * N(0, I)
        return x0 * self.sampler.sigma_max

    def _guidance_model_kwargs(self, n_sample, device):   # reference :829-831 ("hard coding": random ImageNet labels)
        if not self.sampler.class_cond:
            return {}
        return {"y": torch.randint(0, self.sampler.num_classes or 1000, (n_sample,), device=device)}

    def update_sampler_mixed_precision(self, state_dict, mp_trainer, d_sample=None):
        """reference :693-746: one optimiser step per `batchsize` slice of ALL buffered transitions."""
        g = None
        if getattr(mp_trainer, "use_fp16", False) and getattr(mp_trainer, "_aliased", False) and hasattr(self.optimizer, "step_sliced_captured"):
            g = self._step_graph("update_sampler_mp", state_dict, (id(mp_trainer),),
                                 lambda sd=state_dict: self._update_sampler_mp(sd, mp_trainer) + (mp_trainer.captured_stats(),),
                                 [self.sampler, self.v])
        if g is None or g.calls < g.warmup:       # eager (also the warm-up call of a graph: optimize() does its own host bookkeeping)
            if g is not None:
                g.calls += 1
            keys, vals = self._update_sampler_mp(state_dict, mp_trainer)
            return dict(zip(keys, vals.tolist()))
        keys, vals, stats = g()
        back = torch.cat([vals, stats.reshape(-1)]).tolist()            # logs + the K x (grad norm, param norm, overflow flag): one read-back
        mp_trainer.finish_replay(self.optimizer, [back[len(keys) + 3 * i:len(keys) + 3 * i + 3] for i in range(stats.shape[0])])
        return dict(zip(keys, back[:len(keys)]))

    def _update_sampler_mp(self, state_dict, mp_trainer):
        _set_mode(self.v, False)
        _set_mode(self.sampler, True)
        device = state_dict.device if isinstance(state_dict, TransitionRing) else state_dict["state"].device
        permutation = self._randperm(buffer_rows(state_dict), device)
        batchsize = self.batchsize
        if _graph.capturing():
            mp_trainer.begin_captured(self.optimizer)
        for m in range(0, len(permutation), batchsize):
            mp_trainer.zero_grad()
            indices = permutation[m:m + batchsize]
            state = buffer_gather(state_dict, "state", indices)
            t = buffer_gather(state_dict, "timestep", indices)
            y = buffer_gather(state_dict, "y", indices) if self.sampler.class_cond else None
            d_step = self.sampler.sample_step(state, t, **({"y": y} if y is not None else {}))
            next_state, pred_mean, pred_std = d_step["sample"], d_step["mean"], d_step["sigma"]
            running_cost = self.get_running_cost(state, next_state, pred_mean, pred_std, t)
            causal_entropy = torch.log(pred_std.squeeze())
            with self._frozen(self.v):
                sampler_value_loss = self.v(next_state, t + 1, y=y).squeeze()
            non_terminal = (t < self.n_timesteps - self.skip_sampler_tau).float()
            sampler_loss = (sampler_value_loss + (running_cost * self.tau2 - causal_entropy * self.tau1) * non_terminal).mean()
            mp_trainer.backward(sampler_loss)
            mp_trainer.optimize(self.optimizer)
        logs = {"sampler/sampler_loss_": sampler_loss.detach().mean(), "sampler/sampler_value_loss_": sampler_value_loss.detach().mean(),
                "sampler/running_cost_": running_cost.detach().mean(), "sampler/causal_entropy_": causal_entropy.detach().mean()}
        if self.sampler.trainable_beta:
            net = self.sampler.net.module if hasattr(self.sampler.net, "module") else self.sampler.net
            sigma = torch.exp(net.log_betas.detach())
            for t in range(len(sigma)):
                logs[f"sigma/sigma_{t}_"] = sigma[t]
        return self._stack_logs(logs)


class DxMI_Trainer_EV(DxMI_Trainer):
    """Separate energy `f(x)` and time-dependent value `v(x, t)` (reference :865-1078; no config of the snapshot selects it).
    The reference's arithmetic on the HIP modules: contrastive energy step on `f` with its gradient clipped at 0.1, T TD steps
    whose transition is RE-DRAWN with `sampler.sample_step` (:976-978) and whose target is `v(x', t + 1)` for non-terminal
    steps and `f(x')` for the last one, plus `tau2 * running_cost - tau1 * log sigma` (:985-987); policy step on one random
    minibatch of buffered states per `n_generator` with the same mixed terminal value (:1037-1060).  Same helpers as
    DxMI_Trainer: device-side index path (one stable sort instead of T `nonzero`s on the ring), device-side clip, one log sync,
    flat RCCL gradient exchange; `f` and `v` are frozen inside the policy step (the reference accumulates and discards their
    gradients)."""

    def __init__(self, batchsize, tau1=0.0, tau2=0.0, q_beta_schedule="constant", q_beta_start=1.0, q_beta_end=1.0, adavelreg=None,
                 n_timesteps=10, use_sampler_beta=False):
        self.batchsize, self.n_timesteps = batchsize, n_timesteps
        self.tau1, self.tau2 = tau1, tau2
        self.q_beta_schedule, self.q_beta_start, self.q_beta_end = q_beta_schedule, q_beta_start, q_beta_end
        self.adavelreg, self.use_sampler_beta = adavelreg, use_sampler_beta

    def set_models(self, v, sampler, optimizer, optimizer_v, f=None, optimizer_fstar=None):
        """reference :899-919 (note the argument order: v first)."""
        self.f, self.v, self.sampler = f, v, sampler
        self.optimizer, self.optimizer_fstar, self.optimizer_v = optimizer, optimizer_fstar, optimizer_v
        from dxmi_hip.dist import FlatGradSync
        self.sync_v = FlatGradSync(v) if v is not None else (lambda: None)
        self.sync_f = FlatGradSync(f) if f is not None else (lambda: None)
        self.sync_sampler = FlatGradSync(sampler) if sampler is not None else (lambda: None)
        if self.use_sampler_beta:
            net = self.sampler.net.module if hasattr(self.sampler.net, "module") else self.sampler.net
            if hasattr(net, "log_betas"):
                self.betas_for_q = torch.exp(net.log_betas).detach()
        else:
            self.betas_for_q = make_beta_schedule(schedule=self.q_beta_schedule, n_timesteps=self.n_timesteps,
                                                  start=self.q_beta_start, end=self.q_beta_end)

    def get_running_cost(self, state, next_state, t):           # reference :921-927
        return DxMI_Trainer.get_running_cost(self, state, next_state, None, None, t)

    def _terminal_mix(self, next_state, t):
        """v(x', t + 1) where the step is not the last, f(x') where it is (:985-986, :1049-1050)."""
        non_terminal = (t < self.n_timesteps - 1).float()
        return self.v(next_state, t + 1).squeeze() * non_terminal + self.f(next_state).squeeze() * (1 - non_terminal)

    def update_f_v(self, img, d_sample, state_dict):
        """reference :945-1028."""
        if self.adavelreg is not None:
            self.update_adaptive_vel_reg(d_sample)
        x0 = d_sample["l_sample"][-1]
        n_steps, batchsize, device = self.n_timesteps, self.batchsize, img.device
        self.optimizer_fstar.zero_grad()
        _set_mode(self.f, True)
        output = self.f(torch.cat((img.detach(), x0.detach()), 0))
        pos_e, neg_e = output[:x0.shape[0]], output[x0.shape[0]:]
        d_loss = pos_e.mean() - neg_e.mean()
        d_loss.backward()
        self.sync_f()
        self._clip(ops.fast_parameters(self.f), 0.1)
        self.optimizer_fstar.step()
        self.optimizer_fstar.zero_grad()
        _set_mode(self.f, False)
        self.optimizer_v.zero_grad()
        permutation = torch.randperm(batchsize * n_steps)                     # CPU generator, as the reference
        indices = (permutation + (buffer_rows(state_dict) - batchsize * n_steps)).to(device)
        rows_all, ts_all = self._td_rows(state_dict, indices, n_steps, batchsize)
        d_running_cost = {}
        running_cost = v_loss = None
        for i in range(n_steps):
            update_t = n_steps - i - 1
            rows, timestep = rows_all[update_t], ts_all[update_t]
            state = buffer_gather(state_dict, "state", rows)
            with torch.no_grad():                                              # the target is detached (:990): result-neutral
                d_step = self.sampler.sample_step(state, timestep)
                next_state, pred_std = d_step["sample"], d_step["sigma"]
                running_cost = self.get_running_cost(state, next_state, timestep)
                entropy = torch.log(pred_std.squeeze())
                _set_mode(self.v, False)
                target = self._terminal_mix(next_state, timestep) + running_cost * self.tau2 - entropy * self.tau1
            _set_mode(self.v, True)
            v_xt = self.v(state, timestep).squeeze()
            v_loss = F.mse_loss(v_xt, target.detach())
            v_loss.backward()
            self.sync_v()
            self.optimizer_v.step()
            self.optimizer_v.zero_grad()
            d_running_cost[f"running_cost/step_{update_t}_"] = running_cost.detach().mean()
        logs = {"ebm/d_loss_": d_loss.detach(), "ebm/v_loss_": v_loss.detach(), "ebm/pos_e_": pos_e.detach().mean(),
                "ebm/neg_e_": neg_e.detach().mean(), "ebm/running_cost_": running_cost.detach().mean()}
        logs.update(d_running_cost)
        if self.adavelreg is not None:
            for t, beta in enumerate(self.betas_for_q):
                logs[f"adavelreg/beta_for_q_{t}_"] = beta
        return self._to_floats(logs)

    def update_sampler(self, state_dict, n_generator):
        """reference :1030-1078."""
        _set_mode(self.f, False)
        _set_mode(self.v, False)
        _set_mode(self.sampler, True)
        permutation = torch.randperm(buffer_rows(state_dict))
        batchsize = self.batchsize
        n_data = min(len(permutation), batchsize * n_generator)
        device = state_dict.device if isinstance(state_dict, TransitionRing) else state_dict["state"].device
        for m in range(0, n_data, batchsize):
            self.optimizer.zero_grad()
            indices = permutation[m:m + batchsize].to(device)
            state = buffer_gather(state_dict, "state", indices)
            t = buffer_gather(state_dict, "timestep", indices)
            d_step = self.sampler.sample_step(state, t)
            next_state, pred_std = d_step["sample"], d_step["sigma"]
            running_cost = self.get_running_cost(state, next_state, t).mean()
            causal_entropy = torch.log(pred_std.squeeze()).mean()
            with self._frozen(self.v), self._frozen(self.f):
                sampler_value_loss = self._terminal_mix(next_state, t).mean()
            sampler_loss = sampler_value_loss + running_cost * self.tau2 - causal_entropy * self.tau1
            sampler_loss.backward()
            self.sync_sampler()
            self._clip(ops.fast_parameters(self.sampler), 0.1)
            self.optimizer.step()
        logs = {"sampler/sampler_loss_": sampler_loss.detach(), "sampler/sampler_value_loss_": sampler_value_loss.detach(),
                "sampler/running_cost_": running_cost.detach(), "sampler/causal_entropy_": causal_entropy.detach()}
        if self.sampler.trainable_beta:
            net = self.sampler.net.module if hasattr(self.sampler.net, "module") else self.sampler.net
            betas = torch.exp(net.log_betas.detach())
            for t in range(len(betas)):
                logs[f"beta/beta_{t}_"] = betas[t]
        return self._to_floats(logs)
