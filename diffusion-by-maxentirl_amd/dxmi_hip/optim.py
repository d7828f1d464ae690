"""Optimisers of the DxMI train step on the multi-tensor HIP kernels (csrc/optim.hip).

`Adam` / `RAdam` subclass the torch optimisers the reference constructs (train_cifar10.py:283-296,
train_image_large.py:153-160): same constructor, same `param_groups`, same per-parameter state
(`step`, `exp_avg`, `exp_avg_sq`) and therefore interchangeable `state_dict()`s.  Only `step()` changes: one
fused kernel series over all tensors (6 launches for the 330 U-Net tensors) instead of torch's ~10 foreach
passes, with the arithmetic of torch's own implementation (see include/dxmi_hip.h).  Options the kernels do not
implement (amsgrad, weight decay, maximize, capturable, differentiable) fall back to LOUD errors, not to torch.
"""
import math

import torch

from . import graph as _graph
from . import ops
from ._lib import DxmiError


def _check_group(group, name):
    for key in ("amsgrad", "maximize", "capturable", "differentiable"):
        if group.get(key):
            raise DxmiError(f"dxmi_hip.optim.{name}: option {key} is not implemented by the HIP kernel")
    if group.get("weight_decay", 0) != 0:
        raise DxmiError(f"dxmi_hip.optim.{name}: weight_decay != 0 is not implemented by the HIP kernel")


class _FusedBase:
    def _gather(self):
        """-> list of (group, [params with grad]) and per-parameter state, created as torch does."""
        out = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            out.append((group, ps))
        return out


def _ptr_key(params, ms, vs):
    """Identity of the storages the cached pointer arrays describe: load_state_dict() replaces the moment tensors and
    `p.data = ...` / `.to()` replace a parameter's storage while the python objects (and their ids) stay the same."""
    return tuple(t.data_ptr() for ts in (params, ms, vs) for t in ts)


def _bump_versions(params):
    """The kernels write parameters through raw pointers, which autograd's version counters do not see, while every
    packed bf16 weight cache (Model._param_key, modules._pack_t ...) is keyed on (data_ptr, _version): without the bump the
    nets would keep running on the weights of the first pack.  No kernel is launched."""
    torch.autograd.graph.increment_version(params)


class _Batch:
    """Tensors of one fused launch series: equal betas / eps / step count (bias corrections are per launch)."""

    __slots__ = ("params", "grads", "ms", "vs", "lrs", "groups")

    def __init__(self):
        self.params, self.grads, self.ms, self.vs, self.lrs, self.groups = [], [], [], [], [], []


class _BatchView:
    __slots__ = ("params", "ms", "vs")

    def __init__(self, params, ms, vs):
        self.params, self.ms, self.vs = params, ms, vs


def _batches(opt, name):
    """(betas, eps, step) -> _Batch.  Parameters that received their first gradient later than the others (or only
    intermittently) have a lagging step counter; torch.optim handles them per tensor, here they form their own launch."""
    out = {}
    for group, ps in opt._gather():
        if not ps:
            continue
        _check_group(group, name)
        if name == "RAdam" and group.get("decoupled_weight_decay"):
            raise DxmiError("dxmi_hip.optim.RAdam: decoupled_weight_decay is not implemented")
        b1, b2 = group["betas"]
        for p in ps:
            st = opt.state[p]
            st["step"] += 1
            b = out.setdefault((b1, b2, group["eps"], float(st["step"])), _Batch())
            b.params.append(p)
            b.grads.append(p.grad)
            b.ms.append(st["exp_avg"])
            b.vs.append(st["exp_avg_sq"])
            b.lrs.append(group["lr"])
            b.groups.append(group)
    return out


def _replay_counters(opt, b):
    """Host side of a captured optimiser step at REPLAY time: the step counters of the batch's tensors advance as `_batches()`
    advances them in an eager step (the capture pass itself has already been counted there); returns the new count."""
    steps = [opt.state[p]["step"] for p in b.params]
    first = [True]

    def advance():
        if first[0]:
            first[0] = False
        else:
            torch._foreach_add_(steps, 1)
        return float(steps[0])
    return advance


def _cache_for(opt, key, b):
    caches = opt.__dict__.setdefault("_dxmi_cache", {})
    # (the first parameter's address tells apart two batches of equal length: parameters whose step counts differ)
    cache = caches.setdefault(key[:3] + (len(b.params), b.params[0].data_ptr()), {})
    pk = _ptr_key(b.params, b.ms, b.vs)
    if cache.get("ptr_key") != pk:
        cache.clear()
        cache["ptr_key"] = pk
    return cache


class Adam(_FusedBase, torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: optional device scalar multiplied into every gradient (e.g. gradnorm_clip(...)[1:2])."""
        assert closure is None
        cap = _graph.current()
        for key, b in _batches(self, "Adam").items():
            b1, b2, eps, t = key
            if cap is None:
                steps = [-(lr / (1 - b1 ** t)) for lr in b.lrs]            # step_size, in double as torch forms it
                ops.adam_step(b.params, b.grads, b.ms, b.vs, steps, b1, b2, eps, (1 - b2 ** t) ** 0.5, grad_scale=grad_scale,
                              cache=_cache_for(self, key, b))
            else:
                # captured into a hipGraph (dxmi_hip/graph.py): the step-dependent scalars are a HOST INPUT of the graph — formed
                # here, per replay, by the same double arithmetic (the learning rates are read live: schedulers keep working)
                advance, groups = _replay_counters(self, b), list(b.groups)

                def hyper_values(advance=advance, groups=groups, b1=b1, b2=b2):
                    tt = advance()
                    return [(1 - b2 ** tt) ** 0.5] + [-(g["lr"] / (1 - b1 ** tt)) for g in groups]
                hyper = cap.host_input(torch.float32, len(b.params) + 1, hyper_values)
                ops.adam_step(b.params, b.grads, b.ms, b.vs, None, b1, b2, eps, None, grad_scale=grad_scale,
                              cache=_cache_for(self, key, b), hyper=hyper)
            _bump_versions(b.params)
        return None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.pop("_dxmi_cache", None)


class RAdam(_FusedBase, torch.optim.RAdam):
    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, found_inf=None):
        """grad_scale / found_inf: optional device scalars (loss-scale reciprocal, overflow flag: a flagged step is skipped
        on the device; the caller rolls `step` back with `rollback_step()` once it reads the flag)."""
        assert closure is None
        self._dxmi_last_stepped = [self.state[p] for g in self.param_groups for p in g["params"] if p.grad is not None]
        for key, b in _batches(self, "RAdam").items():
            b1, b2, eps, t = key
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            rho_inf = 2 / (1 - b2) - 1
            rho_t = rho_inf - 2 * t * (b2 ** t) / bc2
            rect = -1.0
            if rho_t > 5.0:
                rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
            ops.radam_step(b.params, b.grads, b.ms, b.vs, b.lrs, b1, b2, eps, bc1, math.sqrt(bc2), rect, grad_scale=grad_scale,
                           found_inf=found_inf, cache=_cache_for(self, key, b))
            _bump_versions(b.params)
        return None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.pop("_dxmi_cache", None)
        self.__dict__.pop("_dxmi_slices", None)

    @staticmethod
    def radam_scalars(b1, b2, t):
        """(fp32(1 / bc1), bc2_sqrt, rect | -1) of step count t, formed in double as torch/optim/radam.py forms them."""
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        rho_inf = 2 / (1 - b2) - 1
        rho_t = rho_inf - 2 * t * (b2 ** t) / bc2
        rect = -1.0
        if rho_t > 5.0:
            rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
        return 1.0 / bc1, math.sqrt(bc2), rect

    @torch.no_grad()
    def step_sliced_captured(self, cap, slices, hyper3, grad_scale, found_inf):
        """step_sliced() of a step that is being captured into a hipGraph (MixedPrecisionTrainer under dxmi_hip/graph.py).  The
        step count of a replayed iteration depends on how many earlier iterations of the same replay overflowed, which only the
        device knows: the CALLER selects the row of step-dependent scalars on the device (hyper3 = fp32 [3]: 1/bc1, bc2_sqrt,
        rect) and advances the host counters after the replay (`advance_steps`).  The learning rates are a host input of the
        graph (read live per replay)."""
        ps_all, ms, vs, groups = [], [], [], []
        cache = self.__dict__.setdefault("_dxmi_slices", {})
        betas = eps = None
        for group in self.param_groups:
            _check_group(group, "RAdam")
            if betas is None:
                betas, eps = tuple(group["betas"]), group["eps"]
            if tuple(group["betas"]) != betas or group["eps"] != eps:
                raise DxmiError("RAdam.step_sliced_captured: parameter groups with different betas / eps")
            for P in group["params"]:
                st = self.state[P]
                if len(st) == 0:
                    raise DxmiError("RAdam.step_sliced_captured: optimiser state missing (run one eager step first)")
                views = cache.get(id(P))
                if views is None:
                    raise DxmiError("RAdam.step_sliced_captured: slice views missing (run one eager step first)")
                ps = slices[P]
                ps_all += ps
                ms += views[1]
                vs += views[2]
                groups += [group] * len(ps)
        n = len(ps_all)
        lrs = cap.host_input(torch.float32, n, lambda: [g["lr"] for g in groups])
        hyper = torch.cat([hyper3.reshape(3), lrs])
        ops.radam_step(ps_all, [q.grad for q in ps_all], ms, vs, None, betas[0], betas[1], eps, None, None, None, grad_scale=grad_scale,
                       found_inf=found_inf, cache=_cache_for(self, (betas[0], betas[1], eps, "captured"), _BatchView(ps_all, ms, vs)), hyper=hyper)
        _bump_versions(ps_all)

    def step_count(self):
        """Common step count of the flat master parameters (they step together)."""
        counts = {float(self.state[P]["step"]) for g in self.param_groups for P in g["params"] if len(self.state[P])}
        if len(counts) != 1:
            raise DxmiError(f"RAdam: parameters at different step counts {sorted(counts)}")
        return counts.pop()

    def advance_steps(self, n):
        torch._foreach_add_([self.state[P]["step"] for g in self.param_groups for P in g["params"]], n)

    @torch.no_grad()
    def step_sliced(self, slices, grad_scale=None, found_inf=None):
        """One RAdam step of FLAT master parameters taken tensor by tensor (MixedPrecisionTrainer, models/cm/fp16_util.py:
        204-223): `slices[master] = [model_param, ...]` lists, in flattening order, the model parameters whose storage IS the
        corresponding slice of `master` (the trainer aliases them), each carrying its own `.grad`.  The kernels then read the
        model gradients where autograd left them and update parameter and moment slices in place: no flattened gradient
        copy, no `grad.mul_(1 / scale)` pass (grad_scale), no copy back into the model.  Same arithmetic, state layout
        (`state[master]`: step, flat exp_avg / exp_avg_sq) and step counting as `step()`."""
        cache = self.__dict__.setdefault("_dxmi_slices", {})
        self._dxmi_last_stepped = [self.state[P] for g in self.param_groups for P in g["params"]]
        batches = {}
        for group in self.param_groups:
            _check_group(group, "RAdam")
            if group.get("decoupled_weight_decay"):
                raise DxmiError("dxmi_hip.optim.RAdam: decoupled_weight_decay is not implemented")
            b1, b2 = group["betas"]
            for P in group["params"]:
                ps = slices[P]
                st = self.state[P]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(P, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(P, memory_format=torch.preserve_format)
                st["step"] += 1
                ck = (id(P), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), P.data_ptr(), len(ps))
                views = cache.get(id(P))
                if views is None or views[0] != ck:
                    mf, vf, off, ms, vs = st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), 0, [], []
                    for q in ps:
                        if q.data_ptr() != P.data_ptr() + 4 * off:
                            raise DxmiError("dxmi_hip.optim.RAdam.step_sliced: a model parameter is not the slice of its master")
                        ms.append(mf[off:off + q.numel()])
                        vs.append(vf[off:off + q.numel()])
                        off += q.numel()
                    if off != P.numel():
                        raise DxmiError("dxmi_hip.optim.RAdam.step_sliced: slices do not cover the master parameter")
                    views = cache[id(P)] = (ck, ms, vs)
                b = batches.setdefault((b1, b2, group["eps"], float(st["step"])), _Batch())
                b.params += ps
                b.grads += [q.grad for q in ps]
                b.ms += views[1]
                b.vs += views[2]
                b.lrs += [group["lr"]] * len(ps)
        for key, b in batches.items():
            b1, b2, eps, t = key
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            rho_inf = 2 / (1 - b2) - 1
            rho_t = rho_inf - 2 * t * (b2 ** t) / bc2
            rect = -1.0
            if rho_t > 5.0:
                rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
            ops.radam_step(b.params, b.grads, b.ms, b.vs, b.lrs, b1, b2, eps, bc1, math.sqrt(bc2), rect, grad_scale=grad_scale,
                           found_inf=found_inf, cache=_cache_for(self, key + ("sliced",), b))
            _bump_versions(b.params)
        return None

    def rollback_step(self):
        """Undo the step counter of a step the device skipped (found_inf was set): only of the parameters that took part in the
        last step()/step_sliced() (a parameter whose gradient was None kept its count)."""
        for st in self.__dict__.get("_dxmi_last_stepped", ()):
            st["step"] -= 1
