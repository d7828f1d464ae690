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


class Adam(_FusedBase, torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: optional device scalar multiplied into every gradient (e.g. gradnorm_clip(...)[1:2])."""
        assert closure is None
        params, grads, ms, vs, steps = [], [], [], [], []
        key = None
        for group, ps in self._gather():
            if not ps:
                continue
            _check_group(group, "Adam")
            b1, b2 = group["betas"]
            k = (b1, b2, group["eps"])
            for p in ps:
                st = self.state[p]
                st["step"] += 1
                t = float(st["step"])
                if key is None:
                    key = k + (t,)
                if k + (t,) != key:
                    raise DxmiError("dxmi_hip.optim.Adam: betas / eps / step count must agree across parameter groups")
                params.append(p)
                grads.append(p.grad)
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
                steps.append(-(group["lr"] / (1 - b1 ** t)))          # step_size, in double as torch forms it
        if not params:
            return None
        b1, b2, eps, t = key
        cache = self.__dict__.setdefault("_dxmi_cache", {})
        if cache.get("ids") != [id(p) for p in params]:
            cache.clear()
            cache["ids"] = [id(p) for p in params]
        ops.adam_step(params, grads, ms, vs, steps, b1, b2, eps, (1 - b2 ** t) ** 0.5, grad_scale=grad_scale, cache=cache)
        return None


class RAdam(_FusedBase, torch.optim.RAdam):
    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, found_inf=None):
        """grad_scale / found_inf: optional device scalars (loss-scale reciprocal, overflow flag: a flagged step is skipped
        on the device; the caller rolls `step` back with `rollback_step()` once it reads the flag)."""
        assert closure is None
        params, grads, ms, vs, lrs = [], [], [], [], []
        key = None
        for group, ps in self._gather():
            if not ps:
                continue
            _check_group(group, "RAdam")
            if group.get("decoupled_weight_decay"):
                raise DxmiError("dxmi_hip.optim.RAdam: decoupled_weight_decay is not implemented")
            b1, b2 = group["betas"]
            k = (b1, b2, group["eps"])
            for p in ps:
                st = self.state[p]
                st["step"] += 1
                t = float(st["step"])
                if key is None:
                    key = k + (t,)
                if k + (t,) != key:
                    raise DxmiError("dxmi_hip.optim.RAdam: betas / eps / step count must agree across parameter groups")
                params.append(p)
                grads.append(p.grad)
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
                lrs.append(group["lr"])
        if not params:
            return None
        b1, b2, eps, t = key
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        rho_inf = 2 / (1 - b2) - 1
        rho_t = rho_inf - 2 * t * (b2 ** t) / bc2
        rect = -1.0
        if rho_t > 5.0:
            rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
        cache = self.__dict__.setdefault("_dxmi_cache", {})
        if cache.get("ids") != [id(p) for p in params]:
            cache.clear()
            cache["ids"] = [id(p) for p in params]
        ops.radam_step(params, grads, ms, vs, lrs, b1, b2, eps, bc1, math.sqrt(bc2), rect, grad_scale=grad_scale,
                       found_inf=found_inf, cache=cache)
        return None

    def rollback_step(self):
        """Undo the step counter of a step the device skipped (found_inf was set)."""
        for st in self.state.values():
            if "step" in st:
                st["step"] -= 1
