"""`models.cm.fp16_util.MixedPrecisionTrainer` with the reference's surface (reference: models/cm/fp16_util.py:35-248),
as train_image_large.py:157-170 and DxMI_Trainer_Cond.update_sampler_mixed_precision use it.

On the HIP path the model parameters ARE fp32 masters (kernels read bf16 fragments re-packed from them), so
`use_fp16=True` keeps the reference's bookkeeping — flat fp32 master tensors in the reference's three groups
(`special_key` vector, other <=1-D parameters, matrices viewed (1,-1)), loss scaled by 2**lg_loss_scale, overflow check,
scale growth — while the "model" side it copies to and from is fp32 too.  bf16 gradients have the fp32 exponent range, so
the scaling is exact and never the reason for an overflow; it is kept so `lg_loss_scale` and the skip-on-NaN behaviour
match.  Gradients are all-reduced over ranks (RCCL) in `optimize()` before the norms are taken: the reference gets the
same from its DDP wrapper during backward (train_image_large.py:173).
"""
import numpy as np
import torch
import torch.nn as nn

from dxmi_hip import graph as _graph
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

INITIAL_LOG_LOSS_SCALE = 20.0


def get_param_groups_and_shapes(named_model_params, special_key=None):
    named_model_params = list(named_model_params)
    groups = []
    if special_key is not None:
        groups.append(([(n, p) for (n, p) in named_model_params if special_key in n], (-1)))
        named_model_params = [(n, p) for (n, p) in named_model_params if special_key not in n]
    groups.append(([(n, p) for (n, p) in named_model_params if p.ndim <= 1], (-1)))
    groups.append(([(n, p) for (n, p) in named_model_params if p.ndim > 1], (1, -1)))
    return groups


def make_master_params(param_groups_and_shapes):
    master_params = []
    for param_group, shape in param_groups_and_shapes:
        mp = nn.Parameter(_flatten_dense_tensors([p.detach().float() for (_, p) in param_group]).view(shape))
        mp.requires_grad = True
        master_params.append(mp)
    return master_params


def param_grad_or_zeros(param):
    return param.grad.data.detach() if param.grad is not None else torch.zeros_like(param)


def model_grads_to_master_grads(param_groups_and_shapes, master_params):
    for master_param, (param_group, shape) in zip(master_params, param_groups_and_shapes):
        master_param.grad = _flatten_dense_tensors([param_grad_or_zeros(p).float() for (_, p) in param_group]).view(shape)


def unflatten_master_params(param_group, master_param):
    return _unflatten_dense_tensors(master_param, [p for (_, p) in param_group])


def master_params_to_model_params(param_groups_and_shapes, master_params):
    """One multi-tensor copy per group (the reference loops `param.detach().copy_()` over ~600 tensors per step)."""
    for master_param, (param_group, _) in zip(master_params, param_groups_and_shapes):
        if not param_group:
            continue
        if param_group[0][1].data_ptr() == master_param.data_ptr():     # the model parameters ARE the master's slices:
            # nothing to copy, but the packed-weight caches are keyed on the model tensors' version counters
            torch.autograd.graph.increment_version([param for (_, param) in param_group])
            continue
        dst = [param.detach() for (_, param) in param_group]
        src = list(unflatten_master_params(param_group, master_param.detach().view(-1)))
        torch._foreach_copy_(dst, src)


def zero_master_grads(master_params):
    for param in master_params:
        param.grad = None


def zero_grad(model_params):
    """Drop the gradients instead of zero-filling them (the reference fills ~600 tensors, then autograd ADDS the new
    gradient into each): every backward here is preceded by a zero_grad, so assigning is the same value with neither
    the fill nor the add kernels; param_grad_or_zeros() covers parameters that received no gradient."""
    for param in model_params:
        param.grad = None


def check_overflow(value):
    return (value == float("inf")) or (value == -float("inf")) or (value != value)


class MixedPrecisionTrainer:
    def __init__(self, *, model, use_fp16=False, fp16_scale_growth=1e-3, initial_lg_loss_scale=INITIAL_LOG_LOSS_SCALE,
                 special_key=None):
        self.model, self.use_fp16, self.fp16_scale_growth = model, use_fp16, fp16_scale_growth
        self.model_params = list(self.model.parameters())
        self.master_params = self.model_params
        self.param_groups_and_shapes = None
        self.lg_loss_scale = initial_lg_loss_scale
        self.log = {}     # last grad_norm / param_norm / lg_loss_scale (the reference sends these to its logger)
        self._aliased = False
        if self.use_fp16:
            self.param_groups_and_shapes = get_param_groups_and_shapes(self.model.named_parameters(), special_key=special_key)
            self.master_params = make_master_params(self.param_groups_and_shapes)
            self._alias_model_params()
        from dxmi_hip.dist import FlatGradSync
        self._sync = FlatGradSync(model)

    def _alias_model_params(self):
        """On the HIP path the model parameters are fp32 like their masters, so the two copies the reference keeps (fp16 model,
        fp32 flat masters: fp16_util.py:96-115) can be ONE storage: every model parameter becomes the view of its slice of the
        flat master tensor.  `master_params_to_model_params` then has nothing to copy, and `optimize` can step the masters
        tensor by tensor on the model gradients (dxmi_hip.optim.RAdam.step_sliced).  Device tensors only (the CPU tests keep
        the reference's two-copy bookkeeping)."""
        if not all(p.is_cuda and p.dtype == torch.float32 for p in self.model_params):
            return
        self._slices = {}
        for master, (group, _) in zip(self.master_params, self.param_groups_and_shapes):
            flat, off, ps = master.detach().view(-1), 0, []
            for _, p in group:
                p.data = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                ps.append(p)
            self._slices[master] = ps
        torch.autograd.graph.increment_version(self.model_params)      # packed-weight caches are keyed on (data_ptr, _version)
        self._aliased = True

    def zero_grad(self):
        zero_grad(self.model_params)

    def backward(self, loss):
        cap = _graph.current()
        if cap is not None:
            return self._backward_captured(cap, loss)
        if self.use_fp16:
            (loss * 2 ** self.lg_loss_scale).backward()
        else:
            loss.backward()

    def optimize(self, opt):
        self._sync()   # data-parallel mean of the gradients (no-op on one process; a graph cut inside a captured step)
        if _graph.capturing():
            return self._optimize_captured(opt)
        return self._optimize_fp16(opt) if self.use_fp16 else self._optimize_normal(opt)

    # ------------------------------------------------------------------ captured into a hipGraph (dxmi_hip/graph.py)
    # A replayed update runs K backward / optimize iterations without the host in between, so the loss-scale bookkeeping of
    # optimize() — skip the step, roll the step counter back and halve the scale on overflow; grow the scale otherwise
    # (reference fp16_util.py:204-231) — moves to the device: the number of overflows so far in this replay is a device counter
    # `ovf`, and every iteration gets, as a host input, the table of (loss scale, 1 / scale, 1 / bc1, bc2_sqrt, rect) for EVERY
    # possible value j = 0 .. i of that counter (step count t0 + i - j + 1, lg after i - j growths and j halvings, all formed on
    # the host in double exactly as the eager path forms them); the device picks row `ovf`.  The RAdam kernel skips on the
    # device flag as before.  After the replay `finish_replay` reads the K flags and norms back ONCE and repeats the eager
    # bookkeeping on the host (lg_loss_scale, step counters, logged norms).
    MAX_CAPTURED_ITERS = 64

    def _capture_state(self, cap):
        st = self.__dict__.get("_cap_state")
        if st is None or st["cap"] is not cap or st["capture_id"] != cap.captures:
            if not (self.use_fp16 and self._aliased):
                raise NotImplementedError("MixedPrecisionTrainer inside a StepGraph capture: use_fp16 with device-aliased masters only")
            dev = self.model_params[0].device
            st = {"cap": cap, "capture_id": cap.captures, "i": 0, "ovf": torch.zeros(1, dtype=torch.int64, device=dev),
                  "stats": torch.zeros((self.MAX_CAPTURED_ITERS, 3), dtype=torch.float32, device=dev), "row": None, "opt": None}
            self.__dict__["_cap_state"] = st
        return st

    def _rows(self, i, opt):
        """Host input of iteration i: rows j = 0 .. i (overflows so far in this replay)."""
        from dxmi_hip.optim import RAdam
        lg0, t0 = self.lg_loss_scale, opt.step_count()
        b1, b2 = opt.param_groups[0]["betas"]
        out = []
        for j in range(i + 1):
            lg = lg0
            for _ in range(j):
                lg -= 1
            for _ in range(i - j):
                lg += self.fp16_scale_growth
            scale = 2.0 ** lg
            inv_bc1, bc2_sqrt, rect = RAdam.radam_scalars(b1, b2, t0 + (i - j) + 1)
            out.append([scale, 1.0 / scale, inv_bc1, bc2_sqrt, rect])
        return np.asarray(out, dtype=np.float64).astype(np.float32).reshape(-1)

    def _backward_captured(self, cap, loss):
        st = self._capture_state(cap)
        i = st["i"]
        if i >= self.MAX_CAPTURED_ITERS:
            raise RuntimeError("MixedPrecisionTrainer: more than MAX_CAPTURED_ITERS optimiser iterations in one captured step")
        if st["opt"] is None:
            raise RuntimeError("MixedPrecisionTrainer inside a StepGraph capture: call begin_captured(opt) first")
        opt = st["opt"]
        table = cap.host_input(torch.float32, (i + 1) * 5, lambda i=i, opt=opt: self._rows(i, opt)).view(i + 1, 5)
        st["row"] = table.index_select(0, st["ovf"]).reshape(5)
        (loss * st["row"][0]).backward()

    def begin_captured(self, opt):
        """Called by the trainer at the top of a captured update: binds the optimiser whose step count the host tables follow."""
        st = self._capture_state(_graph.current())
        st["opt"] = opt

    def _optimize_captured(self, opt):
        from dxmi_hip import ops
        st = self._capture_state(_graph.current())
        assert st["opt"] is opt and st["row"] is not None
        for p in self.model_params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        row = st["row"]
        gstat = ops.gradnorm_clip([p.grad for p in self.model_params], 0.0)
        pstat = ops.gradnorm_clip([m.detach() for m in self.master_params], 0.0)
        found = gstat[2:3]
        opt.step_sliced_captured(st["cap"], self._slices, row[2:5], row[1:2], found)
        st["stats"][st["i"]] = torch.stack([gstat[0], pstat[0], gstat[2]])
        st["ovf"] += (found != 0).to(torch.int64)
        zero_grad(self.model_params)
        st["i"] += 1
        st["row"] = None
        return True

    def captured_stats(self):
        """Device [K, 3] (scaled gradient norm, parameter norm, overflow flag) of the K iterations of the captured update."""
        st = self.__dict__["_cap_state"]
        return st["stats"][:st["i"]]

    def finish_replay(self, opt, stats):
        """stats: captured_stats() read back ([K][3] python floats).  The eager bookkeeping of optimize(), iteration by iteration."""
        ok = 0
        for gn, pn, flag in stats:
            self.log["lg_loss_scale"] = self.lg_loss_scale
            scale = 2.0 ** self.lg_loss_scale
            if flag != 0.0 or check_overflow(gn / scale):
                self.lg_loss_scale -= 1
            else:
                self.log["grad_norm"], self.log["param_norm"] = gn / scale, pn
                self.lg_loss_scale += self.fp16_scale_growth
                ok += 1
        opt.advance_steps(ok)

    def _optimize_fp16_sliced(self, opt):
        """`_optimize_fp16` without its five passes over the 1.2 GB of parameters (flatten the gradients, two norm passes with a
        host sync each, `grad.mul_(1 / scale)`, copy the masters back): the norm kernel reads the model gradients where
        autograd left them, the overflow flag stays on the device and the RAdam kernel — skipped there when the flag is set —
        un-scales the gradient as it reads it.  One host read (norms + flag) per call, for the loss-scale bookkeeping."""
        from dxmi_hip import ops
        self.log["lg_loss_scale"] = self.lg_loss_scale
        for p in self.model_params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)            # param_grad_or_zeros(): a zero gradient still decays the moments
        scale = 2.0 ** self.lg_loss_scale
        gstat = ops.gradnorm_clip([p.grad for p in self.model_params], 0.0)        # (norm of the SCALED gradient, -, non-finite flag)
        pstat = ops.gradnorm_clip([m.detach() for m in self.master_params], 0.0)
        found = gstat[2:3]
        inv = torch.full((1,), 1.0 / scale, dtype=torch.float32, device=found.device)
        opt.step_sliced(self._slices, grad_scale=inv, found_inf=found)
        gn, pn, flag = torch.stack([gstat[0], pstat[0], gstat[2]]).tolist()
        zero_grad(self.model_params)
        if flag != 0.0 or check_overflow(gn / scale):
            opt.rollback_step()
            self.lg_loss_scale -= 1
            return False
        self.log["grad_norm"], self.log["param_norm"] = gn / scale, pn
        self.lg_loss_scale += self.fp16_scale_growth
        return True

    def _optimize_fp16(self, opt):
        if self._aliased and hasattr(opt, "step_sliced") and all(P in self._slices for g in opt.param_groups for P in g["params"]):
            return self._optimize_fp16_sliced(opt)
        self.log["lg_loss_scale"] = self.lg_loss_scale
        model_grads_to_master_grads(self.param_groups_and_shapes, self.master_params)
        grad_norm, param_norm = self._compute_norms(grad_scale=2 ** self.lg_loss_scale)
        if check_overflow(grad_norm):
            self.lg_loss_scale -= 1
            zero_master_grads(self.master_params)
            return False
        self.log["grad_norm"], self.log["param_norm"] = grad_norm, param_norm
        for p in self.master_params:
            p.grad.mul_(1.0 / (2 ** self.lg_loss_scale))
        opt.step()
        zero_master_grads(self.master_params)
        master_params_to_model_params(self.param_groups_and_shapes, self.master_params)
        self.lg_loss_scale += self.fp16_scale_growth
        return True

    def _optimize_normal(self, opt):
        grad_norm, param_norm = self._compute_norms()
        self.log["grad_norm"], self.log["param_norm"] = grad_norm, param_norm
        opt.step()
        return True

    def _compute_norms(self, grad_scale=1.0):
        """One device->host transfer for both norms (the reference does two .item() per master tensor)."""
        with torch.no_grad():
            pn = torch.stack([torch.norm(p, p=2, dtype=torch.float32) ** 2 for p in self.master_params]).sum()
            gs = [torch.norm(p.grad, p=2, dtype=torch.float32) ** 2 for p in self.master_params if p.grad is not None]
            gn = torch.stack(gs).sum() if gs else torch.zeros((), device=pn.device)
            gn, pn = torch.stack([gn, pn]).tolist()
        return np.sqrt(gn) / grad_scale, np.sqrt(pn)

    def master_params_to_state_dict(self, master_params):
        state_dict = self.model.state_dict()
        if self.use_fp16:
            for master_param, (param_group, _) in zip(master_params, self.param_groups_and_shapes):
                for (name, _), unflat in zip(param_group, unflatten_master_params(param_group, master_param.view(-1))):
                    assert name in state_dict
                    state_dict[name] = unflat
        else:
            for i, (name, _v) in enumerate(self.model.named_parameters()):
                assert name in state_dict
                state_dict[name] = master_params[i]
        return state_dict

    def state_dict_to_master_params(self, state_dict):
        if self.use_fp16:
            named = [(name, state_dict[name]) for name, _ in self.model.named_parameters()]
            return make_master_params(get_param_groups_and_shapes(named))
        return [state_dict[name] for name, _ in self.model.named_parameters()]
