"""Gradient exchange for the data-parallel DxMI train step.

The reference wraps `sampler.net` and `v` in torch DDP (train_cifar10.py:298-309): NCCL all-reduce of
25 MB buckets during every backward, parameters broadcast from rank 0 at construction.  Here the
exchange is explicit and flat: after a backward, all gradients of a module are packed by ONE multi-tensor
copy into a PERSISTENT contiguous fp32 buffer (143 MB U-Net / 20.5 MB value net, allocated once), averaged by
ONE RCCL all-reduce (xGMI is point-to-point: one large message per link beats many small ones) and unpacked by
one multi-tensor copy.  Backend "nccl" == RCCL on ROCm (ReduceOp.AVG: the division happens inside the
collective); "gloo" for the CPU tests (SUM, then one division).
The reduced set is the module's FIXED list of trainable parameters — a parameter that received no gradient on
this rank contributes zeros — so every rank always issues the same collective with the same element count
(torch DDP has the same contract).
`wire_dtype=torch.bfloat16` halves the bytes on the links (71 MB for the U-Net): gradients are rounded to bf16 for
the exchange only.  It is OFF by default because it changes the averaged gradient in its low bits.
Rank-local state (replay buffer, betas_for_q EMA, randperm) is deliberately NOT synchronised — the
reference keeps it per rank too.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_parameters(module, src=0):
    """DDP-constructor semantics: every rank starts from rank `src`'s parameters and buffers.
    c10d collectives write through the raw pointer and do NOT bump a tensor's version counter, while the modules'
    packed-weight caches are keyed on (data_ptr, _version): the in-place no-op after the broadcast bumps every counter
    (one multi-tensor launch), so a broadcast after a first forward cannot leave stale packed weights behind."""
    if not is_distributed():
        return
    with torch.no_grad():
        ts = list(module.parameters()) + list(module.buffers())
        for t in ts:
            dist.broadcast(t.detach(), src)
        fl = [t for t in ts if t.is_floating_point()]
        if fl:
            torch._foreach_mul_(fl, 1.0)


class FlatGradSync:
    """All-reduce(mean) of a module's gradients through one persistent flat buffer."""

    def __init__(self, module, wire_dtype=torch.float32, force=False):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.wire_dtype = wire_dtype
        self.force = force            # run the collective even on a 1-rank group (tests of the RCCL path)
        self.flat = None
        self.views = None

    def _buffers(self, device):
        if self.flat is None or self.flat.device != device:
            n = sum(p.numel() for p in self.params)
            self.flat = torch.empty(n, dtype=self.wire_dtype, device=device)
            self.views = [c.view_as(p) for c, p in zip(self.flat.split([p.numel() for p in self.params]), self.params)]
        return self.flat, self.views

    def __call__(self):
        if not (is_distributed() or (self.force and dist.is_initialized())):
            return
        if not self.params:
            return
        for p in self.params:           # fixed element count on every rank
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in self.params]
        flat, views = self._buffers(grads[0].device)
        torch._foreach_copy_(views, grads)                    # one multi-tensor gather into the persistent buffer
        world = dist.get_world_size()
        if dist.get_backend() == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)       # RCCL: mean inside the collective
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
        torch._foreach_copy_(grads, views)                    # one multi-tensor scatter back
