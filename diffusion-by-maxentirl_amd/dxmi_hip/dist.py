"""Gradient exchange for the data-parallel DxMI train step.

The reference wraps `sampler.net` and `v` in torch DDP (train_cifar10.py:298-309): NCCL all-reduce of
25 MB buckets during every backward, parameters broadcast from rank 0 at construction.  Here the
exchange is explicit and flat: after a backward, all gradients of a module are packed into ONE
contiguous fp32 buffer (143 MB U-Net / 20.5 MB value net), all-reduced once over RCCL (xGMI is
point-to-point: one large message per link beats many small ones) on the current stream and scattered
back, averaged over ranks.  Backend "nccl" == RCCL on ROCm; "gloo" for the CPU tests.
Rank-local state (replay buffer, betas_for_q EMA, randperm) is deliberately NOT synchronised — the
reference keeps it per rank too.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_parameters(module, src=0):
    """DDP-constructor semantics: every rank starts from rank `src`'s parameters and buffers."""
    if not is_distributed():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


class FlatGradSync:
    """All-reduce(mean) of a module's gradients through one flat buffer."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.flat = None

    def __call__(self):
        if not is_distributed():
            return
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        grads = [p.grad for p in ps]
        if all(g.dtype == torch.float32 for g in grads):
            # one gather kernel, one collective, one multi-tensor scatter (instead of 2 x len(ps) small copies)
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(dist.get_world_size())
            torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
            return
        n = sum(g.numel() for g in grads)
        if self.flat is None or self.flat.numel() != n or self.flat.device != grads[0].device:
            self.flat = torch.empty(n, dtype=torch.float32, device=grads[0].device)
        off = 0
        for g in grads:
            k = g.numel()
            self.flat[off:off + k].copy_(g.reshape(-1))
            off += k
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        off = 0
        for g in grads:
            k = g.numel()
            g.copy_(self.flat[off:off + k].view_as(g))
            off += k
