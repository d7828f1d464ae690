"""Gradient exchange for the data-parallel DxMI train step.

The reference wraps `sampler.net` and `v` in torch DDP (train_cifar10.py:298-309): NCCL all-reduce of
25 MB buckets during every backward, parameters broadcast from rank 0 at construction.  Here the
exchange is explicit and flat: the gradients of a module live in ONE PERSISTENT contiguous buffer (143 MB U-Net /
20.5 MB value net, allocated once) cut into ~32 MB buckets (xGMI is point-to-point: few large messages per link beat
many small ones); a bucket's all-reduce is launched as soon as autograd has produced its gradients, so the exchange
overlaps the rest of the backward, and the means are scattered back by one multi-tensor copy.  Backend "nccl" == RCCL on ROCm (ReduceOp.AVG: the division happens inside the
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

from . import graph as _graph


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
    """All-reduce(mean) of a module's gradients through one persistent flat buffer, overlapped with the backward pass.

    The flat buffer is laid out in REVERSE parameter order (gradients arrive roughly last layer first) and cut into buckets
    of ~`bucket_mb`.  A post-accumulate-grad hook packs each gradient into its slice the moment autograd has produced it; when
    the next bucket in order is complete its all-reduce is launched asynchronously (RCCL: on the collective's own stream,
    behind an event on the compute stream), so the exchange of the deep layers runs under the backward of the shallow ones
    (train_cifar10.py:298-309 gets the same from DDP's 25 MB buckets).  Buckets are always launched in index order, so every
    rank issues the same sequence of collectives.  `sync()` (== `__call__`) after the backward: packs whatever the hooks did not
    see (a parameter without a local gradient contributes zeros to the buffer and keeps `.grad = None` unless another rank
    produced one), launches the remaining buckets and waits.  With the fp32 wire format `sync()` then RE-BINDS every `.grad` to
    its slice of the flat buffer instead of copying the means back (round 4: a 143 MB scatter per U-Net exchange); the slices
    stay untouched until the next backward packs into them, by which time the trainers have dropped the gradients
    (`zero_grad(set_to_none=True)` / `p.grad = None`: REQUIRED with the fp32 wire — a gradient that is kept and accumulated into
    in place would be an alias of a buffer a collective may still be working on; `_on_grad` un-aliases such a gradient by
    cloning it before the two-backward fallback runs).  With a narrower wire format the means are copied back by one
    multi-tensor copy.  Which parameters received a gradient on ANY rank
    travels as one flag per parameter in the TAIL of the last bucket: no collective of its own (round 4 issued a second small
    all-reduce per sync: 12 extra collectives per CIFAR train step, ~30 us of launch latency each on a path — the T + 1
    value-net exchanges — that has nothing to hide behind).  Collectives per sync = number of buckets, exactly.  Two backward
    passes without a `sync()` in between fall back to one blocking all-reduce of the accumulated gradients."""

    def __init__(self, module, wire_dtype=torch.float32, force=False, bucket_mb=32, overlap=True):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.wire_dtype = wire_dtype
        self.force = force            # run the collective even on a 1-rank group (tests of the RCCL path)
        self.flat = None
        self.views = None
        # flat layout: reversed parameter order, buckets = contiguous slices
        self.order = list(reversed(range(len(self.params))))
        itemsize = torch.empty((), dtype=wire_dtype).element_size()
        limit = max(1, int(bucket_mb * (1 << 20) / itemsize))
        self.offset = [0] * len(self.params)
        self.bucket_of = [0] * len(self.params)
        self.buckets = []             # (start, end, n_params)
        pos, start, count = 0, 0, 0
        align = max(1, 16 // itemsize)      # every slice starts on a 16-byte boundary: the multi-tensor Adam / norm / clip kernels
        for i in self.order:                # take their f32x4 path only on 16-byte aligned pointers, and .grad IS the slice after sync()
            self.offset[i] = pos
            self.bucket_of[i] = len(self.buckets)
            pos += (self.params[i].numel() + align - 1) // align * align      # (the pad elements are zeroed once and exchanged as zeros)
            count += 1
            if pos - start >= limit:
                self.buckets.append((start, pos, count))
                start, count = pos, 0
        if count or not self.buckets:
            self.buckets.append((start, pos, count))
        self.numel = pos
        # one "some rank produced a gradient" flag per parameter behind the gradients, inside the LAST bucket: that bucket is
        # launched after every other one (strict index order), i.e. when every hook that will fire has fired
        self.nflags = len(self.params)
        b0, b1, bc = self.buckets[-1]
        self.buckets[-1] = (b0, b1 + self.nflags, bc)
        self.alias_grads = wire_dtype == torch.float32
        self._reset()
        self.hooks = []
        self.overlap = overlap
        if overlap and self.params and hasattr(self.params[0], "register_post_accumulate_grad_hook"):
            self.index = {id(p): i for i, p in enumerate(self.params)}
            for p in self.params:
                self.hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ state of one backward
    def _reset(self):
        self.ready = [False] * len(self.params)
        self.pending = [b[2] for b in self.buckets]
        self.next = 0                 # next bucket to launch
        self.handles = []
        self.dirty = False

    def _active(self):
        return bool(self.params) and (is_distributed() or (self.force and dist.is_initialized()))

    def _buffers(self, device):
        if self.flat is None or self.flat.device != device:
            self.flat = torch.zeros(self.numel + self.nflags, dtype=self.wire_dtype, device=device)
            self.views = [self.flat[self.offset[i]:self.offset[i] + p.numel()].view_as(p) for i, p in enumerate(self.params)]
            self.flags = self.flat[self.numel:]
            self.flags.fill_(1.0)       # refilled at the end of every sync(): in place before the next backward's last bucket
        return self.flat, self.views

    def _pack(self, idx):
        """Gradients of parameters `idx` -> their slices of the flat buffer (a gradient that already IS its slice — kept from the
        last sync() and accumulated into in place — is not copied)."""
        todo = [i for i in idx if self.params[i].grad.data_ptr() != self.views[i].data_ptr()]
        if todo:
            torch._foreach_copy_([self.views[i] for i in todo], [self.params[i].grad for i in todo])

    def _on_grad(self, p):
        if not self._active() or _graph.capturing():      # (captured step: sync() packs and exchanges everything, see there)
            return
        i = self.index[id(p)]
        if self.ready[i]:             # second backward before sync(): the launched buckets hold stale sums
            if self.views is not None and p.grad.data_ptr() == self.views[i].data_ptr():
                p.grad = p.grad.clone()   # accumulated in place into its flat slice (gradients kept across sync()): never leave
                                          # .grad aliasing a buffer whose all-reduce may still be in flight
            self.dirty = True
            return
        self._buffers(p.grad.device)
        self._pack([i])
        self._mark(i)

    def _mark(self, i):
        self.ready[i] = True
        self.pending[self.bucket_of[i]] -= 1
        while self.next < len(self.buckets) and self.pending[self.next] == 0:
            self._launch(self.next)
            self.next += 1

    def _launch(self, b):
        start, end, _ = self.buckets[b]
        buf = self.flat[start:end]
        if dist.get_backend() == "nccl":
            # the collective runs on RCCL's stream behind the work already queued on the compute stream (c10d inserts the
            # event); async: the compute stream is not blocked until sync() waits on the handle
            self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True))
        else:
            self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    # ------------------------------------------------------------------ after the backward
    def _sync_captured(self, cap):
        """sync() of a step that is being captured into hipGraphs (dxmi_hip/graph.py).  A collective is never captured: the
        gradients are packed into the flat buffer by the graph segment that ends here, the bucket all-reduces run EAGERLY between
        two graph launches (`cap.cut`: same collectives, same order, same element counts on every rank as the eager step), and
        the segment that follows reads the means where the collective left them.  The overlap of the exchange with the backward
        is given up on this path (one backward is one graph segment): at the per-rank batches where replay matters the exchange
        is ~1 ms of a step that the host would otherwise stretch by tens of ms."""
        n = len(self.params)
        if any(p.grad is None for p in self.params):
            raise RuntimeError("FlatGradSync inside a StepGraph capture: every trainable parameter must receive a gradient "
                               "(the set of exchanged tensors is frozen into the graph)")
        dev = self.params[0].grad.device
        if self.flat is None or self.flat.device != dev:
            raise RuntimeError("FlatGradSync inside a StepGraph capture: run one eager step first (the flat buffer is allocated there)")
        flat, views = self._buffers(dev)
        self._pack(list(range(n)))

        def exchange():
            backend = dist.get_backend()
            handles = []
            for start, end, _ in self.buckets:            # strict index order, as the eager path launches them
                handles.append(dist.all_reduce(flat[start:end], op=dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM,
                                               async_op=True))
            for h in handles:
                h.wait()
            if backend != "nccl":
                flat.div_(dist.get_world_size())
                self.flags.fill_(1.0)
        cap.cut(exchange)
        if self.alias_grads:
            for i in range(n):
                self.params[i].grad = views[i]
        else:
            torch._foreach_copy_([p.grad for p in self.params], views)
        self._reset()

    def sync(self):
        if not self._active():
            return
        cap = _graph.current()
        if cap is not None:
            return self._sync_captured(cap)
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        dev = next((p.grad.device for p in self.params if p.grad is not None), self.params[0].device)
        flat, views = self._buffers(dev)
        if missing:
            # fixed element count on every rank: a parameter without a local gradient contributes zeros to the FLAT buffer
            # only; its .grad stays None unless some other rank produced one (torch DDP / torch.optim semantics: an unused
            # parameter is skipped by the optimiser, not stepped with a zero gradient).  Its flag goes out as zero: the last
            # bucket (which carries the flags) cannot have been launched yet — its launch waits for every parameter.
            torch._foreach_zero_([views[i] for i in missing])
            self.flags[torch.tensor(missing, device=dev)] = 0.0
        world = dist.get_world_size()
        if self.dirty:
            # every rank must have issued every bucket collective exactly once before the blocking one (a rank whose hooks
            # saw fewer complete buckets would otherwise be a collective behind): flush the rest, discard the results
            while self.next < len(self.buckets):
                self._launch(self.next)
                self.next += 1
            for h in self.handles:
                h.wait()
            have = [i for i in range(len(self.params)) if self.params[i].grad is not None]
            if have:                    # torch's foreach ops reject empty lists (a rank may hold no gradient at all)
                self._pack(have)
            if missing:                 # the discarded collectives wrote into these slices
                torch._foreach_zero_([views[i] for i in missing])
            self.flags.fill_(1.0)
            if missing:
                self.flags[torch.tensor(missing, device=dev)] = 0.0
            if dist.get_backend() == "nccl":
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.div_(world)
        else:
            late = [i for i in self.order if not self.ready[i]]
            if late:                    # gradients the hooks did not see (no hooks, constructed after backward, unused parameters)
                have = [i for i in late if self.params[i].grad is not None]
                if have:
                    self._pack(have)
                for i in late:
                    self._mark(i)
            for h in self.handles:
                h.wait()
            if dist.get_backend() != "nccl":
                flat.div_(world)
        if missing:
            # the flags came back with the last bucket: only a rank with locally missing gradients reads them (one host sync on
            # that rare path)
            got = self.flags[torch.tensor(missing, device=dev)].float().tolist()
            for i, f in zip(missing, got):
                if f > 0:
                    self.params[i].grad = views[i] if self.alias_grads else torch.empty_like(self.params[i])
        have = [i for i in range(len(self.params)) if self.params[i].grad is not None]
        if self.alias_grads:
            for i in have:              # the means stay where the collective left them
                self.params[i].grad = views[i]
        elif have:
            torch._foreach_copy_([self.params[i].grad for i in have], [views[i] for i in have])   # one multi-tensor scatter back
        self.flags.fill_(1.0)           # for the next backward (the next sync zeroes what is missing then)
        self._reset()

    __call__ = sync


def rank_device(local_rank):
    """Device of this rank: cuda:<LOCAL_RANK>, one process per GPU (the reference's torchrun convention).  DXMI_DIST_ONE_DEVICE=1 puts
    every rank on cuda:0 — the world-size-2 tests of the scripts on a 1-GPU box (with DXMI_DIST_BACKEND=gloo: RCCL refuses two ranks on
    a device)."""
    import os
    return "cuda:0" if os.environ.get("DXMI_DIST_ONE_DEVICE") == "1" else f"cuda:{local_rank}"


def dist_backend():
    """'nccl' (= RCCL on ROCm) unless DXMI_DIST_BACKEND overrides it (tests)."""
    import os
    return os.environ.get("DXMI_DIST_BACKEND", "nccl")
