"""hipGraph replay of whole DxMI steps (generation call, value update, policy update) — the host-unbound small-batch path.

The reference divides `training.batchsize` by the number of GPUs (train_cifar10.py:298-301, train_image_large.py:190-193): at 8
GPUs a rank steps 32 (CIFAR T=10), 128 (T=4) or 16 (ImageNet-64) images, and the ~9 k kernel launches of a train step then cost
more host time to ISSUE from Python (47.7 ms measured at round 5) than the GPU needs to run them.  `StepGraph` removes the host
from the loop: the step function runs eagerly `warmup` times (weight packs, workspaces, optimiser state, kernel attributes), is
then stream-captured ONCE into hipGraphs, and every later call is: host producers -> one pinned H2D copy -> hipGraphLaunch.

What makes a python step replayable:
  * static addresses     tensor arguments are copied into static input buffers; everything the step allocates while it is being
                         captured comes from the graph's private pool (torch's caching allocator) and keeps its address;
                         packed weights are refreshed IN PLACE (ops.pack_batch(reuse=...)) and a re-allocation anywhere bumps
                         ops.PACK_GENERATION, which makes every StepGraph re-capture itself;
  * host inputs          values the host computes per step — Adam's bias-corrected step sizes, dropout seeds, `torch.randperm`
                         rows (CPU generator: reference trainer.py:271, :352) — are DECLARED while capturing
                         (`current().host_input(dtype, n, producer)`): the producer fills a slot of one pinned staging block that
                         is uploaded once per replay, and the kernels read device memory (dxmi_adam_step_dev,
                         dxmi_dropout_bf16_dev) instead of by-value arguments.  Producers run in declaration order and advance
                         the same host state (optimiser step counters, dropout call counter, CPU generator) an eager step
                         advances, so an eager run and a replayed run walk the same sequence;
  * cuts                 work that must stay outside a graph (the RCCL all-reduce of a gradient exchange, dxmi_hip/dist.py)
                         calls `current().cut(fn)`: the capture is split there, `fn` runs eagerly between two graph launches —
                         collectives are issued exactly as in the eager step, in the same order on every rank;
  * device RNG           torch's Philox generator is graph-aware (the offset advances per replay).
The capture pass is a REAL step: each segment is launched as soon as its capture ends, so the call that captures returns what an
eager call would have returned.  Outputs are static tensors: they are overwritten by the next call of the same StepGraph.
"""
import gc
import os

import numpy as np
import torch

from ._lib import DxmiError

_CURRENT = None

_NP_OF = {torch.float32: np.float32, torch.int64: np.int64, torch.int32: np.uint32}


def current():
    """The StepGraph being captured on this process, or None (eager execution)."""
    return _CURRENT


def capturing():
    return _CURRENT is not None


def default_enabled():
    """DXMI_GRAPH=0 switches the graph paths of the scripts / bench off (A/B timing); they are on otherwise."""
    return os.environ.get("DXMI_GRAPH", "1") != "0"


class StepGraph:
    def __init__(self, fn, device, warmup=1, modules=(), stage_bytes=1 << 20, name="step"):
        """fn(*tensors) -> tensor | tuple | list | dict | None.  modules: the networks whose packed bf16 weights the step reads or
        whose parameters it updates (objects with `prepare_capture()` / `refresh_packs()`): their packs are refreshed at the END
        of the captured step (inside the capture), so that the next replay — whose python never runs — starts from current
        fragments, and checked on the host before every replay (an eager optimiser step or load_state_dict in between)."""
        self.fn, self.modules, self.name = fn, list(modules), name
        self.device = torch.device(device)
        self.warmup, self.calls = warmup, 0
        self.stage_bytes = stage_bytes
        self.segments = None
        self.generation = None
        self.replays = 0
        self.captures = 0

    # ------------------------------------------------------------------ declared while capturing
    def host_input(self, dtype, n, producer):
        """-> DEVICE tensor of `n` elements of torch dtype `dtype` (float32 / int64 / int32 = raw 32-bit words) that holds
        producer() — a sequence, numpy array or CPU tensor of n values — at every replay."""
        if _CURRENT is not self:
            raise DxmiError("StepGraph.host_input outside this graph's capture")
        npd = _NP_OF[dtype]
        nbytes = n * np.dtype(npd).itemsize
        off = (self.used + 15) & ~15
        if off + nbytes > self.stage_bytes:
            raise DxmiError(f"StepGraph '{self.name}': host inputs exceed the {self.stage_bytes}-byte staging block")
        self.used = off + nbytes
        view = self.stage_np[off:off + nbytes].view(npd)

        def fill():
            v = producer()
            view[:] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
        fill()                                   # the capture pass is a real step
        self.producers.append(fill)
        return self.stage_dev[off:off + nbytes].view(dtype)

    # ------------------------------------------------------------------ a second captured stream (fork / join inside the graph)
    def fork_side(self, reads=()):
        """-> a side stream that has been made to wait for everything captured so far on the current stream: work launched on it
        becomes a parallel branch of the hipGraph (weight-gradient kernels next to the data-gradient chain of a backward: at the
        per-rank batches of a multi-GPU run one kernel does not fill 256 CUs).  reads: tensors allocated on the main stream that the
        branch reads — their blocks are held until the capture ends (record_stream), since the main stream is free to drop them
        while the branch is still pending.  `join_side()` before anything on the main stream consumes the branch's results."""
        if _CURRENT is not self:
            raise DxmiError("StepGraph.fork_side outside this graph's capture")
        if self.side is None:
            self.side = torch.cuda.Stream(self.device)
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        for t in reads:
            if t is not None:
                t.record_stream(self.side)
        self.forked = True
        return self.side

    def join_side(self):
        if self.forked:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self.forked = False

    def cut(self, eager_fn):
        """End the graph segment here, run `eager_fn()` un-captured (now and between the two launches of every replay), start the
        next segment."""
        if _CURRENT is not self:
            raise DxmiError("StepGraph.cut outside this graph's capture")
        self._end_segment()
        eager_fn()
        self.segments.append(("eager", eager_fn))
        self._begin_segment()

    # ------------------------------------------------------------------ capture
    def _begin_segment(self):
        self._g = torch.cuda.CUDAGraph()
        # thread_local: launches from torch's autograd worker thread are captured all the same (capture is a property of the
        # stream), while unrelated threads (RCCL watchdog event queries, PNG writers) cannot invalidate the capture
        self._g.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def _end_segment(self):
        self.join_side()                         # a capture cannot end with an un-joined branch
        self._g.capture_end()
        self.segments.append(("graph", self._g))
        self._upload()
        self._g.replay()
        self._g = None

    def _upload(self):
        if self.used:
            self.stage_dev[:self.used].copy_(self.stage_host[:self.used], non_blocking=True)
            self.uploaded.record()

    def _capture(self, args):
        global _CURRENT
        from . import ops
        if _CURRENT is not None:
            raise DxmiError("nested StepGraph capture")
        dev = self.device
        self.stage_host = torch.empty(self.stage_bytes, dtype=torch.uint8).pin_memory()
        self.stage_np = self.stage_host.numpy()
        self.stage_dev = torch.zeros(self.stage_bytes, dtype=torch.uint8, device=dev)
        self.uploaded = torch.cuda.Event()
        self.used, self.producers, self.segments = 0, [], []
        self.side, self.forked = None, False
        for m in self.modules:
            m.prepare_capture()
        self.static_in = [a.detach().clone() if torch.is_tensor(a) and not getattr(a, "_dxmi_static", False) else a for a in args]
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        gc.collect()
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        self.generation = ops.PACK_GENERATION
        self.captures += 1
        _CURRENT = self
        ok = False
        try:
            with torch.cuda.stream(self.stream):
                self._begin_segment()
                out = self.fn(*self.static_in)
                for m in self.modules:
                    m.refresh_packs()
                self._end_segment()
            ok = True
        finally:
            _CURRENT = None
            if not ok:
                try:
                    if self._g is not None:
                        self._g.capture_end()
                except Exception:
                    pass
                self.segments = None
        cur.wait_stream(self.stream)
        if ops.PACK_GENERATION != self.generation:
            # a packed-weight set was (re)allocated inside the capture: the graph holds addresses of buffers that are gone
            self.segments = None
            raise DxmiError(f"StepGraph '{self.name}': packed weights were re-allocated during capture (run the step eagerly first)")
        self.outputs = out
        return out

    # ------------------------------------------------------------------ call
    def __call__(self, *args):
        from . import ops
        self.calls += 1
        if self.calls <= self.warmup:
            return self.fn(*args)
        if self.segments is not None and self.generation != ops.PACK_GENERATION:
            self.segments = None                 # packed weights moved (module.to(), a new parameter): capture again
        if self.segments is None:
            return self._capture(args)
        for m in self.modules:
            m.refresh_packs()                    # host-side version check; launches only after an eager update in between
        for s, a in zip(self.static_in, args):
            if torch.is_tensor(a) and a.data_ptr() != s.data_ptr():
                s.copy_(a)
        self.uploaded.synchronize()              # the previous replay's upload has left the pinned block
        for fill in self.producers:
            fill()
        self._upload()
        for kind, obj in self.segments:
            if kind == "graph":
                obj.replay()
            else:
                obj()
        self.replays += 1
        return self.outputs


def static(t):
    """Mark a tensor argument whose storage is the same at every call (a replay-ring slot): StepGraph uses it in place instead
    of copying it into a static input buffer."""
    t._dxmi_static = True
    return t
