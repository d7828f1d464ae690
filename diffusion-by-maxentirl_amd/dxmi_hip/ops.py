"""Tensor-level wrappers over the C-ABI.  torch is used only for device memory and streams.

Activations are torch.bfloat16 NHWC tensors of shape [N, H, W, C]; network edges are fp32 NCHW.
Every function launches asynchronously on torch's current stream and allocates its output with
torch.empty unless `out=` is given (pass `out=` inside hipGraph capture to keep addresses fixed).
"""
import ctypes
import os

import torch

from . import _lib
from . import graph as _graph
from ._lib import ConvDesc, check, load

ACT_NONE, ACT_LEAKY02, ACT_RELU, ACT_SILU = 0, 1, 2, 3
IN_NHWC_BF16, IN_NCHW_F32_K27 = 0, 1
OUT_NHWC_BF16, OUT_NCHW_F32 = 0, 1


# torch.cuda.current_stream() costs ~9 us of Python per call (device-index and availability checks, a Stream object) and sits in
# front of every launch: a train step spent 14 ms of its 57 ms of host time there (tools/host_profile.py).  The raw handle of the
# same stream comes from two C calls.
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tree_fingerprint(module):
    """Cheap identity of a module tree: changes when a submodule is added, removed or replaced anywhere below `module`."""
    fp, stack = [], [module]
    while stack:
        m = stack.pop()
        subs = m._modules
        fp.append(id(subs))
        fp.append(len(subs))
        for c in subs.values():
            if c is not None:
                fp.append(id(c))
                stack.append(c)
    return tuple(fp)


def fast_parameters(module):
    """module.parameters() without nn.Module's name-building / de-duplication machinery (a U-Net's `tuple(... for p in
    net.parameters())` was ~1 ms of Python per forward): the submodules are listed once, their `_parameters` dictionaries are read
    live, in the order `parameters()` yields.  The cached list is dropped when the module tree changes: the root's direct children
    are compared on EVERY call (count + identities, ~1 us), the whole tree's fingerprint (~30 us for the U-Net) on every 64th —
    a submodule replaced deeper down is seen within 64 calls, one replaced at the root at once — and on first use the result is
    checked against `module.parameters()` (a parameter shared between two modules would be yielded twice here: de-duplicated
    by identity)."""
    cache = module.__dict__.get("_dxmi_param_modules")
    if cache is not None:
        cache[2] += 1
        subs = module._modules
        if len(subs) != len(cache[4]) or any(a is not b for a, b in zip(subs.values(), cache[4])):
            cache = None
        elif cache[2] & 63 == 0 and _tree_fingerprint(module) != cache[1]:
            cache = None
    if cache is None:
        # every submodule, also those without parameters today: the samplers register `log_betas` on a net after construction
        mods = list(module.modules())
        cache = [mods, _tree_fingerprint(module), 0, False, tuple(module._modules.values())]
        module.__dict__["_dxmi_param_modules"] = cache
        seen = set()
        flat = [prm for m in mods for prm in m._parameters.values() if prm is not None and not (id(prm) in seen or seen.add(id(prm)))]
        cache[3] = len(flat) != sum(1 for m in mods for prm in m._parameters.values() if prm is not None)      # shared parameters
        assert len(flat) == sum(1 for _ in module.parameters()), "fast_parameters: module tree walk disagrees with module.parameters()"
    if cache[3]:
        seen = set()
        for m in cache[0]:
            for prm in m._parameters.values():
                if prm is not None and id(prm) not in seen:
                    seen.add(id(prm))
                    yield prm
        return
    for m in cache[0]:
        for prm in m._parameters.values():
            if prm is not None:
                yield prm


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DxmiError("dxmi_hip ops need device tensors: the HIP library is the only compute path "
                                 "(the CPU oracle lives under oracle/ and is test infrastructure)")


def device_check():
    check(load().dxmi_device_check(), "dxmi_device_check")


def set_tuning(name, value):
    """Kernel-selection knob of the library (include/dxmi_hip.h: dxmi_set_tuning); returns the previous value."""
    old = get_tuning(name)
    check(load().dxmi_set_tuning(name.encode(), int(value)), "dxmi_set_tuning")
    _STATS_P.clear()          # per-shape answers of the selection queries (statistics partials, fused GroupNorm) depend on the knobs
    return old


_TUNING_SAVED = []


def tune_for_throughput(on=True):
    """Training entry points: route convs whose 256-pixel-tile grid leaves most CUs idle (small per-GPU batches) to kernels with
    smaller tiles.  Off (the library default), every layer shape runs one kernel whatever the batch size, which keeps an image's
    result bitwise independent of the batch it rides in (generation scripts, parity tests).  `tune_for_throughput(False)`
    RESTORES the values that were in effect before the matching `tune_for_throughput(True)` (environment overrides such as
    DXMI_CONV_SM / DXMI_CONV_WS_MIN_TILES survive a train leg); `with ops.throughput_tuning():` does both."""
    if on:
        _TUNING_SAVED.append((set_tuning("conv_ws_min_tiles", 96), set_tuning("conv_sm_mask", 13)))
    elif _TUNING_SAVED:
        ws, sm = _TUNING_SAVED.pop()
        set_tuning("conv_ws_min_tiles", ws)
        set_tuning("conv_sm_mask", sm)


class throughput_tuning:
    """`with ops.throughput_tuning():` — tune_for_throughput(True) inside, the previous knob values restored on exit."""

    def __enter__(self):
        tune_for_throughput(True)
        return self

    def __exit__(self, *exc):
        tune_for_throughput(False)
        return False


def get_tuning(name):
    v = ctypes.c_int(0)
    check(load().dxmi_get_tuning(name.encode(), ctypes.byref(v)), "dxmi_get_tuning")
    return v.value


def conv_ws_clock_ghz():
    """Shader clock (GHz) the chip held during the most recent conv_ws_kernel launch (dxmi_conv_ws_last_clock), or None when the
    kernel has not run / the counters did not advance.  Synchronises."""
    buf = (ctypes.c_uint64 * 4)()
    check(load().dxmi_conv_ws_last_clock(buf), "dxmi_conv_ws_last_clock")
    dc, dr = int(buf[2]) - int(buf[0]), int(buf[3]) - int(buf[1])
    return (dc / dr) * 0.1 if dc > 0 and dr > 0 else None


class PackedConvWeight:
    """bf16 MFMA-fragment-ordered copy of an OIHW fp32 weight."""

    __slots__ = ("buf", "Cout", "Cin", "ksize", "k27")

    def __init__(self, buf, Cout, Cin, ksize, k27):
        self.buf, self.Cout, self.Cin, self.ksize, self.k27 = buf, Cout, Cin, ksize, k27


_PACK_BATCH = None
_PACK_REC = None            # the outermost active pack_batch: hands out (and records) the destination buffers of the block
PACK_GENERATION = 0         # bumped whenever a recorded packed-weight buffer is freshly allocated (dxmi_hip/graph.py re-captures)


class pack_batch:
    """`with ops.pack_batch():` — every pack_conv_weight() inside allocates its destination and is DEFERRED; on exit all of
    them run as a few multi-tensor launches (dxmi_pack_conv_weights).  The nets' `_pack` / `_pack_t` use it: after an
    optimiser step ~60 (CIFAR U-Net) to ~330 (ImageNet-64 EDM net) weights are refreshed.
    Every destination the block asks for (packed fragments, and the derived tensors of `pack_tensor`) is recorded in call order
    in `.buffers`; `pack_batch(reuse=previous.buffers)` hands the SAME buffers out again in the same order: a re-pack after an
    optimiser step (parameters updated in place) then rewrites the fragments where they are — the addresses a captured hipGraph
    holds stay valid (dxmi_hip/graph.py).  A block that allocates anything fresh bumps PACK_GENERATION."""

    def __init__(self, reuse=None):
        self.reuse = list(reuse) if reuse is not None else None

    def __enter__(self):
        global _PACK_BATCH, _PACK_REC
        self.plan = None
        self.nested = _PACK_REC is not None
        self.outer = _PACK_BATCH is not None or os.environ.get("DXMI_PACK_BATCH", "1") == "0"     # (=0: one launch per weight, A/B timing)
        if not self.outer:
            _PACK_BATCH = []
        if not self.nested:
            _PACK_REC = self
            self.buffers, self.cursor, self.fresh = [], 0, False
        return self

    def alloc(self, shape, dtype, device):
        """Next destination of the block: the matching buffer of `reuse`, or a new one."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        if self.reuse is not None and self.cursor < len(self.reuse):
            t = self.reuse[self.cursor]
            if tuple(t.shape) != shape or t.dtype != dtype or t.device != torch.device(device):
                raise _lib.DxmiError("pack_batch(reuse=...): the block asks for different buffers than the pack it replays")
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
            self.fresh = True
        self.cursor += 1
        self.buffers.append(t)
        return t

    def __exit__(self, *exc):
        global _PACK_BATCH, _PACK_REC, PACK_GENERATION
        if not self.nested:
            _PACK_REC = None
            if self.fresh:
                PACK_GENERATION += 1
        if self.outer:
            return False
        items, _PACK_BATCH = _PACK_BATCH, None
        if items and exc[0] is None:
            arr = (_lib.PackItem * len(items))()
            for i, (w, out, Cout, Cin, k, flip, k27) in enumerate(items):
                arr[i].w, arr[i].dst = w.data_ptr(), out.data_ptr()
                arr[i].Cout, arr[i].Cin, arr[i].ksize, arr[i].transpose_flip, arr[i].k27 = Cout, Cin, k, int(flip), int(k27)
            self.plan = PackPlan(arr, items)
            self.plan.replay()
        return False


def pack_tensor(shape, dtype, device):
    """Destination for a DERIVED tensor of a weight pack (concatenated biases, zero-padded sources): recorded / reused by the
    enclosing pack_batch like the packed fragments; a plain allocation outside one."""
    if _PACK_REC is not None:
        return _PACK_REC.alloc(shape, dtype, device)
    return torch.empty(shape, dtype=dtype, device=device)


PACK_PLAN_REPLAY = os.environ.get("DXMI_PACK_PLAN", "1") != "0"      # 0: rebuild descriptors and buffers at every re-pack (A/B timing)


class PackPlan:
    """The descriptor array of one pack_batch, replayable: after an optimiser step the SAME fp32 sources (parameters updated in
    place: same addresses) are packed again into the SAME fragment buffers with one call — building the ~330 descriptors and
    buffers of the ImageNet-64 net anew was 3.6 ms of host time per pack, twenty times per EDM train step.  Holds the source and
    destination tensors alive; the caller checks that the parameters have not moved before replaying."""

    def __init__(self, arr, items):
        self.arr, self.items = arr, items

    def replay(self):
        check(load().dxmi_pack_conv_weights(self.arr, len(self.items), _stream()), "dxmi_pack_conv_weights")


def pack_conv_weight(w, transpose_flip=False, k27=False, out=None):
    """w: fp32 [Cout, Cin, k, k] (or [M, K] for a linear layer) on the device.  Inside `with pack_batch():` the launch is
    deferred to the end of the block (the returned buffer is filled then)."""
    _need_cuda(w)
    if w.dim() == 2:
        w = w[:, :, None, None]
    w = w.detach().contiguous().float()
    O, I, k, _ = w.shape
    Cout, Cin = (I, O) if transpose_flip else (O, I)
    lib = load()
    nbytes = lib.dxmi_packed_conv_weight_bytes(Cout, Cin, k, int(k27))
    if out is None:
        out = pack_tensor(nbytes, torch.uint8, w.device)
    assert out.numel() == nbytes
    if _PACK_BATCH is not None:
        _PACK_BATCH.append((w, out, Cout, Cin, k, transpose_flip, k27))      # (w is kept alive until the launch)
    else:
        check(lib.dxmi_pack_conv_weight(_ptr(w), _ptr(out), Cout, Cin, k, int(transpose_flip), int(k27), _stream()),
              "dxmi_pack_conv_weight")
    return PackedConvWeight(out, Cout, Cin, k, k27)


class BlockStats:
    """GroupNorm block statistics of an NHWC bf16 activation: fp32 [N, P, C/2, 2] = (sum, sum of squares) of the stored
    values per channel PAIR, over the pixels of partial p of the image (dxmi_conv_desc.gn_stats / dxmi_gn_block_stats).
    Travels with the tensor it describes; `groupnorm_silu(..., stats=...)` then runs as one streaming read + write."""

    __slots__ = ("buf", "P")

    def __init__(self, buf, P):
        self.buf, self.P = buf, P


_STATS_P = {}


def conv2d(x, pw, *, in1=None, bias=None, addvec=None, residual=None, stride=1, pad=None, pad_br=None, upsample=False,
           act=ACT_NONE, out=None, out_nchw_f32=False, variant=0, mask_src=None, mask_slope=0.0, want_stats=False, fuse_gn=None):
    """x: NHWC bf16 [N,IH,IW,C0] (or NCHW fp32 [N,3,H,W] when pw.k27).
    want_stats: return (out, BlockStats | None) — the output's GroupNorm block statistics written by the conv's own epilogue
    where the selected kernel can (None otherwise: the caller falls back to block_stats() or the one-pass GroupNorm).
    fuse_gn = (gamma, beta, groups, eps, silu, keep_raw): return (out | None, y | None) with y = GroupNorm(+SiLU)(out) written
    by the conv's own epilogue where the selected kernel can (4x4 maps; y None otherwise: the caller normalises `out` itself);
    keep_raw False drops the un-normalised tensor (out None) when y is produced."""
    _need_cuda(x, in1, bias, addvec, residual, out)
    k = pw.ksize
    if pad is None:
        pad = k // 2
    d = ConvDesc()
    if pw.k27:
        N, _, IH, IW = x.shape
        C0, C1 = 3, 0
        assert x.dtype == torch.float32 and x.is_contiguous()
    else:
        N, IH, IW, C0 = x.shape
        C1 = in1.shape[3] if in1 is not None else 0
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        assert in1 is None or (in1.dtype == torch.bfloat16 and in1.is_contiguous() and in1.shape[:3] == x.shape[:3])
        assert C0 + C1 == pw.Cin, f"conv2d: input channels {C0}+{C1} != packed Cin {pw.Cin}"
    VIH, VIW = (IH * 2, IW * 2) if upsample else (IH, IW)
    if pad_br is None:
        pad_br = pad
    # pad = top/left zeros, pad_br = bottom/right zeros (DDPM Downsample: pad=0, pad_br=1, k3 s2;
    # unet_small.py:69-76)
    OH, OW = (VIH + pad + pad_br - k) // stride + 1, (VIW + pad + pad_br - k) // stride + 1
    Cout = pw.Cout
    if out is None:
        if out_nchw_f32:
            out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((N, OH, OW, Cout), dtype=torch.bfloat16, device=x.device)
    d.in0, d.in1, d.wpacked = x.data_ptr(), (in1.data_ptr() if in1 is not None else None), pw.buf.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if addvec is not None:
        assert addvec.dtype == torch.float32 and addvec.stride(-1) == 1
        d.addvec = addvec.data_ptr()
        d.addvec_ld = addvec.stride(0) if addvec.dim() == 2 else 0
    else:
        d.addvec, d.addvec_ld = None, 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.is_contiguous() and tuple(residual.shape) == (N, OH, OW, Cout)
    d.residual = residual.data_ptr() if residual is not None else None
    d.out = out.data_ptr()
    d.N, d.IH, d.IW, d.C0, d.C1, d.OH, d.OW, d.Cout = N, IH, IW, C0, C1, OH, OW, Cout
    d.ksize, d.stride, d.pad, d.upsample, d.act = k, stride, pad, int(upsample), act
    d.in_mode = IN_NCHW_F32_K27 if pw.k27 else IN_NHWC_BF16
    d.out_mode = OUT_NCHW_F32 if out_nchw_f32 else OUT_NHWC_BF16
    d.variant = variant
    if mask_src is not None:
        assert mask_src.dtype == torch.bfloat16 and mask_src.is_contiguous() and tuple(mask_src.shape) == (N, OH, OW, Cout)
    d.mask_src = mask_src.data_ptr() if mask_src is not None else None
    d.mask_slope = float(mask_slope)
    d.gn_stats = None
    stats = None
    if want_stats and not out_nchw_f32 and variant == 0:
        key = (N, IH, IW, C0, C1, OH, OW, Cout, k, stride, pad, int(upsample), act, d.in_mode, residual is not None, mask_src is not None)
        P = _STATS_P.get(key)
        if P is None:
            P = _STATS_P[key] = int(load().dxmi_conv2d_gn_stats_partials(ctypes.byref(d)))
        if P > 0:
            stats = BlockStats(torch.empty((N, P, Cout // 2, 2), dtype=torch.float32, device=x.device), P)
            d.gn_stats = stats.buf.data_ptr()
    d.gn_out, y = None, None
    if fuse_gn is not None:
        assert not want_stats
        gamma, beta, groups, eps, silu, keep_raw = fuse_gn
        d.gn_groups = groups
        d.gn_flags = int(bool(silu)) | (0 if keep_raw else 2)
        key = ("gn", N, IH, IW, C0, C1, OH, OW, Cout, k, stride, pad, int(upsample), act, d.in_mode, d.out_mode, variant,
               residual is not None, mask_src is not None, groups, bool(keep_raw))
        ok = _STATS_P.get(key)
        if ok is None:
            ok = _STATS_P[key] = int(load().dxmi_conv2d_gn_fuse_supported(ctypes.byref(d)))
        if ok:
            assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == Cout and beta.numel() == Cout
            y = torch.empty((N, OH, OW, Cout), dtype=torch.bfloat16, device=x.device)
            d.gn_out, d.gn_gamma, d.gn_beta, d.gn_eps = y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps)
    if PROFILER is not None:
        PROFILER.launch_conv(d)
    else:
        check(load().dxmi_conv2d_fwd(ctypes.byref(d), _stream()), "dxmi_conv2d_fwd")
    if fuse_gn is not None:
        return (out if (y is None or fuse_gn[5]) else None), y
    if stats is not None and stats.P > MAX_APPLY_PARTIALS:
        stats = fold_stats(stats)
    return (out, stats) if want_stats else out


class OpProfiler:
    """Brackets kernel launches with HIP events on the launch stream (bench.py roofline leg) and records, per launch,
    (class, kernel name, algorithmic FLOPs, algorithmic bytes, start, end).  Classes: "conv3x3" / "conv1x1" / "conv_stem" /
    "conv_other" (by dxmi_conv2d_kernel_id), "groupnorm", "attention", "sampler_step", "wgrad", "groupnorm_bwd", "optimizer"."""

    def __init__(self):
        self.records = []

    def bracket(self, cls, name, flops, nbytes, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.append((cls, name, float(flops), float(nbytes), e0, e1))
        return out

    def launch_conv(self, d):
        lib = load()
        kid = lib.dxmi_conv2d_kernel_id(ctypes.byref(d))
        cin = 27 if d.in_mode == IN_NCHW_F32_K27 else (d.C0 + d.C1) * d.ksize * d.ksize
        flops = 2.0 * d.N * d.OH * d.OW * d.Cout * cin
        in_b = d.N * d.IH * d.IW * (d.C0 + d.C1) * (4 if d.in_mode == IN_NCHW_F32_K27 else 2)
        out_b = d.N * d.OH * d.OW * d.Cout * (4 if d.out_mode == OUT_NCHW_F32 else 2)
        res_b = out_b if d.residual else 0
        if d.gn_out:                                  # fused GroupNorm: the normalised tensor out, the raw one only if kept
            res_b += out_b if not (d.gn_flags & 2) else 0
        w_b = d.Cout * cin * 2
        cls = "conv_other" if kid >= 600000 else "conv1x1" if kid >= 500000 else "conv3x3" if kid >= 400000 else "conv_stem" if kid >= 300000 else ("conv1x1" if kid >= 200000 else ("conv3x3" if kid >= 30000 else "conv_other"))
        if kid == 400008 and d.gn_out:
            kid = 400009                                  # conv_ws8_kernel<true> (bench.py kernel_name)
        self.bracket(cls, kid, flops, in_b + out_b + res_b + w_b,
                     lambda: check(lib.dxmi_conv2d_fwd(ctypes.byref(d), _stream()), "dxmi_conv2d_fwd"))

    def summary(self):
        """(class, name) -> dict(launches, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for cls, name, fl, by, e0, e1 in self.records:
            s = out.setdefault((cls, name), {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            s["launches"] += 1
            s["ms"] += e0.elapsed_time(e1)
            s["flops"] += fl
            s["bytes"] += by
        return out


PROFILER = None


def _prof(cls, name, flops, nbytes, fn):
    return fn() if PROFILER is None else PROFILER.bracket(cls, name, flops, nbytes, fn)


_WS = {}


def _workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream): split-K partials, column sums, generic-GroupNorm partials.  Keyed by
    the launch stream so that work overlapped on a second stream never shares it; a buffer captured into a hipGraph is
    never re-allocated (growth during capture raises: size it with one eager call first)."""
    if _graph.capturing():
        # inside a StepGraph capture: a scratch tensor of its own from the graph's private pool (stream-ordered reuse is the
        # allocator's business there; the address is fixed for every replay)
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device)
    key = (str(device), stream.cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and torch.cuda.is_current_stream_capturing():
            raise _lib.DxmiError("dxmi_hip workspace would have to grow during hipGraph capture: run the step once eagerly first")
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


WGRAD_SIDE_STREAM = os.environ.get("DXMI_WGRAD_STREAM", "0") == "1"      # measured slower on gfx950 (B = 32: 35.9 -> 38.9 ms per train step, B = 256: 90.0 -> 92.5): off unless asked for


class wgrad_branch:
    """`with ops.wgrad_branch(reads):` — inside a StepGraph capture the block's launches go to the graph's SIDE stream (a parallel
    branch behind everything captured so far: StepGraph.fork_side); outside a capture, or with DXMI_WGRAD_STREAM=0, a no-op.
    Weight gradients have no consumer until the optimiser: the backward's data-gradient chain need not wait for them.
    `ops.wgrad_join()` before their first consumer on the main stream (the end of every autograd backward here)."""

    def __init__(self, reads=()):
        self.reads, self.ctx = reads, None

    def __enter__(self):
        cap = _graph.current()
        if cap is not None and WGRAD_SIDE_STREAM and PROFILER is None:
            self.ctx = torch.cuda.stream(cap.fork_side(self.reads))
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def wgrad_join():
    cap = _graph.current()
    if cap is not None:
        cap.join_side()


def conv2d_wgrad(x, dy, ksize, **kw):
    """Weight gradient of a conv (see _conv2d_wgrad); inside a captured step it runs as a parallel branch (wgrad_branch)."""
    with wgrad_branch((x, dy, kw.get("in1"), kw.get("out"))):
        return _conv2d_wgrad(x, dy, ksize, **kw)


def _conv2d_wgrad(x, dy, ksize, *, in1=None, pad=None, stride=1, upsample=False, out=None, accumulate=False, with_bias=False):
    """Weight gradient of a conv: x [N,IH,IW,C0] (| in1), dy [N,OH,OW,Cout] NHWC bf16 ->
    fp32 OIHW [Cout, C0+C1, k, k] (written, or added when accumulate).  with_bias: also the bias gradient
    (column sums of dy) from the same launch -> (dw, db)."""
    _need_cuda(x, in1, dy, out)
    N, IH, IW, C0 = x.shape
    C1 = in1.shape[3] if in1 is not None else 0
    _, OH, OW, Cout = dy.shape
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.is_contiguous() and dy.is_contiguous()
    if pad is None:
        pad = ksize // 2
    if out is None:
        out = torch.empty((Cout, C0 + C1, ksize, ksize), dtype=torch.float32, device=x.device)
        accumulate = False
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (Cout, C0 + C1, ksize, ksize)
    lib = load()
    ws = _workspace(lib.dxmi_conv2d_wgrad_workspace_bytes(N, OH, OW, C0 + C1, Cout, ksize), x.device)
    fl = 2.0 * N * OH * OW * Cout * (C0 + C1) * ksize * ksize
    by = 2.0 * (x.numel() + (in1.numel() if in1 is not None else 0) + dy.numel()) + 4.0 * out.numel()
    if with_bias:
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        _prof("wgrad", f"k{ksize}", fl, by, lambda: check(
            lib.dxmi_conv2d_wgrad_bias(_ptr(x), C0, _ptr(in1), C1, _ptr(dy), _ptr(out), _ptr(db), _ptr(ws), N, IH, IW, OH, OW,
                                       Cout, ksize, stride, pad, int(upsample), 0, _stream()), "dxmi_conv2d_wgrad_bias"))
        return out, db
    _prof("wgrad", f"k{ksize}", fl, by, lambda: check(
        lib.dxmi_conv2d_wgrad(_ptr(x), C0, _ptr(in1), C1, _ptr(dy), _ptr(out), _ptr(ws), N, IH, IW, OH, OW, Cout, ksize,
                              stride, pad, int(upsample), int(accumulate), _stream()), "dxmi_conv2d_wgrad"))
    return out


def colsum(x2d, out=None, accumulate=False):
    """Column sums of a bf16 tensor viewed as [P, C] -> fp32 [C] (bias gradients)."""
    _need_cuda(x2d, out)
    C = x2d.shape[-1]
    P = x2d.numel() // C
    assert x2d.dtype == torch.bfloat16 and x2d.is_contiguous()
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x2d.device)
        accumulate = False
    ws = _workspace(max(256, (P + 511) // 512) * C * 4, x2d.device)
    check(load().dxmi_colsum_bf16(_ptr(x2d), _ptr(out), _ptr(ws), P, C, int(accumulate), _stream()), "dxmi_colsum_bf16")
    return out


def colsum_f32(x):
    """fp32 [B, N, C] (or [N, C]) -> [B, C] ([C]): sums over N in a fixed order (one launch; torch's reduce took 14 us)."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, (N, C) = (x.shape[0] if x.dim() == 3 else 1), x.shape[-2:]
    out = torch.empty((B, C) if x.dim() == 3 else (C,), dtype=torch.float32, device=x.device)
    check(load().dxmi_colsum_f32(_ptr(x), _ptr(out), B, N, C, _stream()), "dxmi_colsum_f32")
    return out


def colsum_per_image(x):
    """x [N,H,W,C] bf16 -> [N,C] fp32 sums over H*W."""
    _need_cuda(x)
    N, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(load().dxmi_colsum_blocks_bf16(_ptr(x), _ptr(out), N * H * W, C, H * W, _stream()), "dxmi_colsum_blocks_bf16")
    return out


def value_head_pgrad(s, w, b, dy, ow=None):
    """Parameter gradients of the value head in one launch (dxmi_value_head_pgrad): s [N, C] relu-sum features, w [C], b [1], dy [N],
    ow: the out_scale weight (device scalar) or None -> fp32 [C + 3] = d linear.weight | d linear.bias | d out_scale.weight |
    d out_scale.bias."""
    _need_cuda(s, w, b, dy, ow)
    N, C = s.shape
    assert s.dtype == torch.float32 and s.is_contiguous() and dy.dtype == torch.float32 and dy.is_contiguous() and dy.numel() == N
    out = torch.empty(C + 3, dtype=torch.float32, device=s.device)
    check(load().dxmi_value_head_pgrad(_ptr(s), _ptr(w), _ptr(b), _ptr(dy), _ptr(ow), _ptr(out), N, C, _stream()), "dxmi_value_head_pgrad")
    return out


def td_gather_cost(traj2d, state_rows, beta, *, next_rows=None, next_dense=None, out_pair):
    """One TD step's data (dxmi_td_gather_cost): out_pair [2B, ...] receives [next_state | state] gathered from the ring's trajectory
    block traj2d [rows, CHW] (INT path), returns the running cost [B] = mean_CHW (x' - x)^2 / (2 beta); beta: device scalar."""
    _need_cuda(traj2d, state_rows, beta, next_rows, next_dense, out_pair)
    B = state_rows.numel()
    CHW = traj2d.shape[1]
    assert traj2d.dtype == torch.float32 and traj2d.is_contiguous() and state_rows.dtype == torch.int64 and state_rows.is_contiguous()
    assert out_pair.dtype == torch.float32 and out_pair.is_contiguous() and out_pair.numel() == 2 * B * CHW
    assert beta.dtype == torch.float32 and beta.numel() == 1
    assert next_rows is None or (next_rows.dtype == torch.int64 and next_rows.is_contiguous() and next_rows.numel() == B)
    cost = torch.empty(B, dtype=torch.float32, device=traj2d.device)
    flat = out_pair.view(2 * B, CHW)
    _prof("replay_gather", f"td_row{CHW * 4}", 0.0, 16.0 * B * CHW, lambda: check(
        load().dxmi_td_gather_cost(_ptr(traj2d), _ptr(state_rows), _ptr(next_rows), _ptr(next_dense), _ptr(beta), _ptr(flat[:B]),
                                   _ptr(flat[B:]), _ptr(cost), B, CHW, traj2d.shape[0], _stream()), "dxmi_td_gather_cost"))
    return cost


def td_loss(v, cost, extra=None):
    """mse(v[B:], (v[:B] + extra).detach()) of one TD step (dxmi_td_loss) -> (d loss / d v [2B] with zeros in the target half,
    logs fp32 [3] = (loss, mean v(x_t), mean running cost)); extra: device scalar or None."""
    _need_cuda(v, cost, extra)
    B = cost.numel()
    assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == 2 * B and cost.dtype == torch.float32 and cost.is_contiguous()
    grad = torch.empty(2 * B, dtype=torch.float32, device=v.device)
    logs = torch.empty(3, dtype=torch.float32, device=v.device)
    check(load().dxmi_td_loss(_ptr(v), _ptr(cost), _ptr(extra), _ptr(grad), _ptr(logs), B, _stream()), "dxmi_td_loss")
    return grad, logs


def pool_act_bwd(dout, act_out, pool, slope, out=None):
    _need_cuda(dout, act_out, out)
    N, OH, OW, C = dout.shape
    H, W = (OH * 2, OW * 2) if pool else (OH, OW)
    assert dout.dtype == torch.bfloat16 and dout.is_contiguous() and act_out.shape == dout.shape and act_out.is_contiguous()
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=dout.device)
    check(load().dxmi_pool_act_bwd(_ptr(dout), _ptr(act_out), _ptr(out), N, H, W, C, int(pool), float(slope), _stream()),
          "dxmi_pool_act_bwd")
    return out


def value_head_bwd(feat, w, dy):
    """feat [N,H,W,C] bf16, w [C] fp32, dy [N] fp32 -> (dfeat bf16 like feat, s [N,C] fp32)."""
    _need_cuda(feat, w, dy)
    N, H, W, C = feat.shape
    dfeat = torch.empty_like(feat)
    s = torch.empty((N, C), dtype=torch.float32, device=feat.device)
    check(load().dxmi_value_head_bwd(_ptr(feat), _ptr(w), _ptr(dy), _ptr(dfeat), _ptr(s), N, H * W, C, _stream()),
          "dxmi_value_head_bwd")
    return dfeat, s


def groupnorm_silu_bwd(x, dy, gamma, beta, *, in1=None, add0=None, add1=None, groups=32, eps=1e-6, silu=True):
    """-> (dx0, dx1 or None, dgamma [C], dbeta [C])."""
    _need_cuda(x, in1, dy, add0, add1, gamma, beta)
    N, H, W, C0 = x.shape
    C1 = in1.shape[3] if in1 is not None else 0
    C = C0 + C1
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and tuple(dy.shape) == (N, H, W, C)
    # 12 channels per group on the large maps (the 384-channel concat inputs of the CIFAR net's up path): the register-resident kernel
    # slices them badly (414 us at 32x32, 101 us at 16x16, B = 256) and the generic three-launch path is ahead (285 / 79 us:
    # tools/gn_bwd_time.py); every other shape of the net is faster on the resident kernel.  Only the MEASURED case is routed away
    # (20 / 28 / 36 ... channels per group stay where they were)
    slow_resident = (C // groups) == 12 and H * W >= 256
    if slow_resident or not load().dxmi_groupnorm_silu_bwd_supported(C0, C1, H * W, groups):
        # shapes the register-resident backward cannot slice (the forward falls back the same way): generic three-launch path
        dx0, dx1, dg, db, _ = groupnorm_generic_bwd(x, dy, gamma, beta, in1=in1, add0=add0, add1=add1, groups=groups, eps=eps, silu=silu)
        return dx0, dx1, dg, db
    dx0 = torch.empty_like(x)
    dx1 = torch.empty_like(in1) if in1 is not None else None
    part = torch.empty((2, N, C), dtype=torch.float32, device=x.device)
    nb = 2.0 * N * H * W * C * (3 + (add0 is not None))     # x, dy (+ add) in, dx out
    _prof("groupnorm_bwd", "resident", 0.0, nb, lambda: check(
        load().dxmi_groupnorm_silu_bwd(_ptr(x), C0, _ptr(in1), C1, _ptr(dy), _ptr(add0), _ptr(add1), _ptr(gamma), _ptr(beta),
                                       _ptr(dx0), _ptr(dx1), _ptr(part[0]), _ptr(part[1]), N, H * W, groups, float(eps),
                                       int(silu), _stream()), "dxmi_groupnorm_silu_bwd"))
    red = colsum_f32(part)  # [2, C]: fixed-order reduction over images
    return dx0, dx1, red[0], red[1]


def bgemm(A, B, C, M, N, K, a_strides, a_ld, a_kc, b_strides, b_ld, b_kc, c_strides, c_ld, alpha, outer, inner):
    """Raw batched bf16 GEMM (see include/dxmi_hip.h); tensors are only used for their base pointers."""
    check(load().dxmi_bgemm_bf16(_ptr(A), _ptr(B), _ptr(C), M, N, K, a_strides[0], a_strides[1], a_ld, int(a_kc), b_strides[0],
                                 b_strides[1], b_ld, int(b_kc), c_strides[0], c_strides[1], c_ld, int(C.dtype == torch.float32),
                                 float(alpha), outer, inner, _stream()), "dxmi_bgemm_bf16")
    return C


FUSED_ATTENTION_BWD = True      # A-B switch: False keeps the five-GEMM path for every head size


def attention_bwd(qkv, do, heads, scale, o=None, lse=None):
    """qkv [N,T,3C] ([q|k|v], heads = contiguous channel blocks), do [N,T,C] -> dqkv [N,T,3C] (bf16).
    o: the forward pass's attention output [N,T,C]; with it, 64-wide heads take the fused kernels (no [T,T] tensor in HBM).
    lse: what attention(want_lse=True) returned for these qkv: the fused backward skips its log-sum-exp sweep."""
    _need_cuda(qkv, do, o, lse)
    N, T, C3 = qkv.shape
    C = C3 // 3
    D = C // heads
    dev = qkv.device
    assert qkv.dtype == torch.bfloat16 and do.dtype == torch.bfloat16 and qkv.is_contiguous() and do.is_contiguous()
    if o is not None and FUSED_ATTENTION_BWD and load().dxmi_attention_bwd_supported(T, C, heads):
        assert o.dtype == torch.bfloat16 and o.is_contiguous() and o.numel() == N * T * C
        dqkv = torch.empty_like(qkv)
        ws = _workspace(max(256, load().dxmi_attention_bwd_workspace_bytes(N, T, heads)), dev)
        if lse is not None:
            assert lse.dtype == torch.float32 and lse.is_contiguous() and tuple(lse.shape) == (N, heads, T)
            _prof("attention_bwd", f"T{T}_D{D}", 14.0 * N * T * T * C, 2.0 * (2 * qkv.numel() + 2 * do.numel()) * 2, lambda: check(
                load().dxmi_attention_bwd_lse(_ptr(qkv), _ptr(o), _ptr(do), _ptr(dqkv), _ptr(lse), _ptr(ws), N, T, C, heads, float(scale),
                                              _stream()), "dxmi_attention_bwd_lse"))
            return dqkv
        _prof("attention_bwd", f"T{T}_D{D}", 16.0 * N * T * T * C, 2.0 * (2 * qkv.numel() + 2 * do.numel()) * 2, lambda: check(
            load().dxmi_attention_bwd(_ptr(qkv), _ptr(o), _ptr(do), _ptr(dqkv), _ptr(ws), N, T, C, heads, float(scale), _stream()),
            "dxmi_attention_bwd"))
        return dqkv
    S = torch.empty((N, heads, T, T), dtype=torch.float32, device=dev)
    dP = torch.empty_like(S)
    P = torch.empty((N, heads, T, T), dtype=torch.bfloat16, device=dev)
    dS = torch.empty_like(P)
    dqkv = torch.empty_like(qkv)
    q, k, v = qkv[:, :, 0:C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    dq, dk, dv = dqkv[:, :, 0:C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:]
    qs, os_, ss = (T * C3, D), (T * C, D), (heads * T * T, T * T)
    bgemm(q, k, S, T, T, D, qs, C3, True, qs, C3, True, ss, T, scale, N, heads)          # S = scale Q K^T
    bgemm(do, v, dP, T, T, D, os_, C, True, qs, C3, True, ss, T, 1.0, N, heads)           # dP = dO V^T
    check(load().dxmi_softmax_bwd(_ptr(S), _ptr(dP), _ptr(P), _ptr(dS), N * heads * T, T, _stream()), "dxmi_softmax_bwd")
    bgemm(P, do, dv, T, D, T, ss, T, False, os_, C, False, qs, C3, 1.0, N, heads)         # dV = P^T dO
    bgemm(dS, k, dq, T, D, T, ss, T, True, qs, C3, False, qs, C3, scale, N, heads)        # dQ = scale dS K
    bgemm(dS, q, dk, T, D, T, ss, T, False, qs, C3, False, qs, C3, scale, N, heads)       # dK = scale dS^T Q
    return dqkv


MAX_APPLY_PARTIALS = 8      # more partials per image than this are folded into one before the apply pass (dxmi_gn_stats_fold)


def fold_stats(st, group=32):
    """[N, P, C/2, 2] -> [N, P', C/2, 2] with P' <= MAX_APPLY_PARTIALS: `group` consecutive partials are added per thread, in
    order; a second pass folds what is left (P > 256: the 256x256 maps of the LSUN net)."""
    while st.P > MAX_APPLY_PARTIALS:
        N, P, C2, _ = st.buf.shape
        PG = (P + group - 1) // group
        out = torch.empty((N, PG, C2, 2), dtype=torch.float32, device=st.buf.device)
        _prof("groupnorm", "fold", 0.0, 4.0 * (st.buf.numel() + out.numel()), lambda: check(
            load().dxmi_gn_stats_fold(_ptr(st.buf), _ptr(out), N, P, C2 * 2, group, _stream()), "dxmi_gn_stats_fold"))
        st = BlockStats(out, PG)
    return st


def block_stats(x):
    """GroupNorm block statistics of an activation that has none from its producer (one read of x)."""
    _need_cuda(x)
    N, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    lib = load()
    P = int(lib.dxmi_gn_block_stats_partials(H * W))
    st = BlockStats(torch.empty((N, P, C // 2, 2), dtype=torch.float32, device=x.device), P)
    _prof("groupnorm", "block_stats", 0.0, 2.0 * x.numel(), lambda: check(
        lib.dxmi_gn_block_stats(_ptr(x), _ptr(st.buf), N, H * W, C, _stream()), "dxmi_gn_block_stats"))
    return fold_stats(st) if P > MAX_APPLY_PARTIALS else st


GN_APPLY_SPLIT = os.environ.get("DXMI_GN_APPLY_SPLIT", "0") == "1"      # dxmi_groupnorm_apply_split: finalize launch + prologue-free apply


def groupnorm_apply(x, st, gamma, beta, *, in1=None, st1=None, groups=32, eps=1e-6, silu=True, out=None, scale_shift=None):
    """GroupNorm(+FiLM scale-shift)(+SiLU) of [x | in1] given their block statistics: one streaming read + write
    (dxmi_groupnorm_apply)."""
    _need_cuda(x, in1, gamma, beta, out, st.buf, st1.buf if st1 is not None else None, scale_shift)
    N, H, W, C0 = x.shape
    C1 = in1.shape[3] if in1 is not None else 0
    C = C0 + C1
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and (in1 is None or (in1.is_contiguous() and st1 is not None))
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C
    assert tuple(st.buf.shape) == (N, st.P, C0 // 2, 2) and (st1 is None or tuple(st1.buf.shape) == (N, st1.P, C1 // 2, 2))
    ss_ld = 0
    if scale_shift is not None:
        assert scale_shift.dtype == torch.float32 and scale_shift.stride(-1) == 1 and scale_shift.shape[1] == 2 * C
        ss_ld = scale_shift.stride(0)
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    if GN_APPLY_SPLIT:
        ab = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
        _prof("groupnorm", "apply", 0.0, 4.0 * N * H * W * C, lambda: check(
            load().dxmi_groupnorm_apply_split(_ptr(x), C0, _ptr(st.buf), st.P, _ptr(in1), C1, _ptr(st1.buf) if st1 is not None else None,
                                              st1.P if st1 is not None else 0, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld, _ptr(out),
                                              _ptr(ab), N, H * W, groups, float(eps), int(silu), _stream()), "dxmi_groupnorm_apply_split"))
        return out
    _prof("groupnorm", "apply", 0.0, 4.0 * N * H * W * C, lambda: check(
        load().dxmi_groupnorm_apply(_ptr(x), C0, _ptr(st.buf), st.P, _ptr(in1), C1, _ptr(st1.buf) if st1 is not None else None,
                                    st1.P if st1 is not None else 0, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld, _ptr(out),
                                    N, H * W, groups, float(eps), int(silu), _stream()), "dxmi_groupnorm_apply"))
    return out


def groupnorm_silu(x, gamma, beta, *, in1=None, groups=32, eps=1e-6, silu=True, out=None, scale_shift=None, stats=None, saved=None):
    """GroupNorm(+SiLU).  stats = (BlockStats of x, BlockStats of in1 | None): the streaming apply kernel.  saved: a list that
    receives the statistics tensor groupnorm_generic_bwd(fwd_stats=...) takes, on the streaming and the generic path.  Otherwise the
    register-resident one-pass kernel serves channels-per-group % 4 == 0 slices that fit; everything else (EDM shapes,
    scale-shift norm) goes to the generic two-kernel path."""
    _need_cuda(x, in1, gamma, beta, out, scale_shift)
    N, H, W, C0 = x.shape
    C1 = in1.shape[3] if in1 is not None else 0
    if stats is not None and stats[0] is not None and (in1 is None or stats[1] is not None) \
            and ((C0 + C1) // groups) % 2 == 0 and C0 % 8 == 0 and C1 % 8 == 0:
        if saved is not None:       # the training forward: the same sums in the generic backward's format
            saved.append(gn_blockstats_to_generic(stats[0], stats[1] if in1 is not None else None, N, H * W, C0, C1, groups))
        return groupnorm_apply(x, stats[0], gamma, beta, in1=in1, st1=stats[1] if in1 is not None else None, groups=groups,
                               eps=eps, silu=silu, out=out, scale_shift=scale_shift)
    if scale_shift is not None or not load().dxmi_groupnorm_silu_supported(C0, C1, H * W, groups):
        return groupnorm_generic(x, gamma, beta, in1=in1, groups=groups, eps=eps, silu=silu, out=out, scale_shift=scale_shift, saved=saved)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C0 + C1
    if out is None:
        out = torch.empty((N, H, W, C0 + C1), dtype=torch.bfloat16, device=x.device)
    _prof("groupnorm", "resident", 0.0, 4.0 * N * H * W * (C0 + C1), lambda: check(
        load().dxmi_groupnorm_silu_fwd(_ptr(x), C0, _ptr(in1), C1, _ptr(gamma), _ptr(beta), _ptr(out), N, H * W,
                                       groups, float(eps), int(silu), _stream()), "dxmi_groupnorm_silu_fwd"))
    return out


def gn_blockstats_to_generic(st0, st1, N, HW, C0, C1, groups=32):
    """BlockStats of x (| in1) -> the fp32 statistics tensor groupnorm_generic_bwd(fwd_stats=...) takes (dxmi_gn_blockstats_to_generic)."""
    lib = load()
    ws = torch.empty(lib.dxmi_groupnorm_generic_workspace_bytes(N, HW, C0 + C1) // 4, dtype=torch.float32, device=st0.buf.device)
    check(lib.dxmi_gn_blockstats_to_generic(_ptr(st0.buf), st0.P, C0, _ptr(st1.buf) if st1 is not None else None, st1.P if st1 is not None else 0,
                                            C1, _ptr(ws), N, HW, groups, _stream()), "dxmi_gn_blockstats_to_generic")
    return ws


def groupnorm_generic(x, gamma, beta, *, in1=None, groups=32, eps=1e-5, silu=True, out=None, scale_shift=None, saved=None):
    """saved: a list; the forward's statistics partials (a small fp32 tensor of its own instead of the shared workspace) are
    appended to it for groupnorm_generic_bwd(fwd_stats=...), which then skips its statistics pass over the input."""
    _need_cuda(x, in1, gamma, beta, out, scale_shift)
    N, H, W, C0 = x.shape
    C1 = in1.shape[3] if in1 is not None else 0
    C = C0 + C1
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    ss_ld = 0
    if scale_shift is not None:
        assert scale_shift.dtype == torch.float32 and scale_shift.stride(-1) == 1 and scale_shift.shape[1] == 2 * C
        ss_ld = scale_shift.stride(0)
    lib = load()
    nbytes = lib.dxmi_groupnorm_generic_workspace_bytes(N, H * W, C)
    if saved is not None:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        saved.append(ws)
    else:
        ws = _workspace(nbytes, x.device)
    _prof("groupnorm", "generic", 0.0, 4.0 * N * H * W * C, lambda: check(
        lib.dxmi_groupnorm_generic_fwd(_ptr(x), C0, _ptr(in1), C1, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld, _ptr(out),
                                       _ptr(ws), N, H * W, groups, float(eps), int(silu), _stream()), "dxmi_groupnorm_generic_fwd"))
    return out


def groupnorm_generic_bwd(x, dy, gamma, beta, *, in1=None, add0=None, add1=None, groups=32, eps=1e-5, silu=True, scale_shift=None,
                          fwd_stats=None):
    """-> (dx0, dx1 | None, dgamma [C], dbeta [C], d_scale_shift [N, 2C] | None).  fwd_stats: the tensor groupnorm_generic(saved=...)
    kept for this input."""
    _need_cuda(x, in1, dy, add0, add1, gamma, beta, scale_shift, fwd_stats)
    N, H, W, C0 = x.shape
    C1 = in1.shape[3] if in1 is not None else 0
    C = C0 + C1
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and tuple(dy.shape) == (N, H, W, C)
    dx0 = torch.empty_like(x)
    dx1 = torch.empty_like(in1) if in1 is not None else None
    g = torch.empty((2, N, C), dtype=torch.float32, device=x.device)
    ss_ld = scale_shift.stride(0) if scale_shift is not None else 0
    lib = load()
    ws = _workspace(lib.dxmi_groupnorm_generic_bwd_workspace_bytes(N, H * W, C), x.device)
    if fwd_stats is not None:
        assert fwd_stats.dtype == torch.float32 and fwd_stats.numel() * 4 == lib.dxmi_groupnorm_generic_workspace_bytes(N, H * W, C)
    check(lib.dxmi_groupnorm_generic_bwd_saved(_ptr(x), C0, _ptr(in1), C1, _ptr(dy), _ptr(add0), _ptr(add1), _ptr(gamma), _ptr(beta),
                                               _ptr(scale_shift), ss_ld, _ptr(dx0), _ptr(dx1), _ptr(g), _ptr(fwd_stats), _ptr(ws), N, H * W,
                                               groups, float(eps), int(silu), _stream()), "dxmi_groupnorm_generic_bwd")
    if scale_shift is None:
        red = colsum_f32(g)          # [2, N, C] -> [2, C]: both parameter gradients from ONE launch (fixed order over the images)
        return dx0, dx1, red[1], red[0], None
    d_ss = torch.empty((N, 2 * C), dtype=torch.float32, device=x.device)
    dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
    check(lib.dxmi_gn_ss_grads(_ptr(g), _ptr(scale_shift), ss_ld, _ptr(gamma), _ptr(beta), _ptr(d_ss), _ptr(dgb[0]), _ptr(dgb[1]), N, C,
                               _stream()), "dxmi_gn_ss_grads")
    return dx0, dx1, dgb[0], dgb[1], d_ss


def upsample2x(x, out=None):
    _need_cuda(x, out)
    N, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if out is None:
        out = torch.empty((N, 2 * H, 2 * W, C), dtype=torch.bfloat16, device=x.device)
    check(load().dxmi_upsample2x(_ptr(x), _ptr(out), N, H, W, C, _stream()), "dxmi_upsample2x")
    return out


def edm_precond(x, sigma, sigma_data=0.5):
    _need_cuda(x, sigma)
    N = x.shape[0]
    x_in = torch.empty_like(x)
    t = torch.empty(N, dtype=torch.float32, device=x.device)
    check(load().dxmi_edm_precond(_ptr(x), _ptr(sigma), _ptr(x_in), _ptr(t), N, x.numel() // N, float(sigma_data), _stream()),
          "dxmi_edm_precond")
    return x_in, t


def edm_step(x, model_out, z, sigma, sigma_down, sigma_up, sigma_data=0.5, outs=None):
    _need_cuda(x, model_out, z, sigma, sigma_down, sigma_up)
    N = x.shape[0]
    sample, mean = (torch.empty_like(x), torch.empty_like(x)) if outs is None else outs
    assert sample.is_contiguous() and mean.is_contiguous() and sample.shape == x.shape and mean.shape == x.shape
    check(load().dxmi_edm_step_fwd(_ptr(x), _ptr(model_out), _ptr(z), _ptr(sigma), _ptr(sigma_down), _ptr(sigma_up), _ptr(sample),
                                   _ptr(mean), N, x.numel() // N, float(sigma_data), _stream()), "dxmi_edm_step_fwd")
    return sample, mean


def attention(qkv, heads, scale, out=None, want_lse=False):
    """qkv: [N, T, 3C] bf16 laid out [q|k|v]; returns [N, T, C] bf16.  want_lse: -> (out, lse | None): the row log-sum-exp the
    kernel leaves for attention_bwd(lse=...) (fp32 [N, heads, T], log2 domain; None for the shapes that have no such kernel)."""
    _need_cuda(qkv, out)
    N, T, C3 = qkv.shape
    C = C3 // 3
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    if out is None:
        out = torch.empty((N, T, C), dtype=torch.bfloat16, device=qkv.device)
    lib = load()
    if want_lse and lib.dxmi_attention_fwd_lse_supported(T, C, heads):
        lse = torch.empty((N, heads, T), dtype=torch.float32, device=qkv.device)
        _prof("attention", f"T{T}_D{C // heads}", 4.0 * N * T * T * C, 2.0 * N * T * 4 * C, lambda: check(
            lib.dxmi_attention_fwd_lse(_ptr(qkv), _ptr(out), _ptr(lse), N, T, C, heads, float(scale), _stream()), "dxmi_attention_fwd_lse"))
        return out, lse
    _prof("attention", f"T{T}_D{C // heads}", 4.0 * N * T * T * C, 2.0 * N * T * 4 * C, lambda: check(
        lib.dxmi_attention_fwd(_ptr(qkv), _ptr(out), N, T, C, heads, float(scale), _stream()), "dxmi_attention_fwd"))
    return (out, None) if want_lse else out


def attention_proj_supported(T, C, heads):
    return bool(load().dxmi_attention_proj_supported(T, C, heads))


def pack_attn_proj_weight(w):
    """proj_out weight [256, 256(, 1, 1)] fp32 -> the 128 KiB bf16 fragment image dxmi_attention_proj_fwd reads."""
    _need_cuda(w)
    w2 = w.detach().float().reshape(w.shape[0], -1).contiguous()
    assert tuple(w2.shape) == (256, 256)
    dst = pack_tensor(256 * 256, torch.bfloat16, w.device)
    check(load().dxmi_pack_attn_proj_weight(_ptr(w2), _ptr(dst), _stream()), "dxmi_pack_attn_proj_weight")
    return dst


def attention_proj(qkv, wproj_packed, bias, residual, heads, scale, out=None, want_stats=False):
    """x + proj_out(attention(qkv)) + bias in one launch (see include/dxmi_hip.h).  qkv [N,T,3C], residual [N,T,C] bf16.
    want_stats: -> (out, BlockStats of out) from the same launch."""
    _need_cuda(qkv, wproj_packed, bias, residual, out)
    N, T, C3 = qkv.shape
    C = C3 // 3
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and residual.dtype == torch.bfloat16 and residual.is_contiguous()
    assert residual.numel() == N * T * C and bias.dtype == torch.float32 and bias.numel() == C
    if out is None:
        out = torch.empty((N, T, C), dtype=torch.bfloat16, device=qkv.device)
    stats = BlockStats(torch.empty((N, 8, C // 2, 2), dtype=torch.float32, device=qkv.device), 8) if want_stats else None
    _prof("attention", f"T{T}_D{C // heads}_proj", 4.0 * N * T * T * C + 2.0 * N * T * C * C, 2.0 * N * T * 5 * C, lambda: check(
        load().dxmi_attention_proj_fwd(_ptr(qkv), _ptr(wproj_packed), _ptr(bias), _ptr(residual), _ptr(out),
                                       stats.buf.data_ptr() if want_stats else None, N, T, C, heads, float(scale), _stream()),
        "dxmi_attention_proj_fwd"))
    return (out, stats) if want_stats else out


def attn_block_supported(T, C, heads, groups=32):
    return bool(load().dxmi_attn_block_supported(T, C, heads, groups))


def attn_block_pack(wq, bq, wk, wv, bv, wproj, bproj, scale):
    """Folded + packed weights of a whole AttnBlock (include/dxmi_hip.h: dxmi_attn_block_pack): G = scale Wk^T Wq, W' = Wproj Wv,
    g = scale Wk^T bq, b' = bproj + Wproj bv.  The k bias has no effect on the block (constant along the key axis of the softmax)."""
    ws = [t.detach().float().reshape(t.shape[0], -1).contiguous() for t in (wq, wk, wv, wproj)]
    bs = [t.detach().float().contiguous() for t in (bq, bv, bproj)]
    _need_cuda(*ws, *bs)
    assert all(tuple(w.shape) == (256, 256) for w in ws) and all(b.numel() == 256 for b in bs)
    dst = pack_tensor(int(load().dxmi_attn_block_packed_bytes()), torch.uint8, ws[0].device)
    check(load().dxmi_attn_block_pack(_ptr(ws[0]), _ptr(bs[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(bs[1]), _ptr(ws[3]), _ptr(bs[2]),
                                      float(scale), _ptr(dst), _stream()), "dxmi_attn_block_pack")
    return dst


def attn_block(x, stats, gamma, beta, packed, eps=1e-6, out=None, want_stats=False):
    """x + proj_out(attention(q, k, v of GroupNorm(x))) in ONE launch (reference unet_small.py:167-191; dxmi_attn_block_fwd).
    x [N,16,16,256] or [N,256,256] bf16, stats = its BlockStats; want_stats: -> (out, BlockStats of out)."""
    _need_cuda(x, stats.buf, gamma, beta, packed, out)
    shape = x.shape
    N, C = shape[0], shape[-1]
    T = x.numel() // (N * C)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and tuple(stats.buf.shape) == (N, stats.P, C // 2, 2)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
    if stats.P > MAX_APPLY_PARTIALS:
        stats = fold_stats(stats)
    if out is None:
        out = torch.empty_like(x)
    st = BlockStats(torch.empty((N, 8, C // 2, 2), dtype=torch.float32, device=x.device), 8) if want_stats else None
    _prof("attention", f"block_T{T}_C{C}", 8.0 * N * T * C * C, 4.0 * N * T * C, lambda: check(
        load().dxmi_attn_block_fwd(_ptr(x), _ptr(stats.buf), stats.P, _ptr(gamma), _ptr(beta), float(eps), _ptr(packed), _ptr(out),
                                   st.buf.data_ptr() if want_stats else None, N, T, C, _stream()), "dxmi_attn_block_fwd"))
    return (out, st) if want_stats else out


def timestep_embedding(t, dim, order=0, max_period=10000.0, out=None):
    _need_cuda(t, out)
    t = t.float().contiguous()
    if out is None:
        out = torch.empty((t.numel(), dim), dtype=torch.float32, device=t.device)
    check(load().dxmi_timestep_embedding(_ptr(t), _ptr(out), t.numel(), dim, order, float(max_period), _stream()),
          "dxmi_timestep_embedding")
    return out


def linear(x, pw, bias=None, pre_act=ACT_NONE, post_act=ACT_NONE, out=None, splitk=False):
    """x fp32 [P, K] -> fp32 [P, M]; pw = pack_conv_weight(W[M, K]).
    splitk: allow the split-K form for skinny products with a long K.  Its slice count depends on the number of rows, so the
    fp32 summation order of a row would change with the batch it rides in: only the training backward (linear_bwd) asks for
    it; forward calls keep one reduction order per shape (bitwise batch independence of the generation path)."""
    _need_cuda(x, bias, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == pw.Cin
    P, K = x.shape
    S = load().dxmi_linear_splitk_slices(P, K, pw.Cout) if (splitk and post_act == ACT_NONE) else 1
    if S > 1:      # skinny product with a long K: slices of K in parallel, summed in slice order (deterministic)
        part = torch.empty((S, P, pw.Cout), dtype=torch.float32, device=x.device)
        check(load().dxmi_linear_splitk(_ptr(x), _ptr(pw.buf), _ptr(part), P, K, pw.Cout, pre_act, _stream()), "dxmi_linear_splitk")
        res = part.sum(0) if bias is None else part.sum(0).add_(bias)
        return res if out is None else out.copy_(res)
    if out is None:
        out = torch.empty((P, pw.Cout), dtype=torch.float32, device=x.device)
    check(load().dxmi_linear_fwd(_ptr(x), _ptr(pw.buf), _ptr(bias), _ptr(out), P, K, pw.Cout, pre_act, post_act,
                                 _stream()), "dxmi_linear_fwd")
    return out


def _rows_as_map(t_f32, rows_pad):
    """fp32 [P, K] -> bf16 [1, rows_pad / 16, 16, K] (zero rows appended): the pixel-GEMM operand form of a dense layer's rows."""
    P, K = t_f32.shape
    out = torch.zeros((rows_pad, K), dtype=torch.bfloat16, device=t_f32.device)
    out[:P] = t_f32
    return out.view(1, rows_pad // 16, 16, K)


def linear_bwd(x, dy, pw_t, need_dx=True):
    """Backward of y = x @ W^T + b for the small dense layers of the embedding graphs (temb MLP, temb_proj / emb_layers,
    reference unet_small.py:296-299,123 and models/cm/unet.py:775-779,249) on the HIP kernels — no BLAS library call:
      dx = dy @ W            dxmi_linear_fwd on the transposed pack (pw_t = pack_conv_weight(W, transpose_flip=True))
      dW = dy^T @ x          dxmi_conv2d_wgrad with ksize 1: the rows are the 'pixels' of a 1x1 conv (zero rows pad the count)
      db = column sums of dy
    x [P, K], dy [P, M] fp32 (K, M multiples of 64) -> (dx [P, K] | None, dW [M, K], db [M]) fp32; bf16 MFMA operands as the
    forward's."""
    _need_cuda(x, dy)
    P, K = x.shape
    M = dy.shape[1]
    assert dy.shape[0] == P and x.dtype == torch.float32 and dy.dtype == torch.float32
    dy = dy.contiguous()
    dx = linear(dy, pw_t, splitk=True) if need_dx else None
    rows = 64
    while rows < P:
        rows *= 2
    dw = conv2d_wgrad(_rows_as_map(x, rows), _rows_as_map(dy, rows), 1)          # [M, K, 1, 1]
    return dx, dw.view(M, K), dy.sum(0)


def silu_bwd(pre, g):
    """g * d/dpre [pre * sigmoid(pre)] (elementwise, fp32)."""
    sg = torch.sigmoid(pre)
    return g * (sg * (1 + pre * (1 - sg)))


def var_gather_sched(t, continuous_steps, xmul_tab, cmul_tab, log_betas_all, sigma_out=None):
    _need_cuda(t, continuous_steps, xmul_tab, cmul_tab, log_betas_all, sigma_out)
    assert t.dtype == torch.int64
    N, T = t.numel(), continuous_steps.numel()
    outs = [torch.empty(N, dtype=torch.float32, device=t.device) for _ in range(3)]
    outs.append(sigma_out if sigma_out is not None else torch.empty(N, dtype=torch.float32, device=t.device))
    assert outs[3].is_contiguous() and outs[3].numel() == N and outs[3].dtype == torch.float32
    check(load().dxmi_var_gather_sched(_ptr(t), _ptr(continuous_steps), _ptr(xmul_tab), _ptr(cmul_tab),
                                       _ptr(log_betas_all), *[_ptr(o) for o in outs], N, T, _stream()),
          "dxmi_var_gather_sched")
    return outs  # tau, xmul, cmul, sigma


def var_step(x, eps, z, xmul, cmul, sigma, want_mean=True, want_control=True, outs=None, assoc=0):
    """Fused sampler transition; all tensors fp32, x/eps/z [N,C,H,W], per-sample scalars [N]."""
    _need_cuda(x, eps, z, xmul, cmul, sigma)
    N = x.shape[0]
    CHW = x.numel() // N
    for t_ in (x, eps, z):
        assert t_.dtype == torch.float32 and t_.is_contiguous()
    if outs is None:
        x_next = torch.empty_like(x)
        mean = torch.empty_like(x) if want_mean else None
        control = torch.empty_like(x) if want_control else None
        logp = torch.empty(N, dtype=torch.float32, device=x.device)
    else:
        x_next, mean, control, logp = outs
    nb = 4.0 * N * CHW * (3 + 1 + (mean is not None) + (control is not None))
    _prof("sampler_step", "var_step", 0.0, nb, lambda: check(
        load().dxmi_var_step_fwd(_ptr(x), _ptr(eps), _ptr(z), _ptr(xmul), _ptr(cmul), _ptr(sigma), _ptr(x_next),
                                 _ptr(mean), _ptr(control), _ptr(logp), N, CHW, assoc, _stream()), "dxmi_var_step_fwd"))
    return x_next, mean, control, logp


def var_step_bwd(g_next, g_mean, g_control, g_logp, z, cmul, sigma):
    """Backward of the VAR transition (dxmi_var_step_bwd): -> (d eps [N, ...], d sigma [N]); any gradient may be None."""
    _need_cuda(g_next, g_mean, g_control, g_logp, z, cmul, sigma)
    N = z.shape[0]
    CHW = z.numel() // N
    for t_ in (g_next, g_mean, g_control, z):
        assert t_ is None or (t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == N * CHW)
    assert g_logp is None or (g_logp.dtype == torch.float32 and g_logp.is_contiguous() and g_logp.numel() == N)
    d_eps = torch.empty_like(z)
    d_sigma = torch.empty(N, dtype=torch.float32, device=z.device)
    check(load().dxmi_var_step_bwd(_ptr(g_next), _ptr(g_mean), _ptr(g_control), _ptr(g_logp), _ptr(z), _ptr(cmul), _ptr(sigma), _ptr(d_eps),
                                   _ptr(d_sigma), N, CHW, _stream()), "dxmi_var_step_bwd")
    return d_eps, d_sigma


def edm_step_bwd(g_sample, g_mean, z, sigma, sigma_down, sigma_data=0.5):
    """Backward of the EDM transition (dxmi_edm_step_bwd): -> (d model_output, d sigma_up [N])."""
    _need_cuda(g_sample, g_mean, z, sigma, sigma_down)
    N = z.shape[0]
    CHW = z.numel() // N
    for t_ in (g_sample, g_mean, z):
        assert t_ is None or (t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == N * CHW)
    d_out = torch.empty_like(z)
    d_up = torch.empty(N, dtype=torch.float32, device=z.device)
    check(load().dxmi_edm_step_bwd(_ptr(g_sample), _ptr(g_mean), _ptr(z), _ptr(sigma), _ptr(sigma_down), _ptr(d_out), _ptr(d_up), N, CHW,
                                   float(sigma_data), _stream()), "dxmi_edm_step_bwd")
    return d_out, d_up


def pool_act(x, pool, act, out=None):
    _need_cuda(x, out)
    N, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if out is None:
        out = torch.empty((N, H // 2, W // 2, C) if pool else (N, H, W, C), dtype=torch.bfloat16, device=x.device)
    check(load().dxmi_pool_act(_ptr(x), _ptr(out), N, H, W, C, int(pool), act, _stream()), "dxmi_pool_act")
    return out


def value_head(x, w, b, out_w=None, out_b=None, out=None):
    """out_w / out_b: device scalars of the Linear(1,1) out_scale, or None."""
    _need_cuda(x, w, b, out_w, out_b, out)
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, 1), dtype=torch.float32, device=x.device)
    check(load().dxmi_value_head(_ptr(x), _ptr(w), _ptr(b), _ptr(out_w), _ptr(out_b), _ptr(out), N, H * W, C,
                                 _stream()), "dxmi_value_head")
    return out


def stem_conv_wgrad(x_nchw, dy):
    """Weight gradient of a 3-channel 3x3/s1/p1 image conv: x [N,3,H,W] fp32, dy [N,H,W,Cout] bf16 ->
    fp32 [Cout,3,3,3] (im2col to K=27(+37 zero) then the 1x1 MFMA pixel-GEMM)."""
    _need_cuda(x_nchw, dy)
    N, C, H, W = x_nchw.shape
    assert C == 3 and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    with wgrad_branch((x_nchw, dy)):                    # (im2col, the pixel-GEMM and the slice: one branch of a captured step)
        cols = torch.empty((N, H, W, 64), dtype=torch.bfloat16, device=x_nchw.device)
        check(load().dxmi_im2col27(_ptr(x_nchw), _ptr(cols), N, H, W, _stream()), "dxmi_im2col27")
        dw = _conv2d_wgrad(cols, dy, 1)                 # [Cout, 64, 1, 1]
        return dw[:, :27, 0, 0].reshape(dy.shape[3], 3, 3, 3).contiguous()


def quantize_u8(x, mode=0, nhwc=True, out=None):
    """Output stage: NCHW fp32 [N,C,H,W] -> uint8 [N,H,W,C] (nhwc) or [N,C,H,W]; mode 0 = rescale + save_image rounding
    (generate_cifar10.py:205-209), mode 1 = (x + 1) * 127.5 truncated (generate_large.py:43).  Bit-exact pixel values."""
    _need_cuda(x, out)
    N, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty((N, H, W, C) if nhwc else (N, C, H, W), dtype=torch.uint8, device=x.device)
    assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == x.numel()
    _prof("output_stage", f"quantize{mode}", 0.0, 5.0 * x.numel(), lambda: check(
        load().dxmi_quantize_u8(_ptr(x), _ptr(out), N, C, H * W, mode, int(nhwc), _stream()), "dxmi_quantize_u8"))
    return out


def fid_stats(act):
    """Activation statistics of the FID: act fp32 [N, D] (device) -> (mu fp64 [D], sigma fp64 [D, D]) = np.mean(act, axis=0),
    np.cov(act, rowvar=False) of train_image_large.py:62-69, on the f32-input MFMA (csrc/fid_stats.hip)."""
    _need_cuda(act)
    assert act.dim() == 2 and act.dtype == torch.float32 and act.is_contiguous()
    N, D = act.shape
    lib = load()
    mu = torch.empty(D, dtype=torch.float64, device=act.device)
    sigma = torch.empty((D, D), dtype=torch.float64, device=act.device)
    ws = _workspace(max(256, lib.dxmi_fid_stats_workspace_bytes(N, D)), act.device)
    _prof("fid_stats", f"d{D}", float(N) * D * (D + 128), 4.0 * act.numel() * ((D + 127) // 128 + 1) / 2, lambda: check(
        lib.dxmi_fid_stats(_ptr(act), N, D, _ptr(mu), _ptr(sigma), _ptr(ws), _stream()), "dxmi_fid_stats"))
    return mu, sigma


def nchw_f32_to_nhwc_bf16(x, out=None):
    _need_cuda(x, out)
    N, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    check(load().dxmi_nchw_f32_to_nhwc_bf16(_ptr(x), _ptr(out), N, C, H * W, _stream()), "dxmi_nchw_f32_to_nhwc_bf16")
    return out


def nhwc_bf16_to_nchw_f32(x, out=None):
    _need_cuda(x, out)
    N, H, W, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if out is None:
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    check(load().dxmi_nhwc_bf16_to_nchw_f32(_ptr(x), _ptr(out), N, C, H * W, _stream()), "dxmi_nhwc_bf16_to_nchw_f32")
    return out


# ------------------------------------------------------------------------------------------ train-step tail
def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _numel_array(tensors):
    return (ctypes.c_int64 * len(tensors))(*[t.numel() for t in tensors])


def _need_f32_dense(*lists):
    for ts in lists:
        for t in ts:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise _lib.DxmiError("multi-tensor kernels take contiguous fp32 device tensors")


def adam_step(params, grads, exp_avgs, exp_avg_sqs, step_sizes, beta1, beta2, eps, bc2_sqrt, grad_scale=None,
              write_back_grad=False, cache=None, hyper=None):
    """One torch.optim.Adam update of every tensor in the lists (see dxmi_adam_step).  step_sizes: python floats
    -(lr_i / (1 - beta1^t)).  cache: dict reused across calls for the pointer arrays of params / moments (stable).
    hyper: DEVICE fp32 [1 + n] = (bc2_sqrt, step sizes) instead of the two host arguments (dxmi_adam_step_dev: a step that is
    replayed from a hipGraph)."""
    _need_f32_dense(params, grads, exp_avgs, exp_avg_sqs)
    n = len(params)
    if cache is None or cache.get("n") != n:
        c = {"n": n, "p": _ptr_array(params), "m": _ptr_array(exp_avgs), "v": _ptr_array(exp_avg_sqs), "numel": _numel_array(params)}
        if cache is not None:
            cache.update(c)
        cache = c
    if "elems" not in cache:
        cache["elems"] = sum(p.numel() for p in params)
    garr = _ptr_array(grads)
    if hyper is not None:
        assert hyper.is_cuda and hyper.dtype == torch.float32 and hyper.numel() == n + 1
        check(load().dxmi_adam_step_dev(cache["p"], garr, cache["m"], cache["v"], cache["numel"], n, beta1, beta2, eps, _ptr(hyper),
                                        _ptr(grad_scale), int(write_back_grad), _stream()), "dxmi_adam_step_dev")
        return
    ss = (ctypes.c_float * n)(*step_sizes)
    _prof("optimizer", "adam", 0.0, 28.0 * cache["elems"], lambda: check(
        load().dxmi_adam_step(cache["p"], garr, cache["m"], cache["v"], cache["numel"], ss, n, beta1, beta2, eps,
                              bc2_sqrt, _ptr(grad_scale), int(write_back_grad), _stream()), "dxmi_adam_step"))


def radam_step(params, grads, exp_avgs, exp_avg_sqs, lrs, beta1, beta2, eps, bc1, bc2_sqrt, rect, grad_scale=None,
               found_inf=None, cache=None, hyper=None):
    """One torch.optim.RAdam update (see dxmi_radam_step); rect < 0 selects the un-rectified branch.
    hyper: DEVICE fp32 [3 + n] = (fp32(1 / bc1), bc2_sqrt, rect, lrs) instead of the host arguments (dxmi_radam_step_dev)."""
    _need_f32_dense(params, grads, exp_avgs, exp_avg_sqs)
    n = len(params)
    if cache is None or cache.get("n") != n:
        c = {"n": n, "p": _ptr_array(params), "m": _ptr_array(exp_avgs), "v": _ptr_array(exp_avg_sqs), "numel": _numel_array(params)}
        if cache is not None:
            cache.update(c)
        cache = c
    if hyper is not None:
        assert hyper.is_cuda and hyper.dtype == torch.float32 and hyper.numel() == n + 3
        check(load().dxmi_radam_step_dev(cache["p"], _ptr_array(grads), cache["m"], cache["v"], cache["numel"], n, beta1, beta2, eps,
                                         _ptr(hyper), _ptr(grad_scale), _ptr(found_inf), _stream()), "dxmi_radam_step_dev")
        return
    lr = (ctypes.c_float * n)(*lrs)
    check(load().dxmi_radam_step(cache["p"], _ptr_array(grads), cache["m"], cache["v"], cache["numel"], lr, n, beta1, beta2, eps,
                                 bc1, bc2_sqrt, rect, _ptr(grad_scale), _ptr(found_inf), _stream()), "dxmi_radam_step")


def gradnorm_clip(grads, max_norm, scale_in_place=True, out=None):
    """Global L2 norm of `grads` and the clip coefficient, on the device: returns fp32 [3] = (norm, coef, non-finite flag).
    max_norm <= 0: norm only."""
    _need_f32_dense(grads)
    lib = load()
    numel = _numel_array(grads)
    nb = lib.dxmi_mt_blocks(numel, len(grads))
    dev = grads[0].device
    ws = _workspace(int(nb) * 4 + 64, dev)
    if out is None:
        out = torch.empty(3, dtype=torch.float32, device=dev)
    check(lib.dxmi_gradnorm_clip(_ptr_array(grads), numel, len(grads), float(max_norm), _ptr(ws), _ptr(out),
                                 int(scale_in_place and max_norm > 0), _stream()), "dxmi_gradnorm_clip")
    return out


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ without the host round trip: returns the total norm as a DEVICE scalar."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return None
    return gradnorm_clip(grads, max_norm)[0]


def dropout(x, p, seed, out=None):
    """Counter-hash dropout of a bf16 tensor (see dxmi_dropout_bf16); call again with the same seed on the gradient.
    seed: python int, or a DEVICE int32 tensor holding the 32-bit seed (dxmi_dropout_bf16_dev: hipGraph-replayed steps)."""
    _need_cuda(x, out)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    if torch.is_tensor(seed):
        assert seed.is_cuda and seed.dtype == torch.int32 and seed.numel() == 1
        check(load().dxmi_dropout_bf16_dev(_ptr(x), _ptr(out), x.numel(), float(p), _ptr(seed), _stream()), "dxmi_dropout_bf16_dev")
        return out
    check(load().dxmi_dropout_bf16(_ptr(x), _ptr(out), x.numel(), float(p), int(seed) & 0xFFFFFFFF, _stream()), "dxmi_dropout_bf16")
    return out


def gather_rows(src, idx, out=None):
    """out[r] = src[idx[r]] along dim 0 (INT path of the replay buffer); idx int64 on the device."""
    _need_cuda(src, idx, out)
    assert src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous() and idx.dim() == 1
    n = idx.numel()
    row_bytes = (src.numel() // max(src.shape[0], 1)) * src.element_size()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    assert out.is_contiguous() and out.dtype == src.dtype and out.numel() * out.element_size() == n * row_bytes
    if n:
        _prof("replay_gather", f"row{row_bytes}", 0.0, 2.0 * n * row_bytes, lambda: check(
            load().dxmi_gather_rows(_ptr(src), _ptr(idx), _ptr(out), n, src.shape[0], row_bytes, _stream()), "dxmi_gather_rows"))
    return out


# ------------------------------------------------------------------------------------------ InceptionV3 of the FID (f4)
class PackedGConv:
    """BatchNorm-folded bf16 weights [ceil32(Cout)][KH * KW][ceil16(Cin)] + fp32 bias of one BasicConv2d (dxmi_gconv_pack)."""

    __slots__ = ("w", "bias", "Cout", "Cin", "CinP", "KH", "KW")

    def __init__(self, w, bias, Cout, Cin, CinP, KH, KW):
        self.w, self.bias, self.Cout, self.Cin, self.CinP, self.KH, self.KW = w, bias, Cout, Cin, CinP, KH, KW


def gconv_pack(weight, bn=None, eps=1e-3):
    """weight fp32 [Cout, Cin, KH, KW]; bn = (gamma, beta, running_mean, running_var) or None."""
    _need_cuda(weight)
    w = weight.detach().float().contiguous()
    Cout, Cin, KH, KW = w.shape
    lib = load()
    wp = torch.empty(int(lib.dxmi_gconv_packed_elems(Cout, Cin, KH, KW)), dtype=torch.bfloat16, device=w.device)
    CoutP, CinP = (Cout + 31) // 32 * 32, (Cin + 15) // 16 * 16
    bias = torch.empty(CoutP, dtype=torch.float32, device=w.device)
    bnp = [t.detach().float().contiguous() for t in bn] if bn is not None else [None] * 4
    check(lib.dxmi_gconv_pack(_ptr(w), _ptr(bnp[0]), _ptr(bnp[1]), _ptr(bnp[2]), _ptr(bnp[3]), float(eps), _ptr(wp), _ptr(bias), Cout, Cin, KH, KW,
                              _stream()), "dxmi_gconv_pack")
    return PackedGConv(wp, bias, Cout, Cin, CinP, KH, KW)


def gconv(x, pk, stride=(1, 1), pad=(0, 0), relu=True, out=None, coff=0):
    """x NHWC bf16 [N, IH, IW, pk.CinP] -> NHWC bf16 [N, OH, OW, Cout] (or the channel window [coff, coff + Cout) of `out`)."""
    _need_cuda(x, out)
    N, IH, IW, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and C == pk.CinP, f"gconv: input has {C} channels, the packed weight wants {pk.CinP}"
    OH, OW = (IH + 2 * pad[0] - pk.KH) // stride[0] + 1, (IW + 2 * pad[1] - pk.KW) // stride[1] + 1
    if out is None:
        out = torch.empty((N, OH, OW, pk.Cout), dtype=torch.bfloat16, device=x.device)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and tuple(out.shape[:3]) == (N, OH, OW)
    fl = 2.0 * N * OH * OW * pk.Cout * pk.Cin * pk.KH * pk.KW
    _prof("inception", f"gconv{pk.KH}x{pk.KW}", fl, 2.0 * (x.numel() + N * OH * OW * pk.Cout + pk.w.numel()), lambda: check(
        load().dxmi_gconv_fwd(_ptr(x), _ptr(pk.w), _ptr(pk.bias), _ptr(out), N, IH, IW, C, pk.Cout, pk.KH, pk.KW, stride[0], stride[1], pad[0],
                              pad[1], out.shape[3], coff, int(relu), _stream()), "dxmi_gconv_fwd"))
    return out


def pool3x3(x, stride, pad, avg_exclude_pad=False, out=None, coff=0):
    """3x3 max pool, or the average over the in-bounds pixels of the window (count_include_pad=False)."""
    _need_cuda(x, out)
    N, IH, IW, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    OH, OW = (IH + 2 * pad - 3) // stride + 1, (IW + 2 * pad - 3) // stride + 1
    if out is None:
        out = torch.empty((N, OH, OW, C), dtype=torch.bfloat16, device=x.device)
    check(load().dxmi_pool3x3(_ptr(x), _ptr(out), N, IH, IW, C, stride, pad, int(avg_exclude_pad), out.shape[3], coff, _stream()), "dxmi_pool3x3")
    return out


def global_avgpool(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(load().dxmi_global_avgpool(_ptr(x), _ptr(out), N, H * W, C, _stream()), "dxmi_global_avgpool")
    return out


def resize_bilinear_nhwc16(x, OH, OW, normalize=True):
    """NCHW fp32 [N, 3, H, W] -> NHWC bf16 [N, OH, OW, 16] (bilinear, align_corners=False; 2 x - 1 when normalize)."""
    _need_cuda(x)
    N, C, H, W = x.shape
    assert C == 3 and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((N, OH, OW, 16), dtype=torch.bfloat16, device=x.device)
    check(load().dxmi_resize_bilinear_nhwc16(_ptr(x), _ptr(out), N, H, W, OH, OW, int(normalize), _stream()), "dxmi_resize_bilinear_nhwc16")
    return out
