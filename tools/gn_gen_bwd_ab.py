"""GPU box: generic GroupNorm(+SiLU) backward on the EDM nets' train shapes (batch 16), one launch (gn_gen_bwd_fused_kernel) against the
reduce + apply launches (knob gn_bwd_fused), graph-captured, with the forward's saved statistics as the train step passes them.
    python tools/gn_gen_bwd_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)
N = int(os.environ.get("N", 16))


SIDE = torch.cuda.Stream()


def graph_time(fn, n=10):
    with torch.cuda.stream(SIDE):           # the library's workspace is per stream: warm the capture stream's
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=SIDE):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


tot = [0.0, 0.0]
for (H, C0, C1, ss) in [(64, 192, 0, False), (64, 192, 0, True), (64, 192, 192, False), (32, 384, 0, True), (32, 192, 0, False), (32, 384, 384, False),
                        (32, 384, 192, False), (16, 576, 0, True), (16, 384, 0, False), (16, 576, 576, False), (16, 576, 384, False),
                        (8, 768, 0, True), (8, 576, 0, False), (8, 768, 768, False), (8, 768, 576, False)]:
    C = C0 + C1
    x0 = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    dy = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    add = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16) if not C1 else None
    g_, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    sst = torch.randn(N, 2 * C, device=dev) * 0.1 if ss else None
    saved = []
    ops.groupnorm_generic(x0, g_, b_, in1=x1, eps=1e-5, silu=True, scale_shift=sst, saved=saved)
    us = []
    for fused in (0, int(os.environ.get("FUSED", 1))):      # FUSED=2: the one-launch form on every shape it fits (the default knob: <= 256-pixel maps)
        ops.set_tuning("gn_bwd_fused", fused)
        us.append(graph_time(lambda: ops.groupnorm_generic_bwd(x0, dy, g_, b_, in1=x1, add0=add, eps=1e-5, silu=True, scale_shift=sst, fwd_stats=saved[0])))
    by = 2.0 * N * H * H * (3 * C + (C0 if add is not None else 0))
    tot[0] += us[0]; tot[1] += us[1]
    print(f"  {N} x {H}x{H} {C0}+{C1} ss={int(ss)}: two launches {us[0]:7.1f} us | one launch {us[1]:7.1f} us ({by / us[1] / 1e3:6.0f} GB/s algorithmic)", flush=True)
print(f"  sum: {tot[0]:.1f} -> {tot[1]:.1f} us")
