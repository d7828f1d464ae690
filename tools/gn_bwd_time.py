"""GPU box: resident GroupNorm(+SiLU) backward on the CIFAR-10 net's shapes (graph-captured); DXMI_LIB selects the library build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)


def graph_time(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def event_time(fn, n=20):
    """(the generic path's workspace is keyed on the stream: plain event timing of eager launches instead of a captured graph)"""
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("lib:", os.path.basename(os.environ.get("DXMI_LIB", "default")))
for (N, H, C0, C1) in [(256, 32, 128, 0), (256, 32, 128, 128), (256, 32, 256, 128), (256, 16, 128, 0), (256, 16, 256, 0), (256, 16, 256, 128), (256, 16, 256, 256), (256, 8, 256, 0), (256, 8, 256, 256), (256, 4, 256, 0), (256, 4, 256, 256)]:
    C = C0 + C1
    x0 = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    dy = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    add = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    g_, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    us = graph_time(lambda: ops.groupnorm_silu_bwd(x0, dy, g_, b_, in1=x1, add0=add, silu=True))
    by = 2.0 * N * H * H * (3 * C + C0)
    # the generic two-pass kernels (EDM path), with and without the forward's saved statistics
    saved = []
    ops.groupnorm_generic(x0, g_, b_, in1=x1, groups=32, eps=1e-6, silu=True, saved=saved)
    us_g = event_time(lambda: ops.groupnorm_generic_bwd(x0, dy, g_, b_, in1=x1, add0=add, groups=32, eps=1e-6, silu=True))
    us_s = event_time(lambda: ops.groupnorm_generic_bwd(x0, dy, g_, b_, in1=x1, add0=add, groups=32, eps=1e-6, silu=True, fwd_stats=saved[0]))
    print(f"  {H}x{H} {C0}+{C1}: resident {us:7.1f} us (eager {event_time(lambda: ops.groupnorm_silu_bwd(x0, dy, g_, b_, in1=x1, add0=add, silu=True)):7.1f})  {by/us/1e3:6.0f} GB/s | generic {us_g:7.1f} us | generic, saved statistics {us_s:7.1f} us")
