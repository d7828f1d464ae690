"""GPU box: streaming GroupNorm(+scale-shift)(+SiLU) per shape (block statistics given): us and TB/s.  B=100 python tools/gn_apply_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
B = int(os.environ.get("B", 100))
SHAPES = [(64, 192, 0, False), (64, 192, 0, True), (64, 192, 192, False), (32, 384, 0, True), (32, 384, 384, False), (32, 384, 192, False),
          (16, 576, 0, True), (16, 576, 576, False), (8, 768, 0, True), (32, 128, 0, False), (16, 256, 0, False)]
for (H, C0, C1, ss) in SHAPES:
    C = C0 + C1
    x0 = torch.randn(B, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(B, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    s0 = ops.block_stats(x0)
    s1 = ops.block_stats(x1) if C1 else None
    if s0 is None:
        print(H, C0, C1, "no block statistics for this shape")
        continue
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    sst = torch.randn(B, 2 * C, device=dev) * 0.1 if ss else None
    out = torch.empty(B, H, H, C, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.groupnorm_apply(x0, s0, g, b, in1=x1, st1=s1, eps=1e-5, silu=True, out=out, scale_shift=sst)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"B={B} {H}x{H} {C0}+{C1} ss={int(ss)} P={s0.P}: {us:7.1f} us  {4.0 * B * H * H * C / us / 1e6:5.2f} TB/s")
