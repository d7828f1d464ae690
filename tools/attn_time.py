"""GPU box: attention kernel timing at the U-Net shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
for (N, T, heads, D) in [(256, 256, 1, 256), (100, 1024, 6, 64), (100, 256, 9, 64), (100, 64, 12, 64), (16, 1024, 16, 64)]:
    C = heads * D
    qkv = torch.randn(N, T, 3 * C, device="cuda:0").to(torch.bfloat16)
    for _ in range(3): ops.attention(qkv, heads, D ** -0.5)
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.attention(qkv, heads, D ** -0.5); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[5]
    print(f"N={N} T={T} heads={heads} D={D}: {t:.1f} us  {4.0 * N * heads * T * T * D / t / 1e6:.0f} TFLOP/s")
