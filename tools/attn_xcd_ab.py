"""GPU box: attention forward + fused backward at the ADM nets' shapes; run with DXMI_ATTN_XCD=0/1 (block order).  Median of 10."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops


def med(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[5]


for (N, T, heads, D) in [(100, 1024, 6, 64), (100, 256, 9, 64), (100, 64, 12, 64), (16, 1024, 6, 64), (16, 256, 9, 64), (16, 4096, 8, 64)]:
    C = heads * D
    qkv = torch.randn(N, T, 3 * C, device="cuda:0").to(torch.bfloat16)
    do = torch.randn(N, T, C, device="cuda:0").to(torch.bfloat16)
    o = ops.attention(qkv, heads, D ** -0.5)
    tf = med(lambda: ops.attention(qkv, heads, D ** -0.5))
    tb = med(lambda: ops.attention_bwd(qkv, do, heads, D ** -0.5, o=o))
    print(f"N={N} T={T} heads={heads} D={D}: fwd {tf:7.1f} us {4.0 * N * heads * T * T * D / tf / 1e6:5.0f} TFLOP/s | bwd {tb:7.1f} us")
