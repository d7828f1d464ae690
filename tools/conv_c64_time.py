"""GPU box: the value net's 64 -> 64 convs (conv_c64_kernel vs conv_pipe_kernel via DXMI_CONV_C64=0 in another process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)


def graph_time(fn, n=20):
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


print("DXMI_CONV_C64 =", os.environ.get("DXMI_CONV_C64"), " OCC =", os.environ.get("DXMI_CONV_C64_OCC"))
for (N, H, res, mask) in [(256, 32, False, False), (256, 32, True, False), (256, 32, False, True), (256, 16, False, False), (256, 16, True, True)]:
    x = torch.randn(N, H, H, 64, device=dev).to(torch.bfloat16)
    pw = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
    b = torch.randn(64, device=dev)
    r = torch.randn(N, H, H, 64, device=dev).to(torch.bfloat16) if res else None
    m = torch.randn(N, H, H, 64, device=dev).to(torch.bfloat16) if mask else None
    out = torch.empty(N, H, H, 64, device=dev, dtype=torch.bfloat16)
    us = graph_time(lambda: ops.conv2d(x, pw, bias=b, residual=r, mask_src=m, mask_slope=0.2, act=ops.ACT_LEAKY02, out=out))
    by = 2.0 * N * H * H * 64 * (2 + (res is True) + (mask is True))
    print(f"N{N} {H}x{H} res={int(res)} mask={int(mask)}: {us:6.1f} us  {2.0*N*H*H*64*64*9/us/1e6:6.0f} TFLOP/s  {by/us/1e3:6.0f} GB/s")
