"""GPU box: conv_ws8_kernel on the 8x8 shapes of the CIFAR-10 net (graph-captured time); DXMI_LIB selects the library build."""
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


print("lib:", os.environ.get("DXMI_LIB", "default"))
tot = 0
for (N, C0, C1, Cout, res, cnt) in [(256, 256, 0, 256, True, 5), (256, 512, 0, 256, False, 3), (256, 256, 0, 256, False, 2), (256, 384, 0, 256, False, 1)]:
    x0 = torch.randn(N, 8, 8, C0, device=dev).to(torch.bfloat16)
    pw = ops.pack_conv_weight(torch.randn(Cout, C0, 3, 3, device=dev) * 0.03)
    b = torch.randn(Cout, device=dev); v = torch.randn(N, Cout, device=dev)
    r = torch.randn(N, 8, 8, Cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(N, 8, 8, Cout, device=dev, dtype=torch.bfloat16)
    us = graph_time(lambda: ops.conv2d(x0, pw, bias=b, addvec=v, residual=r, out=out))
    tot += us * cnt
    print(f"  8x8 {C0}->{Cout} res={int(res)}: {us:6.1f} us  {2.0*N*64*Cout*C0*9/us/1e6:6.0f} TFLOP/s")
print(f"  weighted per forward: {tot:.1f} us")
