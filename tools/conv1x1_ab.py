"""GPU box: 1x1 convs of the U-Nets (graph-captured device time, correctness vs torch fp32); DXMI_CONV1X1_WS=0/1 to A/B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch, torch.nn.functional as F
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("DXMI_CONV1X1_WS =", os.environ.get("DXMI_CONV1X1_WS"))
for (N, H, C0, C1, Cout, res) in [(256, 16, 256, 0, 768, False), (256, 16, 256, 0, 256, True), (256, 32, 256, 128, 128, False), (256, 32, 128, 128, 128, False),
                                  (256, 16, 256, 256, 256, False), (256, 16, 128, 0, 256, False), (130, 16, 256, 0, 768, False), (131, 16, 256, 128, 256, True)]:
    x0 = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    w = torch.randn(Cout, C0 + C1, 1, 1, device=dev) * 0.05
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.conv2d(x0, pw, in1=x1, bias=bias, residual=r, out=out)
    y = f().float()
    xin = torch.cat([x0, x1], 3) if C1 else x0
    ref = xin.float() @ w[:, :, 0, 0].to(torch.bfloat16).float().t() + bias
    if res: ref = ref + r.float()
    rel = ((y - ref).norm() / ref.norm()).item()
    us = timeit(f)
    mb = (xin.numel() + out.numel() * (2 if res else 1)) * 2 / 1e6
    print(f"N{N} {H}x{H} {C0}+{C1}->{Cout} res={res}: rel {rel:.2e}  {us:.1f} us  {mb/us/1e3*1e3/1e3:.2f} TB/s")
