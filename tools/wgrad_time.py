"""GPU box: conv weight-gradient kernel on the U-Net / value-net shapes (graph-captured device time)."""
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


print("DXMI_WGRAD_WGS =", os.environ.get("DXMI_WGRAD_WGS"))
tot = 0.0
for (N, H, Cin, Cout, k, cnt) in [(256, 32, 128, 128, 3, 8), (256, 32, 256, 128, 3, 2), (256, 32, 384, 128, 3, 1), (256, 16, 256, 256, 3, 10), (256, 16, 512, 256, 3, 2),
                                  (256, 8, 256, 256, 3, 10), (256, 4, 256, 256, 3, 12), (256, 16, 256, 768, 1, 5), (256, 16, 256, 256, 1, 5)]:
    x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16)
    out = torch.empty(Cout, Cin, k, k, device=dev)
    f = lambda: ops.conv2d_wgrad(x, dy, k, out=out)
    dw = f()
    if N * H * H <= 65536:
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Cout, Cin, k, k), dy.float().permute(0, 3, 1, 2), padding=k // 2)
        rel = ((dw - ref).norm() / ref.norm()).item()
    else:
        rel = float("nan")
    us = timeit(f)
    fl = 2.0 * N * H * H * Cout * Cin * k * k
    tot += us * cnt
    print(f"N{N} {H}x{H} {Cin}->{Cout} k{k}: rel {rel:.2e}  {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s")
print(f"weighted total {tot/1e3:.2f} ms")
