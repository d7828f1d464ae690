"""GPU box: calibrate achievable HBM bandwidth (torch copy / fill / read-reduce) and our GN kernel."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_(); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a)); print(f"copy {mb} MB: {2*mb/1024/t:.2f} TB/s ({t*1e6:.1f} us)")
    t = timeit(lambda: b.fill_(1.0)); print(f"fill {mb} MB: {mb/1024/t:.2f} TB/s ({t*1e6:.1f} us)")
    t = timeit(lambda: a.float().sum() if mb <= 256 else None)
    t = timeit(lambda: torch.add(a, 1.0, out=b)); print(f"add  {mb} MB: {2*mb/1024/t:.2f} TB/s ({t*1e6:.1f} us)")
for (N, H, C) in [(256, 32, 128), (256, 16, 256), (256, 32, 256)]:
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16); g = torch.ones(C, device=dev); bb = torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    t = timeit(lambda: ops.groupnorm_silu(x, g, bb, out=out))
    mbs = x.numel() * 2 / 2**20
    print(f"gn_silu N{N} H{H} C{C}: {2*mbs/1024/t:.2f} TB/s ({t*1e6:.1f} us, {mbs:.0f} MB in + out)")
