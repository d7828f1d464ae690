"""GPU box: resident GroupNorm kernel with and without the SiLU (how VALU-bound is it?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
for (N, H, C0, C1) in [(256, 32, 128, 0), (256, 32, 256, 128), (256, 16, 256, 0), (256, 16, 256, 256), (256, 8, 256, 0)]:
    C = C0 + C1
    x0 = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = []
    for silu in (True, False):
        for _ in range(3): ops.groupnorm_silu(x0, g, b, in1=x1, eps=1e-6, silu=silu)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.groupnorm_silu(x0, g, b, in1=x1, eps=1e-6, silu=silu)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    mb = N * H * H * C * 2 * 2 / 1e6
    print(f"N={N} {H}x{H} C={C0}+{C1}: silu {res[0]:.1f} us ({mb/res[0]/1e3:.2f} TB/s)   no silu {res[1]:.1f} us ({mb/res[1]/1e3:.2f} TB/s)")
