"""GPU box: one-pass register-resident GroupNorm vs the generic two-kernel path on the U-Net's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
for (N, H, C0, C1) in [(256, 32, 128, 0), (256, 32, 256, 0), (256, 32, 256, 128), (256, 16, 256, 0), (256, 16, 256, 256), (256, 8, 256, 0), (256, 8, 512, 0), (256, 4, 256, 0)]:
    C = C0 + C1
    x0 = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    res = []
    for fn in (ops.groupnorm_silu, ops.groupnorm_generic):
        for _ in range(3): fn(x0, g, b, in1=x1, eps=1e-6, silu=True)
        ts = []
        for _ in range(10):
            big.zero_()   # evict the Infinity Cache between repetitions? (keeps both variants on equal footing)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(x0, g, b, in1=x1, eps=1e-6, silu=True); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        res.append(sorted(ts)[len(ts) // 2])
    mb = N * H * H * C * 2 * 2 / 1e6
    print(f"N={N} {H}x{H} C={C0}+{C1}: resident {res[0]:.1f} us ({mb/res[0]/1e3*1e3:.0f} GB/s)  generic {res[1]:.1f} us ({mb/res[1]*1e3/1e3:.0f} GB/s)")
