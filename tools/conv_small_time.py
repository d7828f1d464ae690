"""GPU box: time the small-map 3x3 conv shapes (B=256) with HIP events."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops
dev = "cuda:0"
B = int(os.environ.get("B", 256))
for (cin, cout, h, res) in ((256, 256, 8, True), (256, 256, 8, False), (512, 256, 8, False), (256, 256, 4, True), (512, 256, 4, False)):
    x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
    pw = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    bias = torch.randn(cout, device=dev)
    r = torch.randn(B, h, h, cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(B, h, h, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.conv2d(x, pw, bias=bias, residual=r, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.conv2d(x, pw, bias=bias, residual=r, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{cin:4d}->{cout:4d} @{h:2d} res={int(res)}: {us:7.1f} us  {2.0 * B * h * h * cout * cin * 9 / us / 1e6:7.0f} TFLOP/s")
