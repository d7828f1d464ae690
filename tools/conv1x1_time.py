"""GPU box: time (and check against an fp32 matmul) the 1x1 conv shapes of the CIFAR-10 U-Net at B=256."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops
dev = "cuda:0"
B = int(os.environ.get("B", 256))
SH = os.environ.get("SHAPES")
CASES = [tuple(int(v) for v in t.split(",")) for t in SH.split(";")] if SH else None
for (c0, c1, cout, h, res) in CASES or ((256, 0, 768, 16, False), (256, 0, 256, 16, True), (256, 256, 256, 16, True), (128, 0, 256, 16, True),
                               (256, 128, 256, 16, True), (128, 128, 128, 32, True), (256, 128, 128, 32, True), (192, 0, 384, 16, False)):
    cin = c0 + c1
    x0 = torch.randn(B, h, h, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(B, h, h, c1, device=dev).to(torch.bfloat16) if c1 else None
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(cout, device=dev)
    r = torch.randn(B, h, h, cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(B, h, h, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.conv2d(x0, pw, bias=bias, residual=r, out=out, in1=x1)
    torch.cuda.synchronize()
    x = x0 if x1 is None else torch.cat([x0, x1], -1)
    ref = x.float().reshape(-1, cin) @ w.to(torch.bfloat16).float().reshape(cout, cin).t() + bias
    if res:
        ref = ref + r.float().reshape(-1, cout)
    err = (out.float().reshape(-1, cout) - ref).abs().max().item() / ref.abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.conv2d(x0, pw, bias=bias, residual=r, out=out, in1=x1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    mb = B * h * h * (cin + cout * (2 if res else 1)) * 2 / 1e6
    print(f"{c0:4d}+{c1:3d}->{cout:4d} @{h:2d} res={int(res)}: {us:7.1f} us  {mb / us * 1e3 / 1e3:6.2f} TB/s  max rel err {err:.2e}")
