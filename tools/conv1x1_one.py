"""GPU box: a few launches of one 1x1 conv shape (for rocprofv3 --pmc): python3 tools/conv1x1_one.py C0 C1 Cout H res"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops
c0, c1, cout, h, res = [int(v) for v in sys.argv[1:6]]
dev, B = "cuda:0", 256
x0 = torch.randn(B, h, h, c0, device=dev).to(torch.bfloat16)
x1 = torch.randn(B, h, h, c1, device=dev).to(torch.bfloat16) if c1 else None
pw = ops.pack_conv_weight(torch.randn(cout, c0 + c1, 1, 1, device=dev) * 0.05)
bias = torch.randn(cout, device=dev)
r = torch.randn(B, h, h, cout, device=dev).to(torch.bfloat16) if res else None
out = torch.empty(B, h, h, cout, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.conv2d(x0, pw, bias=bias, residual=r, out=out, in1=x1)
torch.cuda.synchronize()
