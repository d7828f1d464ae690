"""GPU box: 3x3 conv shapes whose (pixel tile, cout tile) grid leaves most CUs idle (EDM nets at the train batch): time per launch
with the wave-specialised kernels and with DXMI_CONV_WS=0 (conv_pipe_kernel: 64-pixel tiles).  B=16 python tools/conv_small_grid.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops
dev = "cuda:0"
B = int(os.environ.get("B", 16))
SHAPES = tuple(tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")) if "SHAPES" in os.environ else ((192, 192, 64), (384, 192, 64), (384, 384, 32), (768, 384, 32), (576, 576, 16), (1152, 576, 16), (768, 768, 8), (1536, 768, 8))
for (cin, cout, h) in SHAPES:
    x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
    pw = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    bias = torch.randn(cout, device=dev)
    out = torch.empty(B, h, h, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.conv2d(x, pw, bias=bias, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.conv2d(x, pw, bias=bias, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tiles = (B * h * h // 256) * ((cout + 127) // 128)
    print(f"B={B} {cin:4d}->{cout:4d} @{h:2d}: {us:7.1f} us  {2.0 * B * h * h * cout * cin * 9 / us / 1e6:7.0f} TFLOP/s   ws tiles {tiles}")
