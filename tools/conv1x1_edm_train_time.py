"""GPU box: the 1x1 convs of the ImageNet-64 ADM net at its TRAIN batch (16 images), graph-captured: us per launch, the kernel the
dispatcher picks, TFLOP/s and TB/s (forward shapes and the data-gradient shapes = transposed weights).
    N=16 python tools/conv1x1_edm_train_time.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
N = int(os.environ.get("N", 16))
SIDE = torch.cuda.Stream()


def graph_time(fn, n=20):
    with torch.cuda.stream(SIDE):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=SIDE):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


ops.tune_for_throughput(True)
tot = 0.0
for (c0, c1, cout, h, res) in [(384, 0, 1152, 32, 0), (384, 0, 384, 32, 1), (1152, 0, 384, 32, 0), (576, 0, 1728, 16, 0), (576, 0, 576, 16, 1), (1728, 0, 576, 16, 0),
                               (768, 0, 2304, 8, 0), (768, 0, 768, 8, 1), (2304, 0, 768, 8, 0), (384, 192, 384, 32, 0), (384, 384, 384, 32, 0), (576, 384, 576, 16, 0),
                               (576, 576, 576, 16, 0), (768, 576, 768, 8, 0), (768, 768, 768, 8, 0), (192, 0, 384, 32, 0), (384, 0, 576, 16, 0), (576, 0, 768, 8, 0)]:
    cin = c0 + c1
    x0 = torch.randn(N, h, h, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, h, h, c1, device=dev).to(torch.bfloat16) if c1 else None
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(cout, device=dev)
    r = torch.randn(N, h, h, cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(N, h, h, cout, device=dev, dtype=torch.bfloat16)
    us = graph_time(lambda: ops.conv2d(x0, pw, bias=bias, residual=r, out=out, in1=x1))
    gf = 2.0 * N * h * h * cin * cout / 1e9
    mb = (N * h * h * (cin + cout * (2 if res else 1)) * 2 + cin * cout * 2) / 1e6
    tot += us
    print(f"{c0:4d}+{c1:3d}->{cout:4d} @{h:2d} res={res}: {us:7.1f} us  {gf / us * 1e3:6.0f} TFLOP/s  {mb / us:5.2f} TB/s  ({gf:5.1f} GFLOP, {mb:5.1f} MB)", flush=True)
print(f"sum {tot:.1f} us")
