"""GPU box: conv_stem_kernel (3 -> 128, 32x32, B=256) graph-captured; DXMI_LIB selects a (possibly ablated) library build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
x = torch.randn(256, 3, 32, 32, device=dev)
pw = ops.pack_conv_weight(torch.randn(128, 3, 3, 3, device=dev) * 0.2, k27=True)
b = torch.randn(128, device=dev)
out = torch.empty(256, 32, 32, 128, device=dev, dtype=torch.bfloat16)
for kw in (dict(), dict(act=ops.ACT_LEAKY02)):
    fn = lambda: ops.conv2d(x, pw, bias=b, out=out, **kw)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(os.path.basename(os.environ.get("DXMI_LIB", "default")), kw, f"{best:.1f} us  {out.numel()*2/best/1e3:.0f} GB/s out")
