"""GPU box: the 1x1 conv shapes of the ImageNet-64 ADM net with K = 576 / 768 (q|k|v, proj_out, skip_connection; B = 100),
graph-captured time per launch on the kernel the library selects: DXMI_CONV1X1_RW8=0 / 1 switches conv1x1_rw8_kernel (round 5) off /
on; SHAPES=small_k lists the K = 384 / 512 shapes (ImageNet-64 level 1, CIFAR-10, LSUN) — DESIGN 5.5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)


def graph_time(fn, n=10, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


B = int(os.environ.get("B", 100))
print("library:", os.environ.get("DXMI_LIB", "product"), " B =", B)
tot = 0.0
SHAPES = [(16, 576, 0, 1728, False, 7, "qkv 576"), (16, 576, 0, 576, True, 7, "proj 576"),
          (8, 768, 0, 2304, False, 8, "qkv 768"), (8, 768, 0, 768, True, 8, "proj 768"),
          (16, 576, 0, 768, False, 1, "skip 576->768 (8x8 after down)"), (32, 384, 384, 384, False, 2, "skip 768->384 @32"),
          (32, 384, 192, 384, False, 1, "skip 576->384 @32"), (64, 384, 192, 192, False, 1, "skip 576->192 @64")]
if os.environ.get("SHAPES") == "small_k":      # K = 384 / 512 shapes: ImageNet-64 level 1 (B = 100), CIFAR-10 (B = 256 via B=256)
    SHAPES = [(32, 384, 0, 1152, False, 7, "qkv 384"), (32, 384, 0, 384, True, 7, "proj 384"), (64, 192, 192, 192, False, 3, "skip 384->192 @64"),
              (16, 256, 256, 256, False, 2, "cifar 512->256 @16"), (16, 256, 128, 256, False, 1, "cifar 384->256 @16"),
              (32, 256, 128, 128, False, 1, "cifar 384->128 @32"), (32, 512, 0, 1536, False, 1, "lsun-like qkv 512 @32")]
for (H, C0, C1, Cout, res, count, what) in SHAPES:
    x0 = torch.randn(B, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(B, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    w = torch.randn(Cout, C0 + C1, 1, 1, device=dev) * 0.05
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.conv2d(x0, pw, in1=x1, bias=bias, residual=r, out=out)
    us = graph_time(f)
    fl = 2.0 * B * H * H * Cout * (C0 + C1)
    by = 2.0 * B * H * H * (C0 + C1 + Cout * (2 if res else 1))
    tot += us * count
    print(f"{what:34s} {H}x{H} {C0}+{C1}->{Cout}: {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s  {by / us / 1e6:5.2f} TB/s   x{count} per forward")
print(f"sum over a forward's launches of these shapes: {tot / 1e3:.2f} ms")
