"""GPU box: A/B of the streaming GroupNorm path in the whole U-Net forward (B=256) + conv_ws with/without the statistics epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
from models.DxMI.unet_small import Model
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32).to(dev).eval()
x = torch.randn(256, 3, 32, 32, device=dev)
t = torch.full((256,), 500.0, device=dev)
with torch.no_grad():
    outs = {}
    for rep in range(3):
        for mode, hw in (("stream", 256), ("resident", 1 << 30)):
            net.STREAM_GN_MIN_HW = hw
            for _ in range(3): y = net(x, t)
            us = timeit(lambda: net(x, t), 20)
            outs[mode] = y
            print(f"forward B=256 GN {mode}: {us/1e3:.3f} ms")
    d = (outs["stream"] - outs["resident"]).norm() / outs["resident"].norm()
    print("rel diff stream vs resident:", d.item())

for (N, H, Cin, Cout, res) in [(256, 32, 128, 128, True), (256, 32, 128, 128, False), (256, 32, 256, 128, False), (256, 16, 256, 256, True), (256, 16, 512, 256, False)]:
    xx = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.03)
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    f0 = lambda: ops.conv2d(xx, pw, bias=bias, residual=r, out=out)
    f1 = lambda: ops.conv2d(xx, pw, bias=bias, residual=r, out=out, want_stats=True)
    for _ in range(20): f0(); f1()
    a, b = [], []
    for rep in range(5):
        a.append(timeit(f0, 100)); b.append(timeit(f1, 100))
    print(f"conv N{N} {H}x{H} {Cin}->{Cout} res={res}: plain {min(a):.1f} us (med {sorted(a)[2]:.1f})  +stats {min(b):.1f} us (med {sorted(b)[2]:.1f})")
