"""GPU box: conv_ws_kernel shapes of the CIFAR U-Net (graph-captured device time) + whole forward; run with DXMI_CONV_WS_SPLIT=0/1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch, torch.nn.functional as F
from dxmi_hip import ops
from models.DxMI.unet_small import Model
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("DXMI_CONV_WS_SPLIT =", os.environ.get("DXMI_CONV_WS_SPLIT"))
for (N, H, C0, C1, Cout, res, ups) in [(256, 32, 128, 0, 128, True, False), (256, 32, 128, 0, 128, False, False), (256, 32, 256, 128, 128, False, False),
                                       (256, 32, 256, 0, 256, False, True), (256, 16, 256, 0, 256, True, False), (256, 16, 256, 256, 256, False, False),
                                       (256, 16, 128, 0, 256, False, False)]:
    IH = H // 2 if ups else H
    x0 = torch.randn(N, IH, IH, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, IH, IH, C1, device=dev).to(torch.bfloat16) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device=dev) * 0.03
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(Cout, device=dev)
    temb = torch.randn(N, Cout, device=dev)
    r = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.conv2d(x0, pw, in1=x1, bias=bias, addvec=temb, residual=r, upsample=ups, out=out, want_stats=True)
    y = f()[0][:4].float()
    xin = torch.cat([x0, x1], 3) if C1 else x0
    xi = xin[:4].float().permute(0, 3, 1, 2)
    if ups: xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    ref = F.conv2d(xi, w.to(torch.bfloat16).float(), bias, padding=1) + temb[:4, :, None, None]
    if res: ref = ref + r[:4].float().permute(0, 3, 1, 2)
    rel = ((y - ref.permute(0, 2, 3, 1)).norm() / ref.norm()).item()
    us = timeit(f)
    fl = 2.0 * N * H * H * Cout * (C0 + C1) * 9
    print(f"N{N} {H}x{H} {C0}+{C1}->{Cout} res={res} ups={ups}: rel {rel:.2e}  {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s")
net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32).to(dev).eval()
x = torch.randn(256, 3, 32, 32, device=dev)
t = torch.full((256,), 500.0, device=dev)
with torch.no_grad():
    for _ in range(3): net(x, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): net(x, t)
    e1.record(); torch.cuda.synchronize()
    print(f"forward B=256: {e0.elapsed_time(e1)/20:.3f} ms")
