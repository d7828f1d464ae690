"""GPU box: small-map 3x3 convs (4x4, 8x8) — correctness vs torch fp32 and time; run with DXMI_CONV_SM=0 / 3 to A/B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch, torch.nn.functional as F
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=50):
    """device time per launch: n launches captured in one hipGraph (the eager loop is host-bound below ~12 us per launch)"""
    for _ in range(5): fn()
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
        e0.record()
        g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("DXMI_CONV_SM =", os.environ.get("DXMI_CONV_SM"), "NL =", os.environ.get("DXMI_CONV_SM_NL"))
for (N, H, C0, C1, Cout, res) in [(256, 4, 256, 0, 256, False), (256, 4, 256, 0, 256, True), (256, 4, 256, 256, 256, False), (256, 8, 256, 0, 256, False),
                                  (256, 8, 256, 0, 256, True), (256, 8, 256, 256, 256, False), (37, 4, 256, 0, 256, True), (5, 8, 256, 256, 256, True)]:
    x0 = torch.randn(N, H, H, C0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device=dev).to(torch.bfloat16) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device=dev) * 0.03
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(Cout, device=dev)
    temb = torch.randn(N, Cout, device=dev)
    r = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.conv2d(x0, pw, in1=x1, bias=bias, addvec=temb, residual=r, out=out)
    y = f().float()
    xin = torch.cat([x0, x1], 3) if C1 else x0
    ref = F.conv2d(xin.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), bias, padding=1) + temb[:, :, None, None]
    if res: ref = ref + r.float().permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)
    rel = ((y - ref).norm() / ref.norm()).item()
    d = ops._lib.ConvDesc() if False else None
    us = timeit(f)
    fl = 2.0 * N * H * H * Cout * (C0 + C1) * 9
    print(f"N{N} {H}x{H} {C0}+{C1}->{Cout} res={res}: rel {rel:.2e}  {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s")
