"""GPU box: every conv shape of one U-Net forward (EDM nets, or `cifar10 256` for the DDPM net) with its kernel id, launch count and graph-captured
time: where the 1x1 / small-kernel time goes.   python tools/edm_conv_shapes.py imagenet64_T10 100 [ksize]"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import configs_builtin
from models.cm.script_util import create_model_and_diffusion
from models.DxMI.openai_diffusion import OpenAIDiffusion
from dxmi_hip import ops

name, B = sys.argv[1], int(sys.argv[2])
only_k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.manual_seed(0)
if name == "cifar10":         # the DDPM U-Net of BASELINE configs[1] behind the VARSampler
    sys.path.insert(0, ROOT)
    import bench
    s = bench.build_sampler("cuda:0", 10)
    net = s.net
else:
    cfg = configs_builtin.get(name)
    net, diffusion = create_model_and_diffusion(**cfg.diffusion)
    for n, p in net.named_parameters():
        if p.abs().max() == 0:
            torch.nn.init.normal_(p, std=0.02)
    s = OpenAIDiffusion(net, diffusion, **cfg.sampler)
    net.to("cuda:0").eval()
shapes = collections.Counter()
orig = ops.conv2d


def spy(x, pw, **kw):
    if not pw.k27:
        N, IH, IW, C0 = x.shape
        C1 = kw["in1"].shape[3] if kw.get("in1") is not None else 0
        shapes[(N, IH, C0, C1, pw.Cout, pw.ksize, kw.get("stride", 1), bool(kw.get("upsample", False)), kw.get("residual") is not None,
                kw.get("addvec") is not None)] += 1
    return orig(x, pw, **kw)


ops.conv2d = spy
with torch.no_grad():
    s.sample(B, device="cuda:0")          # packs, warms
    shapes.clear()
    T = s.n_timesteps
    s.sample(B, device="cuda:0")
torch.cuda.synchronize()
ops.conv2d = orig


def graph_time(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


rows = []
for (N, H, C0, C1, Cout, k, stride, ups, res, vec), cnt in shapes.items():
    if only_k and k != only_k:
        continue
    x0 = torch.randn(N, H, H, C0, device="cuda:0").to(torch.bfloat16)
    x1 = torch.randn(N, H, H, C1, device="cuda:0").to(torch.bfloat16) if C1 else None
    pw = ops.pack_conv_weight(torch.randn(Cout, C0 + C1, k, k, device="cuda:0") * 0.03)
    OH = (H * (2 if ups else 1)) // stride
    r = torch.randn(N, OH, OH, Cout, device="cuda:0").to(torch.bfloat16) if res else None
    v = torch.randn(N, Cout, device="cuda:0") if vec else None
    bias = torch.randn(Cout, device="cuda:0")
    out = torch.empty(N, OH, OH, Cout, device="cuda:0", dtype=torch.bfloat16)
    kw = dict(in1=x1, bias=bias, residual=r, addvec=v, stride=stride, upsample=ups, out=out)
    if stride == 2:
        kw.update(pad=0, pad_br=1)
    prof = ops.OpProfiler(); ops.PROFILER = prof
    ops.conv2d(x0, pw, **kw)
    ops.PROFILER = None
    torch.cuda.synchronize()
    kid = prof.records[-1][1]
    us = graph_time(lambda: ops.conv2d(x0, pw, **kw))
    fl = 2.0 * N * OH * OH * Cout * (C0 + C1) * k * k
    by = 2.0 * (N * H * H * (C0 + C1) + N * OH * OH * Cout * (2 if res else 1))
    rows.append((cnt / T * us, cnt // T, kid, N, H, C0, C1, Cout, k, stride, ups, res, us, fl / us / 1e6, by / us / 1e3))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{name} B={B}: conv time per forward {tot/1e3:.2f} ms")
for r in rows[:40]:
    print(f"  {r[0]:8.1f} us/fwd  x{r[1]:2d}  kid {r[2]:6d}  {r[4]:3d}x{r[4]:<3d} {r[5]:4d}+{r[6]:<4d}->{r[7]:4d} k{r[8]} s{r[9]} ups={int(r[10])} res={int(r[11])}  {r[12]:7.1f} us  {r[13]:6.0f} TFLOP/s  {r[14]:6.0f} GB/s")
