"""GPU box: is the streaming GroupNorm apply pass at the ceiling of a launch of its size?  For the shapes of the CIFAR-10 generation
step (256 images) and of C4 (100 images): gn_apply_kernel against a plain bf16 copy (torch copy_, the same bytes in and out) and a
bf16 `x * a + b` elementwise pass, each over a ROTATION of 8 distinct input / output buffers (1-2 GB: neither L2 nor the 256 MB MALL
holds a launch's data from the previous one), graph-captured so that no host time is in the numbers.
    python tools/gn_apply_ceiling.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
R = 8
SHAPES = [(256, 32, 128), (256, 32, 256), (256, 16, 256), (256, 32, 384), (100, 64, 192), (100, 32, 384), (100, 16, 576)]


def graph_time(fns, reps=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                for f in fns:
                    f()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


for (B, H, C) in SHAPES:
    xs = [torch.randn(B, H, H, C, device=dev).to(torch.bfloat16) for _ in range(R)]
    outs = [torch.empty_like(x) for x in xs]
    sts = [ops.block_stats(x) for x in xs]
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mb = 4.0 * B * H * H * C / 1e6
    ops.GN_APPLY_SPLIT = False
    t_gn = graph_time([lambda i=i: ops.groupnorm_apply(xs[i], sts[i], g, b, eps=1e-6, silu=True, out=outs[i]) for i in range(R)])
    ops.GN_APPLY_SPLIT = True        # finalize launch (one workgroup per image) + prologue-free apply: both launches are in the time
    t_sp = graph_time([lambda i=i: ops.groupnorm_apply(xs[i], sts[i], g, b, eps=1e-6, silu=True, out=outs[i]) for i in range(R)])
    ops.GN_APPLY_SPLIT = False
    t_cp = graph_time([lambda i=i: outs[i].copy_(xs[i]) for i in range(R)])
    t_ax = graph_time([lambda i=i: torch.add(xs[i], 1.0, out=outs[i]) for i in range(R)])
    print(f"B={B} {H}x{H}x{C} ({mb:6.1f} MB in+out, P={sts[0].P}): gn_apply {t_gn:6.1f} us {mb / t_gn:5.2f} TB/s | copy_ {t_cp:6.1f} us {mb / t_cp:5.2f} TB/s"
          f" | x+1 {t_ax:6.1f} us {mb / t_ax:5.2f} TB/s | gn_apply / copy = {t_gn / t_cp:.2f} | split (finalize + apply) {t_sp:6.1f} us")
