"""GPU box: A/B of proj_out fused behind the attention (16x16 AttnBlocks) in the graph-captured U-Net forward + the kernels alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
from models.DxMI.unet_small import Model
dev = "cuda:0"
torch.manual_seed(0)


def graph_time(fn, n=20, reps=5):
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


N, T, C = 256, 256, 256
qkv = torch.randn(N, T, 3 * C, device=dev).to(torch.bfloat16)
x = torch.randn(N, T, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C, 1, 1, device=dev) * 0.06
b = torch.randn(C, device=dev)
wp, pw = ops.pack_attn_proj_weight(w), ops.pack_conv_weight(w)
out, a = torch.empty_like(x), torch.empty_like(x)
print(f"attention alone        {graph_time(lambda: ops.attention(qkv, 1, 0.0625, out=a)):.1f} us")
print(f"proj conv alone        {graph_time(lambda: ops.conv2d(a.view(N, 16, 16, C), pw, bias=b, residual=x.view(N, 16, 16, C), out=out.view(N, 16, 16, C))):.1f} us")
print(f"attention + proj conv  {graph_time(lambda: (ops.attention(qkv, 1, 0.0625, out=a), ops.conv2d(a.view(N, 16, 16, C), pw, bias=b, residual=x.view(N, 16, 16, C), out=out.view(N, 16, 16, C)))):.1f} us")
print(f"fused                  {graph_time(lambda: ops.attention_proj(qkv, wp, b, x, 1, 0.0625, out=out)):.1f} us")

net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32).to(dev).eval()
xi = torch.randn(256, 3, 32, 32, device=dev)
t = torch.full((256,), 500.0, device=dev)
with torch.no_grad():
    res, outs = {}, {}
    for mode in (False, True, False, True):
        net.FUSE_ATTN_PROJ = mode
        outs[mode] = net(xi, t)
        res.setdefault(mode, []).append(graph_time(lambda: net(xi, t), n=5, reps=7) / 1e3)
    for mode in (False, True):
        print(f"forward B=256 fused_attn_proj={mode}: {min(res[mode]):.4f} ms")
    print("rel diff:", ((outs[True] - outs[False]).norm() / outs[False].norm()).item())
