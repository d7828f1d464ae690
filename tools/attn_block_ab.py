"""GPU box: the 16x16 AttnBlock as one launch (ops.attn_block) against the three launches it replaces (graph-captured device time,
256 images) + the U-Net forward and a generation step with FUSE_ATTN_BLOCK on / off, alternating."""
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


N, T, C = int(os.environ.get("N", 256)), 256, 256
x = (torch.randn(N, 16, 16, C, device=dev) * 1.3 + 0.2).to(torch.bfloat16)
ws = {k: torch.randn(C, C, 1, 1, device=dev) * 0.06 for k in "qkvp"}
bs = {k: torch.randn(C, device=dev) * 0.3 for k in "qkvp"}
gamma, beta = 1 + 0.3 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)
st = ops.block_stats(x)
packed = ops.attn_block_pack(ws["q"], bs["q"], ws["k"], ws["v"], bs["v"], ws["p"], bs["p"], 0.0625)
qkv_w = ops.pack_conv_weight(torch.cat([ws["q"], ws["k"], ws["v"]], 0))
qkv_b = torch.cat([bs["q"], bs["k"], bs["v"]], 0).contiguous()
wp = ops.pack_attn_proj_weight(ws["p"])
hn, qkv, out = torch.empty_like(x), torch.empty(N, 16, 16, 3 * C, device=dev, dtype=torch.bfloat16), torch.empty_like(x)


def three():
    ops.groupnorm_silu(x, gamma, beta, eps=1e-6, silu=False, stats=(st, None), out=hn)
    ops.conv2d(hn, qkv_w, bias=qkv_b, out=qkv)
    return ops.attention_proj(qkv.view(N, T, 3 * C), wp, bs["p"], x.view(N, T, C), 1, 0.0625, out=out.view(N, T, C), want_stats=True)


one = lambda: ops.attn_block(x, st, gamma, beta, packed, out=out, want_stats=True)
y1 = one()[0].float().clone()
y3 = three()[0].float().view(N, 16, 16, C).clone()
print("one launch vs three launches, rel-L2 of the branch:", ((y1 - y3).norm() / (y3 - x.float()).norm()).item())
t3, t1 = graph_time(three), graph_time(one)
fl = 8.0 * N * T * C * C
print(f"three launches {t3:.1f} us   one launch {t1:.1f} us  ({fl / t1 / 1e6:.0f} TFLOP/s of its 4 GEMMs, {4.0 * N * T * C / t1 / 1e6:.2f} TB/s algorithmic)")

net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32).to(dev).eval()
xi = torch.randn(256, 3, 32, 32, device=dev)
t = torch.full((256,), 500.0, device=dev)
with torch.no_grad():
    res, outs = {}, {}
    for mode in (False, True, False, True, False, True):
        net.FUSE_ATTN_BLOCK = mode
        outs[mode] = net(xi, t)
        res.setdefault(mode, []).append(graph_time(lambda: net(xi, t), n=5, reps=7) / 1e3)
    for mode in (False, True):
        print(f"forward B=256 FUSE_ATTN_BLOCK={mode}: {min(res[mode]):.4f} ms  (runs: {', '.join(f'{v:.4f}' for v in res[mode])})")
    print("rel diff of the forward:", ((outs[True] - outs[False]).norm() / outs[False].norm()).item())
