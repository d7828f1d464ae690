"""GPU box: A/B of the GroupNorm fused into conv_sm's epilogue (4x4 maps) in the graph-captured U-Net forward (B=256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
from models.DxMI.unet_small import Model
dev = "cuda:0"
torch.manual_seed(0)

net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32).to(dev).eval()
x = torch.randn(256, 3, 32, 32, device=dev)
t = torch.full((256,), 500.0, device=dev)
graphs, outs = {}, {}
with torch.no_grad():
    for mode in (False, True):
        net.FUSE_GN_SMALL = mode
        for _ in range(3): net(x, t)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs[mode] = net(x, t)
        graphs[mode] = g
    res = {False: [], True: []}
    for rep in range(7):
        for mode in (False, True):
            g = graphs[mode]
            for _ in range(5): g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): g.replay()
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 30)
    for mode in (False, True):
        r = sorted(res[mode])
        print(f"forward B=256 fused_gn={mode}: min {r[0]:.4f} ms  med {r[3]:.4f} ms")
    d = (outs[True] - outs[False]).norm() / outs[False].norm()
    print("rel diff fused vs separate:", d.item())
