"""Debug aid (GPU box): first ResnetBlock of the real net, op by op."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops  # noqa: E402
from models.DxMI.unet_small import Model  # noqa: E402
from oracle import Precision, unet_small as ounet  # noqa: E402
from oracle.weights import formula_tensor  # noqa: E402

DEV = "cuda:0"
nchw = lambda y: y.float().cpu().permute(0, 3, 1, 2).contiguous()
nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def report(name, got, ref):
    rel = ((got - ref).norm() / ref.norm()).item()
    same = (got == ref).float().mean().item()
    print(f"{name:28s} rel-L2 {rel:.3e}  identical {100 * same:.2f}%  max|d| {(got - ref).abs().max().item():.3e}")


net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32)
sd = {k: formula_tensor(k, v.shape) for k, v in net.state_dict().items()}
net.load_state_dict(sd)
net = net.to(DEV).eval()
g = np.load(os.path.join(ROOT, "tests/golden/unet_small_forward.npz"))
x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
prec = Precision("bf16")
pk = net.packed()
with torch.no_grad():
    # oracle pieces
    temb = ounet.timestep_embedding_sincos(t, 128)
    temb = F.linear(prec.act(temb), prec.w(sd["temb.dense.0.weight"]), sd["temb.dense.0.bias"])
    temb = F.linear(prec.act(ounet.swish(temb)), prec.w(sd["temb.dense.1.weight"]), sd["temb.dense.1.bias"])
    s_temb = ounet.swish(temb)
    h0 = prec.act(ounet._conv(sd, "conv_in", prec.act(x), prec, padding=1))
    pre = "down.0.block.0"
    a1 = prec.act(ounet.swish(ounet._gn(sd, pre + ".norm1", h0)))
    c1 = ounet._conv(sd, pre + ".conv1", a1, prec, padding=1)
    tproj = F.linear(prec.act(s_temb), prec.w(sd[pre + ".temb_proj.weight"]), sd[pre + ".temb_proj.bias"])
    h1 = prec.act(c1 + tproj[:, :, None, None])
    a2 = prec.act(ounet.swish(ounet._gn(sd, pre + ".norm2", h1)))
    c2 = ounet._conv(sd, pre + ".conv2", a2, prec, padding=1)
    out = prec.act(h0 + c2)
    # HIP pieces
    b = net.down[0].block[0]
    emb = ops.timestep_embedding(t.to(DEV), 128, order=0)
    e1 = ops.linear(emb, pk["dense0"], net.temb.dense[0].bias, post_act=ops.ACT_SILU)
    st = ops.linear(e1, pk["dense1"], net.temb.dense[1].bias, post_act=ops.ACT_SILU)
    tp = ops.linear(st, pk["tproj"], pk["tproj_bias"])
    H0 = ops.conv2d(x.to(DEV), pk["conv_in"], bias=net.conv_in.bias)
    report("conv_in", nchw(H0), h0)
    off = pk[id(b), "toff"]
    report("tproj", tp[:, off:off + 128].cpu(), tproj)
    A1 = ops.groupnorm_silu(H0, b.norm1.weight, b.norm1.bias)
    report("norm1+silu", nchw(A1), a1)
    H1 = ops.conv2d(A1, pk[id(b), "conv1"], bias=b.conv1.bias, addvec=tp[:, off:off + 128])
    report("conv1+temb", nchw(H1), h1)
    H1n = ops.conv2d(A1, pk[id(b), "conv1"], bias=b.conv1.bias)
    report("conv1 (no temb)", nchw(H1n), prec.act(c1))
    A2 = ops.groupnorm_silu(H1, b.norm2.weight, b.norm2.bias)
    report("norm2+silu", nchw(A2), a2)
    A2s = ops.groupnorm_silu(nhwc(h1), b.norm2.weight, b.norm2.bias)
    report("norm2+silu (same input)", nchw(A2s), a2)
    OUT = ops.conv2d(A2, pk[id(b), "conv2"], bias=b.conv2.bias, residual=H0)
    report("conv2+res", nchw(OUT), out)
    OUTs = ops.conv2d(nhwc(a2), pk[id(b), "conv2"], bias=b.conv2.bias, residual=H0)
    report("conv2+res (same input)", nchw(OUTs), out)
