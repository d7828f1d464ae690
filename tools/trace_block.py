"""Debug aid (GPU box): one ResnetBlock, op by op, HIP vs the oracle's bf16 storage model.
Reports rel-L2 and the fraction of bf16 outputs that are bit-identical."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops  # noqa: E402

DEV = "cuda:0"
bf = lambda x: x.to(torch.bfloat16).float()
nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
nchw = lambda y: y.float().cpu().permute(0, 3, 1, 2).contiguous()


def report(name, got, ref):
    rel = ((got - ref).norm() / ref.norm()).item()
    same = (got == ref).float().mean().item()
    print(f"{name:28s} rel-L2 {rel:.3e}  identical {100 * same:.2f}%  max|d| {(got - ref).abs().max().item():.3e}")


g = torch.Generator().manual_seed(0)
N, C, H = 2, 128, 32
x = bf(torch.randn(N, C, H, H, generator=g))
gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
w = torch.randn(C, C, 3, 3, generator=g) * (3.0 / (9 * C)) ** 0.5
b = 0.1 * torch.randn(C, generator=g)

# 1. GroupNorm + SiLU
ref_gn = bf(F.silu(F.group_norm(x, 32, gamma, beta, 1e-6)))
got_gn = nchw(ops.groupnorm_silu(nhwc(x), gamma.to(DEV), beta.to(DEV)))
report("groupnorm+silu", got_gn, ref_gn)
ref_gn2 = bf(F.group_norm(x, 32, gamma, beta, 1e-6))
got_gn2 = nchw(ops.groupnorm_silu(nhwc(x), gamma.to(DEV), beta.to(DEV), silu=False))
report("groupnorm (no silu)", got_gn2, ref_gn2)

# 2. conv 3x3 on the SAME input (the reference GN output), bf16-rounded weights
ref_c = bf(F.conv2d(ref_gn, bf(w), b, padding=1))
pw = ops.pack_conv_weight(w.to(DEV))
got_c = nchw(ops.conv2d(nhwc(ref_gn), pw, bias=b.to(DEV)))
report("conv3x3 (bf16 out)", got_c, ref_c)
ref_c64 = bf(F.conv2d(ref_gn.double(), bf(w).double(), b.double(), padding=1).float())
report("conv3x3 vs fp64-accum ref", got_c, ref_c64)
report("torch fp32 conv vs fp64", ref_c, ref_c64)
