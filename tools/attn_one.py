"""GPU box: one attention shape a few times (for rocprofv3 --pmc passes).  python tools/attn_one.py N T heads D"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
N, T, heads, D = (int(a) for a in sys.argv[1:5])
qkv = torch.randn(N, T, 3 * heads * D, device="cuda:0").to(torch.bfloat16)
for _ in range(5):
    ops.attention(qkv, heads, D ** -0.5)
torch.cuda.synchronize()
