"""Debug aid (GPU box): compare the HIP U-Net block by block with the oracle's bf16 storage model."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from models.DxMI.unet_small import Model  # noqa: E402
from oracle import Precision, unet_small as ounet  # noqa: E402
from oracle.weights import formula_tensor  # noqa: E402

net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32)
sd = {k: formula_tensor(k, v.shape) for k, v in net.state_dict().items()}
net.load_state_dict(sd)
net = net.to("cuda:0").eval()
g = np.load(os.path.join(ROOT, "tests/golden/unet_small_forward.npz"))
x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
th, to, t32 = [], [], []
yh = net.forward_inference(x.cuda(), t.cuda(), trace=th).cpu()
with torch.no_grad():
    yo = ounet.forward(sd, ounet.UNetSmallConfig(), x, t, Precision("bf16"), trace=to)
    y32 = ounet.forward(sd, ounet.UNetSmallConfig(), x, t, Precision("fp32"), trace=t32)
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
for (n1, a), (n2, b), (n3, c) in zip(th, to, t32):
    assert n1 == n2 == n3
    a = a.float().cpu()
    if a.dim() == 4:
        a = a.permute(0, 3, 1, 2)
    print(f"{n1:24s} hip-vs-bf16model {rel(a, b):.3e}   hip-vs-fp32 {rel(a, c):.3e}   bf16model-vs-fp32 {rel(b, c):.3e}")
print("out", rel(yh, yo), rel(yh, y32), rel(yo, y32))
