"""GPU box: first layer of the EDM net whose output for images [40:47] differs between a batch of 100 and a batch of 7."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import configs_builtin
from models.cm.script_util import create_model_and_diffusion
from dxmi_hip import ops
DEV = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "imagenet64_T10"
cfg = configs_builtin.CONFIGS[name]
torch.manual_seed(0)
with torch.device(DEV):
    net, _ = create_model_and_diffusion(**dict(cfg["diffusion"]))
with torch.no_grad():
    for p in net.parameters():
        if float(p.abs().max()) == 0: torch.nn.init.normal_(p, std=0.02)
net.to(DEV).eval()
res = cfg["diffusion"]["image_size"]
B, lo, hi = (100, 40, 47) if res == 64 else (16, 5, 7)
g = torch.Generator(device=DEV).manual_seed(1)
x = torch.randn(B, 3, res, res, device=DEV, generator=g)
t = torch.full((B,), 3.5, device=DEV)
y = torch.randint(0, 1000, (B,), device=DEV, generator=g) if cfg["diffusion"]["class_cond"] else None
ta, tb = [], []
with torch.no_grad():
    oa = net.forward_inference(x, t, y, trace=ta)
    ob = net.forward_inference(x[lo:hi].contiguous(), t[lo:hi].contiguous(), None if y is None else y[lo:hi].contiguous(), trace=tb)
first = None
for (na, va), (nb, vb) in zip(ta, tb):
    a = va[lo:hi] if va.shape[0] == B else va
    same = torch.equal(a, vb)
    if not same and first is None:
        first = na
        d = (a.float() - vb.float()).abs().max().item()
        print(f"first difference at {na}: shape {tuple(vb.shape)} max abs diff {d:.3e}, fraction differing {(a != vb).float().mean().item():.4f}")
print("output equal:", torch.equal(oa[lo:hi], ob), "| first differing layer:", first)
