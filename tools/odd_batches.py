import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch, bench
from models.DxMI.trainer import append_buffer, reset_buffer
dev = torch.device("cuda:0")
s = bench.build_sampler(dev, 10)
g = torch.Generator().manual_seed(1)
noise8 = [torch.randn(8, 3, 32, 32, generator=g) for _ in range(11)]
with torch.no_grad():
    ref = s.sample(8, device=dev, noise=noise8)["sample"].cpu()
    for B in (1, 3, 5, 7):
        out = s.sample(B, device=dev, noise=[n[:B] for n in noise8])["sample"].cpu()
        print("B", B, "max abs diff vs first rows of B=8:", (out - ref[:B]).abs().max().item())
for B in (3, 5):
    tr = bench.build_trainer(s, dev, B, 10)
    imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
    logs = bench.train_step(tr, s, imgs, dev)
    print("train B", B, "ok", all(v == v for v in logs[0].values()))
