"""GPU box: only the value update (update_f_v, B=256, T=10) in a loop — for rocprofv3 --kernel-trace --stats."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
import bench
from models.DxMI.trainer import append_buffer, reset_buffer
dev = torch.device("cuda:0")
B, T = 256, 10
s = bench.build_sampler(dev, T)
tr = bench.build_trainer(s, dev, B, T)
imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
s.eval()
d = s.sample(B, device=dev)
buf = append_buffer(reset_buffer(dev), d)
which = sys.argv[1] if len(sys.argv) > 1 else "fv"
fn = (lambda: tr.update_f_v(imgs, d, buf)) if which == "fv" else (lambda: tr.update_sampler(buf, 1))
fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fn()
torch.cuda.synchronize()
print(which, "ms per call", (time.perf_counter() - t0) / 5 * 1e3)
