"""GPU box: wall-clock split of the DxMI train step (B=256, T=10) into its phases."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
import bench
from models.DxMI.trainer import append_buffer, reset_buffer
dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 256)), 10
s = bench.build_sampler(dev, T)
tr = bench.build_trainer(s, dev, B, T)
imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
def sync(): torch.cuda.synchronize(); return time.perf_counter()
bench.train_step(tr, s, imgs, dev)
acc = {"sample": 0, "append": 0, "update_f_v": 0, "update_sampler": 0}
n = 3
for _ in range(n):
    t0 = sync(); s.eval(); d = s.sample(B, device=dev)
    t1 = sync(); buf = append_buffer(reset_buffer(dev), d)
    t2 = sync(); tr.update_f_v(imgs, d, buf)
    t3 = sync(); tr.update_sampler(buf, 1)
    t4 = sync()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += v
print({k: round(1e3 * v / n, 1) for k, v in acc.items()}, "ms; total", round(1e3 * sum(acc.values()) / n, 1))
