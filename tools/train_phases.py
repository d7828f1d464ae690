"""GPU box: wall-clock split of the DxMI train step (B=256, T=10) into its phases; PAIR=0/1 switches the one-forward TD target +
prediction of round 5 (DxMI_Trainer.PAIR_TD_FORWARD), both settings alternate in one process."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
import bench
from models.DxMI.trainer import append_buffer, reset_buffer
from models.DxMI.replay import TransitionRing
dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 256)), 10
s = bench.build_sampler(dev, T)
tr = bench.build_trainer(s, dev, B, T)
imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
def sync(): torch.cuda.synchronize(); return time.perf_counter()
from dxmi_hip import ops
ops.tune_for_throughput(True)          # as train_cifar10.py
ring = TransitionRing(1, T, B, (3, 32, 32), dev)
bench.train_step(tr, s, imgs, dev, ring)
for rep in range(2):
    for pair in (False, True):
        type(tr).PAIR_TD_FORWARD = pair
        bench.train_step(tr, s, imgs, dev, ring)
        acc = {"sample": 0, "append": 0, "update_f_v": 0, "update_sampler": 0}
        n = 4
        for _ in range(n):
            t0 = sync(); s.eval(); d = s.sample(B, device=dev, out=ring.next_slot())
            t1 = sync(); buf = append_buffer(ring, d)
            t2 = sync(); tr.update_f_v(imgs, d, buf)
            t3 = sync(); tr.update_sampler(buf, 1)
            t4 = sync(); reset_buffer(dev, ring=ring)
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += v
        print(f"PAIR_TD_FORWARD={pair}:", {k: round(1e3 * v / n, 1) for k, v in acc.items()}, "ms; total", round(1e3 * sum(acc.values()) / n, 1))
