"""GPU box, under rocprofv3 --kernel-trace --stats: N hipGraph replays of the CIFAR-10 T=10 generation call or DxMI train step at a
small per-rank batch.   MODE=gen|train B=32 N=10 python3 tools/graph_profile.py
Prints wall ms per step; the kernel_stats.csv then gives the kernel time per step (total / (N + warm-up))."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
from dxmi_hip import ops
from models.DxMI.replay import TransitionRing
dev = torch.device("cuda:0")
B, T, N = int(os.environ.get("B", 32)), int(os.environ.get("T", 10)), int(os.environ.get("N", 10))
mode, graph = os.environ.get("MODE", "train"), os.environ.get("GRAPH", "1") == "1"
s = bench.build_sampler(dev, T)
s.use_graph = graph
if mode == "gen":
    step = lambda: s.sample(B, device=dev)
else:
    ops.tune_for_throughput(True)
    tr = bench.build_trainer(s, dev, B, T)
    tr.use_graphs = graph
    ring = TransitionRing(1, T, B, (3, 32, 32), dev)
    imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
    step = lambda: bench.train_step(tr, s, imgs, dev, ring)
W = 3
for _ in range(W):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
print(json.dumps({"mode": mode, "graph": graph, "B": B, "T": T, "timed_steps": N, "warmup_steps": W, "wall_ms_per_step": round(1e3 * (time.perf_counter() - t0) / N, 2)}))
