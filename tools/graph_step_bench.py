"""GPU box: CIFAR-10 T=10 generation call and DxMI train step, eager vs hipGraph replay (dxmi_hip/graph.py), at the per-rank batches
the reference runs (train_cifar10.py:298-301: 256 // N).  Prints per batch: wall ms per step (synchronised), the host time to ISSUE
a step (time until the python call returns, GPU still busy) and the GPU-side time between events.
    B="32 128 256" python tools/graph_step_bench.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from dxmi_hip import ops  # noqa: E402
from models.DxMI.replay import TransitionRing  # noqa: E402

dev = torch.device("cuda:0")
T = int(os.environ.get("T", 10))
STEPS = int(os.environ.get("STEPS", 6))


def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    issue = 0.0
    for _ in range(n):
        ti = time.perf_counter()
        fn()
        issue += time.perf_counter() - ti
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"wall_ms": round(1e3 * wall / n, 2), "host_issue_ms": round(1e3 * issue / n, 2), "gpu_ms": round(e0.elapsed_time(e1) / n, 2)}


out = {}
for B in [int(b) for b in os.environ.get("B", "32 128 256").split()]:
    row = {}
    for mode in ("eager", "graph"):
        s = bench.build_sampler(dev, T)
        s.use_graph = mode == "graph"
        torch.cuda.manual_seed(1)
        for _ in range(3):
            s.sample(B, device=dev)
        row[f"gen_{mode}"] = timed(lambda: s.sample(B, device=dev), STEPS)
        ops.tune_for_throughput(True)
        tr = bench.build_trainer(s, dev, B, T)
        tr.use_graphs = mode == "graph"
        ring = TransitionRing(1, T, B, (3, 32, 32), dev)
        imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
        for _ in range(3):
            bench.train_step(tr, s, imgs, dev, ring)
        row[f"train_{mode}"] = timed(lambda: bench.train_step(tr, s, imgs, dev, ring), STEPS)
        ops.tune_for_throughput(False)
        del tr, s, ring
        torch.cuda.empty_cache()
    out[B] = row
    print(B, json.dumps(row), flush=True)
