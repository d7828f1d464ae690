"""GPU box: same-process A/B of the fused TD step (DxMI_Trainer.FUSED_TD_STEP) under hipGraph replay: ms per train step, alternating.
    B="32 256" python tools/td_fused_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
from dxmi_hip import ops
from models.DxMI.replay import TransitionRing
from models.DxMI.trainer import DxMI_Trainer
dev = torch.device("cuda:0")
T = 10
ops.tune_for_throughput(True)
for B in [int(b) for b in os.environ.get("B", "32 256").split()]:
    runs = {}
    for fused in (False, True):
        DxMI_Trainer.FUSED_TD_STEP = fused
        s = bench.build_sampler(dev, T)
        s.use_graph = True
        tr = bench.build_trainer(s, dev, B, T)
        tr.use_graphs = True
        ring = TransitionRing(1, T, B, (3, 32, 32), dev)
        imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
        for _ in range(3):
            bench.train_step(tr, s, imgs, dev, ring)
        runs[fused] = (tr, s, ring, imgs)
    res = {False: [], True: []}
    for rep in range(4):
        for fused in (False, True):
            tr, s, ring, imgs = runs[fused]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                bench.train_step(tr, s, imgs, dev, ring)
            torch.cuda.synchronize()
            res[fused].append(round((time.perf_counter() - t0) / 5 * 1e3, 2))
    print(f"B={B}: generic TD loop {res[False]} ms   fused TD step {res[True]} ms")
    del runs
    torch.cuda.empty_cache()
