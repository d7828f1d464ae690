"""GPU box: does hipGraph capture of VARSampler.sample work with the ctypes launches, and what does it buy?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
dev = torch.device("cuda:0")
s = bench.build_sampler(dev, 10)
B = 256
with torch.no_grad():
    for _ in range(2): d = s.sample(B, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): d = s.sample(B, device=dev)
    torch.cuda.synchronize()
    print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): s.sample(B, device=dev)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        out = s.sample(B, device=dev)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a = out["sample"].clone()
    g.replay(); torch.cuda.synchronize()
    b = out["sample"].clone()
    print("finite", bool(torch.isfinite(a).all()), "different draws per replay", bool((a != b).any()), a.std().item(), b.std().item())
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print("graph ms/step", (time.perf_counter() - t0) / 5 * 1e3)
