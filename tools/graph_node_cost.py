"""GPU box: what one more tiny kernel costs inside a replayed hipGraph (a linear chain captured from one stream) and when the same
launches are issued eagerly: N dependent in-place adds on a 1-element / 1 M-element tensor.   python tools/graph_node_cost.py"""
import time
import torch
dev = "cuda:0"
for numel in (1, 1 << 20):
    x = torch.zeros(numel, device=dev)
    N = 4000
    for _ in range(10): x.add_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): x.add_(1.0)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / N * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N): x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / (3 * N) * 1e6
    print(f"numel {numel}: eager {eager:.2f} us per launch, graph replay {rep:.2f} us per node")
