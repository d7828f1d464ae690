"""GPU box: does splitting the 256-image generation step into two 128-image half batches on two streams (captured as two
hipGraphs, replayed concurrently) beat one 256-image batch?  Per-image results are batch-independent, so this would be a pure
scheduling choice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
dev = torch.device("cuda:0")
s = bench.build_sampler(dev, 10)


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def capture(B, stream):
    g = torch.cuda.CUDAGraph()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(2): s.sample(B, device=dev)
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=stream):
        out = s.sample(B, device=dev)
    torch.cuda.synchronize()
    return g, out


with torch.no_grad():
    print("eager 1 x 256: %.2f ms" % timed(lambda: s.sample(256, device=dev)))
    print("eager 2 x 128 sequential: %.2f ms" % timed(lambda: (s.sample(128, device=dev), s.sample(128, device=dev))))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    g256, _ = capture(256, sa)
    print("graph 1 x 256: %.2f ms" % timed(lambda: g256.replay()))
    ga, _ = capture(128, sa)
    gb, _ = capture(128, sb)

    def both():
        with torch.cuda.stream(sa): ga.replay()
        with torch.cuda.stream(sb): gb.replay()
    print("graphs 2 x 128 on two streams: %.2f ms" % timed(both))
