"""GPU box: per-shape timing of every conv launch in one U-Net forward (HIP events)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
import bench
from dxmi_hip import ops

class Prof(ops.OpProfiler):
    def launch_conv(self, d):
        n0 = len(self.records)
        super().launch_conv(d)
        cls, kid, fl, by, e0, e1 = self.records[n0]
        self.records[n0] = (cls, (kid, d.N, d.OH, d.C0 + d.C1, d.Cout, d.ksize, d.stride, d.upsample, 'res' if d.residual else ('tv' if d.addvec else '-')), fl, by, e0, e1)

    def bracket(self, cls, name, flops, nbytes, fn):
        if not cls.startswith("conv"):
            return fn()
        return super().bracket(cls, name, flops, nbytes, fn)

dev = torch.device("cuda:0")
s = bench.build_sampler(dev, 10)
B = int(os.environ.get("B", 256))
x = torch.randn(B, 3, 32, 32, device=dev); t = torch.full((B,), 100.0, device=dev)
with torch.no_grad():
    for _ in range(3): s.net(x, t)
    prof = Prof(); ops.PROFILER = prof
    for _ in range(5): s.net(x, t)
    torch.cuda.synchronize(); ops.PROFILER = None
summ = prof.summary()
tot = sum(v["ms"] for v in summ.values())
print(f"{'kernel,N,OH,Cin,Cout,k,s,up':40s} {'n':>4s} {'us/launch':>10s} {'TFLOP/s':>8s} {'%conv':>6s}")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{str(k):48s} {v['launches']//5:4d} {1e3*v['ms']/v['launches']:10.1f} {v['flops']/(v['ms']*1e-3)/1e12:8.0f} {100*v['ms']/tot:6.1f}")
print("conv total ms/forward", tot / 5)
