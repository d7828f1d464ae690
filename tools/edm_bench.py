"""EDM generation throughput + per-kernel-family time (run on the GPU box).
    python tools/edm_bench.py imagenet64_T10 100 | lsun_bedroom_T4 16"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import configs_builtin
from models.cm.script_util import create_model_and_diffusion
from models.DxMI.openai_diffusion import OpenAIDiffusion
from dxmi_hip import ops

name, B = sys.argv[1], int(sys.argv[2])
cfg = configs_builtin.get(name)
torch.manual_seed(0)
net, diffusion = create_model_and_diffusion(**cfg.diffusion)
for n, p in net.named_parameters():   # zero_module layers get weights so no work is skipped
    if p.abs().max() == 0:
        torch.nn.init.normal_(p, std=0.02)
s = OpenAIDiffusion(net, diffusion, **cfg.sampler)
net.to("cuda:0").eval()
with torch.no_grad():
    s.sample(B, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        d = s.sample(B, device="cuda:0")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print(f"{name}: B={B} T={s.n_timesteps}: {dt*1e3:.1f} ms/batch, {B/dt:.1f} img/s, finite={bool(torch.isfinite(d['sample']).all())}")
if len(sys.argv) > 3:
    ops_mod = ops
    prof = ops.OpProfiler()
    ops_mod.PROFILER = prof
    with torch.no_grad():
        s.sample(B, device="cuda:0")
    torch.cuda.synchronize()
    ops_mod.PROFILER = None
    tot = 0
    for kid, v in sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"]):
        print(f"  conv kid {kid}: {v['launches']} launches {v['ms']:.1f} ms {v['flops']/v['ms']/1e9:.0f} TFLOP/s {v['bytes']/v['ms']/1e6:.0f} GB/s")
        tot += v["ms"]
    print(f"  conv total {tot:.1f} ms of {dt*1e3:.1f}")
