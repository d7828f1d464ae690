"""Measurement aid (GPU box, NOT a test, NOT product): the "reference PyTorch-ROCm path" for the EDM backbone — the
oracle's op-for-op restatement of models/cm/unet.py + OpenAIDiffusion (NCHW, unfused torch ops, MIOpen/rocBLAS) run
eagerly on the same MI355X in fp32 and under fp16 autocast (the reference converts the torso to fp16).
    python tests/perf/eager_reference_edm_gpu.py [imagenet64|lsun] [batch]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import edm  # noqa: E402
from oracle.weights import formula_tensor  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "imagenet64"
if which == "imagenet64":
    cfg, T, B, sch_kw, res = edm.EDMConfig(), 10, 100, {}, 64
else:
    cfg = edm.EDMConfig(image_size=256, model_channels=256, channel_mult=(1, 1, 2, 2, 4, 4), attention_resolutions=(8, 16, 32),
                        num_classes=None)
    T, B, sch_kw, res = 4, 16, dict(stochastic_last=True, rho=4.0), 256
if len(sys.argv) > 2:
    B = int(sys.argv[2])
dev = torch.device("cuda:0")
sd = {k: formula_tensor(k, s).to(dev) for k, s in edm.state_dict_shapes(cfg).items()}
sch = edm.EDMSchedule(T, **sch_kw)
for k in ("sigmas", "sigma_up", "sigma_down", "log_betas"):
    setattr(sch, k, getattr(sch, k).to(dev))
torch.set_default_device(dev)
kw = {"y": torch.randint(0, 1000, (B,))} if cfg.num_classes else {}
x0 = torch.randn(B, 3, res, res) * 80.0
zs = [torch.randn(B, 3, res, res) for _ in range(T)]
for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("fp16-autocast", torch.autocast("cuda", dtype=torch.float16))):
    with torch.no_grad(), ctx:
        model = lambda x, t, **k: edm.unet_forward(sd, cfg, x, t, **k).float()
        fn = lambda: edm.sample(model, sch, x0, zs, **kw)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            d = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print(f"eager torch-ROCm {which} {name}: B={B} T={T}: {dt * 1e3:.1f} ms/batch -> {B / dt:.1f} images/s")
