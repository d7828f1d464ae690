"""Measurement aid (GPU box, NOT a test, NOT product): the "reference PyTorch-ROCm path" of
BASELINE.json's >=3x target — the oracle's op-for-op restatement of the reference (NCHW fp32,
unfused torch ops, MIOpen/rocBLAS underneath) run eagerly on the same MI355X, T=10 generation.
    python tests/perf/eager_reference_gpu.py [batch]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from oracle import schedule as osched, unet_small as ounet, var_sampler as ovs  # noqa: E402
from oracle.weights import formula_tensor  # noqa: E402
from models.DxMI.unet_small import Model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 10
dev = torch.device("cuda:0")
net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32)
sd = {k: formula_tensor(k, v.shape).to(dev) for k, v in net.state_dict().items()}
s = osched.var_schedule(T)
sched = {k: torch.from_numpy(v).to(dev) for k, v in s.items() if k != "user_defined_eta"}
cfg = ounet.UNetSmallConfig()
# monkeypatch-free device handling: the oracle builds helper tensors on CPU; move via default device
torch.set_default_device(dev)
noise = [torch.randn(B, 3, 32, 32) for _ in range(T + 1)]
for dtype_name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16-autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
    with torch.no_grad(), ctx:
        fn = lambda: ovs.sample(lambda x, t: ounet.forward(sd, cfg, x, t).float(), sched, sched["log_betas"], noise)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print(f"eager torch-ROCm {dtype_name}: B={B} T={T}: {dt * 1e3:.1f} ms/step -> {B / dt:.0f} images/s")
