"""GPU box: HOST cost of one generation / train step per rank — the Python + ctypes + allocator time with every dxmi_* kernel call
stubbed out (returns 0 without launching), against the real step time on the GPU.  An 8-rank node runs eight such launch
streams on the host cores it exposes (the driver's bench saw 16 usable cores); host_ms x ranks / cores must stay well under
the GPU step time for the ranks not to starve their GPUs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
from dxmi_hip import _lib, ops

dev = torch.device("cuda", 0)
sampler = bench.build_sampler(dev, 10)
B = 256


def gen():
    return sampler.sample(B, device=dev)


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n           # time until the last launch is queued
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    return host, wall


with torch.no_grad():
    h_real, w_real = timed(gen, 5)
    real = _lib.load()

    class Stub:
        """every kernel entry point returns DXMI_OK without launching; size queries go to the real library"""
        def __getattr__(self, name):
            f = getattr(real, name)
            if name.endswith("_bytes") or name.endswith("_partials") or name.endswith("_supported") or name in ("dxmi_mt_blocks", "dxmi_last_error", "dxmi_conv2d_kernel_id", "dxmi_device_check", "dxmi_version"):
                return f
            return lambda *a: 0
    _lib._lib = Stub()
    h_stub, w_stub = timed(gen, 5)
    _lib._lib = real
cores = bench.usable_cores()
print(f"generation step (256 images x T=10): GPU wall {w_real*1e3:.1f} ms; host time to queue it {h_real*1e3:.1f} ms; "
      f"host time with kernels stubbed {w_stub*1e3:.1f} ms  -> one rank keeps {w_stub/w_real:.2f} of a core busy; "
      f"8 ranks on {cores} usable cores: {8*w_stub/w_real/cores:.2f} of the host")


# ---- the DxMI train step (bench.py's second leg): host time with the kernels stubbed against the real step
from models.DxMI.replay import TransitionRing
ops.tune_for_throughput(True)
tr = bench.build_trainer(sampler, dev, B, 10)
ring = TransitionRing(1, 10, B, (3, 32, 32), dev)
imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1


def train():
    return bench.train_step(tr, sampler, imgs, dev, ring)


for _ in range(2):
    train()
h_real, w_real = timed(train, 4)
_lib._lib = Stub()
h_stub, w_stub = timed(train, 4)
_lib._lib = real
print(f"train step: GPU wall {w_real*1e3:.1f} ms; host time with kernels stubbed {w_stub*1e3:.1f} ms -> {w_stub/w_real:.2f} of a core per rank")
