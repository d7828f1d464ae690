"""GPU box: cProfile of the HOST side of one generation step and one train step (every dxmi_* kernel call stubbed out)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
from dxmi_hip import _lib, ops
from models.DxMI.replay import TransitionRing

dev = torch.device("cuda", 0)
sampler = bench.build_sampler(dev, 10)
B = 256
ops.tune_for_throughput(True)
tr = bench.build_trainer(sampler, dev, B, 10)
ring = TransitionRing(1, 10, B, (3, 32, 32), dev)
imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
for _ in range(2):
    bench.train_step(tr, sampler, imgs, dev, ring)
torch.cuda.synchronize()
real = _lib.load()


class Stub:
    def __getattr__(self, name):
        f = getattr(real, name)
        if name.endswith(("_bytes", "_partials", "_supported", "_slices")) or name in ("dxmi_mt_blocks", "dxmi_last_error", "dxmi_conv2d_kernel_id",
                                                                                              "dxmi_device_check", "dxmi_version", "dxmi_get_tuning", "dxmi_set_tuning"):
            return f
        return lambda *a: 0


_lib._lib = Stub()
which = sys.argv[1] if len(sys.argv) > 1 else "train"
fn = (lambda: bench.train_step(tr, sampler, imgs, dev, ring)) if which == "train" else (lambda: sampler.sample(B, device=dev))
fn()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    fn()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
