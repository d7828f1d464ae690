"""GPU box: which 3x3 kernel should the small-batch train step use?  Sweep of the conv_ws_min_tiles knob (fewer (256-pixel, 128-cout)
tiles than this -> conv_pipe_kernel's 64-pixel tiles) and conv_sm_mask under hipGraph replay; ms per DxMI train step / generation call.
    B=32 python tools/small_batch_routing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
import bench
from dxmi_hip import ops
from models.DxMI.replay import TransitionRing
dev = torch.device("cuda:0")
T, B = 10, int(os.environ.get("B", 32))
for mt, sm in [(0, 9), (96, 13), (130, 13), (192, 13), (260, 13), (520, 13), (100000, 13), (260, 15)]:
    ops.set_tuning("conv_ws_min_tiles", mt)
    ops.set_tuning("conv_sm_mask", sm)
    s = bench.build_sampler(dev, T)
    s.use_graph = True
    tr = bench.build_trainer(s, dev, B, T)
    tr.use_graphs = True
    ring = TransitionRing(1, T, B, (3, 32, 32), dev)
    imgs = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
    for _ in range(3):
        bench.train_step(tr, s, imgs, dev, ring)
        s.sample(B, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        bench.train_step(tr, s, imgs, dev, ring)
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / 6 * 1e3
    t0 = time.perf_counter()
    for _ in range(6):
        s.sample(B, device=dev)
    torch.cuda.synchronize()
    t_gen = (time.perf_counter() - t0) / 6 * 1e3
    print(f"B={B} conv_ws_min_tiles={mt} conv_sm_mask={sm}: train {t_train:.2f} ms / step, generation {t_gen:.2f} ms / call")
    del tr, s, ring
    torch.cuda.empty_cache()
