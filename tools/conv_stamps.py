"""GPU box, stamp build (make STAMPS=1; the stamps cost registers, so absolute times read ~10 % high): per-workgroup phase timeline of the dominant conv kernel from s_memtime stamps."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops, _lib
dev = "cuda:0"
B, cin, cout, h = 256, 128, 128, 32
res = int(os.environ.get("RES", 1))
x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
pw = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
bias = torch.randn(cout, device=dev)
r = torch.randn(B, h, h, cout, device=dev).to(torch.bfloat16) if res else None
out = torch.empty(B, h, h, cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv2d(x, pw, bias=bias, residual=r, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((2048, 24), dtype=np.uint64)
lib.dxmi_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
st = buf[:512].astype(np.int64)
t0 = st[:, 0].min()
cu = (st[:, 1] & 0xF); xcc = (st[:, 1] >> 32) & 0xF
rel = (st[:, 2:] - t0) / 100.0      # s_memtime: 100 MHz ticks?  print raw deltas too
print("stamp unit check: kernel span (ticks)", (st[:, 2:][st[:, 2:] > 0].max() - t0))
np.set_printoptions(linewidth=250, suppress=True)
for b in (0, 1, 8, 255, 256, 257, 264, 511):
    row = st[b, 2:]
    row = row[row > 0] - t0
    print(f"wg {b:3d} xcc {xcc[b]} cu {cu[b]:2d} start {st[b,0]-t0:7d}:", " ".join(f"{v:7d}" for v in row))
# phase statistics over all workgroups: stamps per tile = [tile start, K end, (DE: after vmcnt0, after barrier,) epi end]
per = 3
names = ["K-loop", "epilogue"]
ph = [[] for _ in range(per - 1)]
for b in range(512):
    row = st[b, 2:]; row = row[row > 0]
    for i in range(0, len(row) - per + 1, per):
        for j in range(per - 1):
            ph[j].append(row[i + j + 1] - row[i + j])
for n_, v in zip(names, ph):
    print(f"{n_:10s} cycles: median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}")
