"""GPU box, stamp build (make -C diffusion-by-maxentirl_amd/csrc STAMPS=1): cycles MFMA wave 0 of every conv_ws workgroup waits in
each step barrier (s_memtime ticks), for one layer: python tools/ws_stamps.py CIN COUT H RES"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops, _lib
dev = "cuda:0"
cin, cout, h, res = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 128, 32, 0))]
B = 256
x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
pw = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
bias = torch.randn(cout, device=dev)
r = torch.randn(B, h, h, cout, device=dev).to(torch.bfloat16) if res else None
out = torch.empty(B, h, h, cout, device=dev, dtype=torch.bfloat16)
import time
t_end = time.time() + 2.5            # >= 2 s of back-to-back launches: the clock the chip settles at under this load
while time.time() < t_end:
    for _ in range(200):
        ops.conv2d(x, pw, bias=bias, residual=r, out=out)
    torch.cuda.synchronize()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((256, 176), dtype=np.uint32)
lib.dxmi_debug_read_ws_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
w = buf.astype(np.int64)
np.set_printoptions(linewidth=250, suppress=True)
S = 9 * cin // 32
n = min(160, 2 * S)
med = np.median(w[:, :n], axis=0)
print(f"{cin}->{cout} @{h} res={res}: steps per tile {S}; median barrier wait per step (ticks), first {n} steps:")
for i in range(0, n, 9):
    print(f"  steps {i:3d}..{i+8:3d}:", " ".join(f"{v:6.0f}" for v in med[i:i + 9]))
print("sum over a tile (median wg):", med[:S].sum(), " second tile:", med[S:2 * S].sum() if n >= 2 * S else "-")
# tile-switch timeline (second tile of every workgroup): ticks relative to the loader's E2 exit
ts = w[:, 150:158]
ok = ts[:, 0] > 0
rel = (ts[ok] - ts[ok][:, [0]]) & 0xFFFFFFFF
rel = np.where(rel > 1 << 31, rel - (1 << 32), rel)
print("after E2 (ticks, median): loader E2 exit 0, loader at B_0", np.median(rel[:, 1]), "| bulk E2 exit", np.median(rel[:, 2]),
      "bulk at B_0 / B_1 / B_2", np.median(rel[:, 3]), np.median(rel[:, 4]), np.median(rel[:, 5]), "| MFMA wave 0 E2 exit", np.median(rel[:, 7]))
fine = w[:, 160:166]
okf = ok & (fine[:, 0] > 0) & (fine[:, 3] > 0)
relf = (fine[okf] - ts[okf][:, [0]]) & 0xFFFFFFFF
print("bulk wave 6 after E2 (ticks, median): loop top, -, -, after fetch_table, after halo(2), after first piece:",
      [float(np.median(relf[:, i])) for i in (0, 3, 4, 5)])
# in-kernel clock (guide, DVFS give-back item 6): d(s_memtime) / d(s_memrealtime) x 100 MHz over the K loops of a workgroup
ck = w[:, 168:172]
okc = (ck[:, 0] > 0) & (ck[:, 2] > 0)
dt = (ck[okc, 2] - ck[okc, 0]) & 0xFFFFFFFF
dr = (ck[okc, 3] - ck[okc, 1]) & 0xFFFFFFFF
print("in-kernel clock (median over workgroups): %.0f MHz  (kernel body %.1f us)" % (np.median(dt / np.maximum(dr, 1)) * 100.0, np.median(dr) / 100.0))
# launch timeline from the 100 MHz real-time counter: kernel entry -> P0 -> last E2 -> after the final drain (wave 0 of every workgroup)
rt = w[:, [173, 169, 171, 175]]
okr = (rt > 0).all(axis=1)
r0 = rt[okr][:, 0].min()
print("real time (us, median over workgroups): entry %.2f, P0 %.2f, last E2 %.2f, end %.2f; kernel span first entry -> last end %.2f us" % (
    np.median(rt[okr][:, 0] - r0) / 100.0, np.median(rt[okr][:, 1] - r0) / 100.0, np.median(rt[okr][:, 2] - r0) / 100.0,
    np.median(rt[okr][:, 3] - r0) / 100.0, (rt[okr][:, 3].max() - r0) / 100.0))
te = w[:, [158, 159, 166, 157]]
oke = (te > 0).all(axis=1)
d = (te[oke][:, 1:] - te[oke][:, :-1]) & 0xFFFFFFFF
print("first tile end, MFMA wave 0 (ticks, median): wait at E1 %.0f, epilogue %.0f, wait at E2 %.0f" % tuple(np.median(d, axis=0)))
