"""GPU box: phase timeline of attn_block256_kernel from the stamp build (tools/build_variant.sh abstamps attn_block.hip -DDXMI_AB_STAMPS;
run with DXMI_LIB=.../libdxmi_abstamps.so): s_memtime deltas of every wave of the first 16 workgroups between the phase boundaries."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import numpy as np
import torch
from dxmi_hip import ops, _lib
dev = "cuda:0"
torch.manual_seed(0)
N, T, C = int(os.environ.get("N", 256)), 256, 256
x = (torch.randn(N, 16, 16, C, device=dev) * 1.3 + 0.2).to(torch.bfloat16)
ws = {k: torch.randn(C, C, 1, 1, device=dev) * 0.06 for k in "qkvp"}
bs = {k: torch.randn(C, device=dev) * 0.3 for k in "qkvp"}
gamma, beta = 1 + 0.3 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)
st = ops.block_stats(x)
packed = ops.attn_block_pack(ws["q"], bs["q"], ws["k"], ws["v"], bs["v"], ws["p"], bs["p"], 0.0625)
out = torch.empty_like(x)
ost = ops.BlockStats(torch.empty((N, 8, C // 2, 2), dtype=torch.float32, device=dev), 8)
lib0 = _lib.load()
def launch():
    _lib.check(lib0.dxmi_attn_block_fwd(x.data_ptr(), st.buf.data_ptr(), st.P, gamma.data_ptr(), beta.data_ptr(), 1e-6, packed.data_ptr(),
                                        out.data_ptr(), ost.buf.data_ptr(), N, T, C, ops._stream()), "attn_block")
for _ in range(3):
    launch()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        launch()
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"graph of 20 launches: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (stamp build)")
lib = _lib.load()
buf = np.zeros((16, 8, 12), dtype=np.uint32)
fn = lib.dxmi_debug_read_ab_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, buf.nbytes) == 0
names = ["prologue: DMAs issued, statistics, first barrier", "table, second barrier, x^ fragments", "phase 1 (G x^)", "barrier + phase 2 (S)",
         "softmax", "phase 3 (PV)", "Z affine", "barrier + phase 4 (W' Z) + epilogue", "stores + statistics"]
d = np.diff(buf[:, :, :10].astype(np.int64), axis=2) & 0xFFFFFFFF          # [wg, wave, 9]
t0 = buf[:, :, 0].astype(np.int64)
print("cycles (s_memtime)  mean over 16 workgroups x 8 waves | wave 0 | wave 7")
for k in range(9):
    print(f"  {names[k]:52s} {d[:, :, k].mean():9.0f} | {d[:, 0, k].mean():9.0f} | {d[:, 7, k].mean():9.0f}")
print(f"  total {np.mean((buf[:, :, 9].astype(np.int64) - t0) & 0xFFFFFFFF):.0f}")
