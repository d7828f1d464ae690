import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-by-maxentirl_amd"))
from dxmi_hip import ops
dev="cuda:0"
x = torch.randn(256, 32, 32, 128, device=dev).to(torch.bfloat16)
pw = ops.pack_conv_weight(torch.randn(3, 128, 3, 3, device=dev) * 0.05)
b = torch.randn(3, device=dev)
for _ in range(5): ops.conv2d(x, pw, bias=b, out_nchw_f32=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.conv2d(x, pw, bias=b, out_nchw_f32=True)
e1.record(); torch.cuda.synchronize()
print("conv_out 128->3 @32 B=256: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
