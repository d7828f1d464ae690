"""GPU box, under `rocprofv3 --kernel-trace`: the weight-gradient kernel and its reduce launch per shape (10 repetitions each, in
order); tools/wgrad_split_parse.py reads the per-dispatch durations back from the trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)
SHAPES = [(256, 32, 128, 128, 3), (256, 32, 256, 128, 3), (256, 16, 256, 256, 3), (256, 16, 512, 256, 3), (256, 8, 256, 256, 3),
          (256, 4, 256, 256, 3), (256, 16, 256, 768, 1), (256, 16, 256, 256, 1), (256, 32, 256, 128, 1), (256, 16, 512, 256, 1), (256, 16, 128, 256, 1), (256, 16, 128, 128, 3)]
if __name__ == "__main__":
    for (N, H, Cin, Cout, k) in SHAPES:
        x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16)
        out = torch.empty(Cout, Cin, k, k, device=dev)
        for _ in range(10):
            ops.conv2d_wgrad(x, dy, k, out=out, with_bias=True)
        torch.cuda.synchronize()
