"""GPU box: streaming GroupNorm (block statistics from the conv epilogue / the stats kernel + apply) against the one-pass
resident kernel: values, statistics, timing per shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# 1. conv_ws epilogue statistics vs the stats kernel vs torch
for (N, H, Cin, Cout, res) in [(8, 32, 128, 128, True), (8, 16, 256, 256, False), (8, 32, 256, 192, False), (256, 32, 128, 128, True), (256, 16, 256, 256, True)]:
    x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.03)
    pw = ops.pack_conv_weight(w)
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(N, H, H, Cout, device=dev).to(torch.bfloat16) if res else None
    y, st = ops.conv2d(x, pw, bias=bias, residual=r, want_stats=True)
    y2 = ops.conv2d(x, pw, bias=bias, residual=r)
    assert st is not None, "conv_ws should emit statistics here"
    assert torch.equal(y, y2)
    st2 = ops.block_stats(y)
    yf = y.float().view(N, H * H, Cout // 4, 4)
    ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], -1)      # [N, C/4, 2]
    a, b = st.buf.sum(1), st2.buf.sum(1)
    print(f"conv N{N} {H}x{H} {Cin}->{Cout} res={res}: P={st.P}/{st2.P}  epilogue-vs-torch {((a-ref).abs().max()/ref.abs().max()).item():.2e}  "
          f"kernel-vs-torch {((b-ref).abs().max()/ref.abs().max()).item():.2e}")
    t0 = timeit(lambda: ops.conv2d(x, pw, bias=bias, residual=r))
    t1 = timeit(lambda: ops.conv2d(x, pw, bias=bias, residual=r, want_stats=True))
    print(f"    conv {t0:.1f} us  with stats {t1:.1f} us")

# 2. apply vs resident
for (N, H, C0, C1, silu) in [(256, 32, 128, 0, True), (256, 32, 256, 0, True), (256, 32, 256, 128, True), (256, 32, 128, 128, True), (256, 16, 256, 0, False),
                             (256, 16, 256, 256, True), (256, 16, 256, 128, True), (256, 16, 128, 0, True), (256, 8, 256, 0, True), (256, 8, 256, 256, True), (3, 32, 128, 0, True)]:
    C = C0 + C1
    x0 = (torch.randn(N, H, H, C0, device=dev) * 1.7 + 0.3).to(torch.bfloat16)
    x1 = (torch.randn(N, H, H, C1, device=dev) * 0.6 - 0.2).to(torch.bfloat16) if C1 else None
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    s0 = ops.block_stats(x0)
    s1 = ops.block_stats(x1) if C1 else None
    ya = ops.groupnorm_silu(x0, g, b, in1=x1, silu=silu, stats=(s0, s1))
    yr = ops.groupnorm_silu(x0, g, b, in1=x1, silu=silu)
    same = (ya == yr).float().mean().item()
    err = (ya.float() - yr.float()).abs().max().item()
    big = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    def cold(fn):
        ts = []
        for _ in range(8):
            big.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]
    ta, tr_ = timeit(lambda: ops.groupnorm_silu(x0, g, b, in1=x1, silu=silu, stats=(s0, s1))), timeit(lambda: ops.groupnorm_silu(x0, g, b, in1=x1, silu=silu))
    ca, cr = cold(lambda: ops.groupnorm_silu(x0, g, b, in1=x1, silu=silu, stats=(s0, s1))), cold(lambda: ops.groupnorm_silu(x0, g, b, in1=x1, silu=silu))
    ts = timeit(lambda: ops.block_stats(x0))
    mb = N * H * H * C * 4 / 1e6
    print(f"gn N{N} {H}x{H} C={C0}+{C1} silu={silu}: identical {same:.5f} maxerr {err:.3g} | warm apply {ta:.1f} us ({mb/ta:.2f} TB/s) resident {tr_:.1f} us ({mb/tr_:.2f}) | "
          f"cold apply {ca:.1f} ({mb/ca:.2f}) resident {cr:.1f} ({mb/cr:.2f}) | stats kernel {ts:.1f} us")
