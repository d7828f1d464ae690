"""Seeded sweep over conv / linear shapes on the GPU (edge cases of the tile dispatch: ragged batches, cout tiles that are
half empty, every kernel family — pipelined 3x3, stride 2, upsampled, streaming 1x1, stem, small GEMM) against torch fp32
on the same bf16-rounded operands.  Tolerance: one bf16 rounding of the output (rel-L2 <= 4e-3)."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from dxmi_hip import ops as o
    o.device_check()
    return o


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _cases():
    rnd = random.Random(20241002)
    cases = []
    for _ in range(36):
        k = rnd.choice([1, 1, 3, 3, 3])
        H = rnd.choice([4, 8, 16, 32, 64])
        N = rnd.choice([1, 2, 3, 5, 7, 9, 17])
        if H == 64:
            N = min(N, 3)
        C0 = rnd.choice([64, 128, 192, 256, 320, 384])
        C1 = rnd.choice([0, 0, 0, 64, 128, 192])
        if (C0 + C1) % 64:
            C1 = 0
        if C0 % 32:
            C0 = 64
        Cout = rnd.choice([64, 128, 192, 256, 384, 576])
        stride = 2 if (k == 3 and H >= 8 and rnd.random() < 0.2) else 1
        ups = (k == 3 and stride == 1 and H <= 32 and rnd.random() < 0.2)
        fuse = rnd.choice(["none", "bias", "bias+res", "bias+vec", "bias+res+vec+act"])
        cases.append((N, C0, C1, Cout, H, k, stride, ups, fuse))
    return cases


@pytest.mark.parametrize("N,C0,C1,Cout,H,k,stride,ups,fuse", _cases())
def test_conv_sweep(ops, N, C0, C1, Cout, H, k, stride, ups, fuse):
    g = torch.Generator().manual_seed(N * 1000003 + C0 * 131 + C1 * 17 + Cout * 7 + H + k)
    Cin = C0 + C1
    x = bf(torch.randn(N, Cin, H, H, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xi, w, b if "bias" in fuse else None, stride=stride, padding=k // 2)
    kw = {}
    if "res" in fuse:
        res = bf(torch.randn(ref.shape, generator=g))
        ref = ref + res
        kw["residual"] = nhwc(res)
    if "vec" in fuse:
        vec = torch.randn(N, Cout, generator=g)
        ref = ref + vec[:, :, None, None]
        kw["addvec"] = vec.to(DEV)
    if "act" in fuse:
        ref = F.leaky_relu(ref, 0.2)
        kw["act"] = ops.ACT_LEAKY02
    y = ops.conv2d(nhwc(x[:, :C0]), ops.pack_conv_weight(w.to(DEV)), in1=nhwc(x[:, C0:]) if C1 else None,
                   bias=b.to(DEV) if "bias" in fuse else None, stride=stride, pad=k // 2, upsample=ups, **kw)
    torch.cuda.synchronize()
    assert nchw(y).shape == ref.shape
    assert rel_l2(nchw(y), ref) < 4e-3


@pytest.mark.parametrize("N,Cout,H,act", [(1, 64, 32, False), (3, 128, 32, True), (5, 192, 64, False), (2, 256, 16, True), (9, 320, 8, False)])
def test_stem_conv_sweep(ops, N, Cout, H, act):
    g = torch.Generator().manual_seed(N + Cout + H)
    x = torch.randn(N, 3, H, H, generator=g)
    w = bf(torch.randn(Cout, 3, 3, 3, generator=g) / math.sqrt(27))
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(bf(x), w, b, padding=1)
    if act:
        ref = F.leaky_relu(ref, 0.2)
    y = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w.to(DEV), k27=True), bias=b.to(DEV), act=ops.ACT_LEAKY02 if act else ops.ACT_NONE)
    assert rel_l2(nchw(y), ref) < 4e-3


@pytest.mark.parametrize("P,K,M", [(1, 32, 4), (31, 64, 36), (33, 128, 128), (100, 768, 1536), (256, 512, 2816), (257, 1024, 20), (5000, 64, 64)])
def test_linear_sweep(ops, P, K, M):
    g = torch.Generator().manual_seed(P + K + M)
    x = torch.randn(P, K, generator=g)
    w = bf(torch.randn(M, K, generator=g) / math.sqrt(K))
    b = torch.randn(M, generator=g)
    ref = F.linear(bf(F.silu(x)), w, b)
    got = ops.linear(x.to(DEV), ops.pack_conv_weight(w.to(DEV)), b.to(DEV), pre_act=ops.ACT_SILU).cpu()
    assert rel_l2(got, ref) < 2e-3
