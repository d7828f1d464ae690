"""Per-kernel numerics of the HIP library against fp32 torch-CPU restatements of the same op.

Inputs are rounded to bf16 first, so the only differences are accumulation order and the final
bf16 rounding of the output: tolerances are stated per test.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16).float()


def nhwc(x):  # NCHW fp32 cpu -> NHWC bf16 cuda
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw(y):  # NHWC bf16 cuda -> NCHW fp32 cpu
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.fixture(scope="module")
def ops():
    from dxmi_hip import ops as o
    o.device_check()
    return o


CONV_CASES = [
    # N, Cin, Cout, H, k, stride, upsample, variant
    (2, 128, 128, 32, 3, 1, False, 0),
    (2, 128, 128, 32, 3, 1, False, 1),
    (3, 256, 256, 16, 3, 1, False, 0),
    (5, 256, 256, 8, 3, 1, False, 0),     # SUBS=4 with a ragged last image group
    (18, 256, 256, 4, 3, 1, False, 0),    # SUBS=16, ragged
    (2, 128, 256, 16, 1, 1, False, 0),    # 1x1
    (2, 256, 768, 16, 1, 1, False, 1),    # fused qkv
    (2, 128, 128, 32, 3, 2, False, 0),    # DDPM downsample, pad (0,1,0,1)
    (3, 256, 256, 8, 3, 2, False, 0),
    (2, 256, 256, 8, 3, 1, True, 0),      # nearest x2 upsample + conv
    (2, 256, 256, 16, 3, 1, True, 1),
    (2, 128, 3, 32, 3, 1, False, 0),      # conv_out style narrow head
    (1, 64, 96, 64, 3, 1, False, 0),      # Cout % 128 != 0 -> narrow kernel, 64x64 image
]


@pytest.mark.parametrize("N,Cin,Cout,H,k,stride,ups,variant", CONV_CASES)
def test_conv2d(ops, N, Cin, Cout, H, k, stride, ups, variant):
    g = torch.Generator().manual_seed(1234 + Cin + Cout + H)
    x = bf(torch.randn(N, Cin, H, H, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    if stride == 2:
        ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w, b, stride=2)
        pad, pad_br = 0, 1
    else:
        ref = F.conv2d(xi, w, b, padding=k // 2)
        pad, pad_br = k // 2, k // 2
    pw = ops.pack_conv_weight(w.to(DEV))
    y = ops.conv2d(nhwc(x), pw, bias=b.to(DEV), stride=stride, pad=pad, pad_br=pad_br, upsample=ups, variant=variant)
    torch.cuda.synchronize()
    got = nchw(y)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    # output rounded to bf16: <= 2^-8 relative + accumulation-order noise
    assert err <= 2e-2 * ref.abs().max().item() and rel_l2(got, ref) < 4e-3, (err, rel_l2(got, ref))


def test_conv2d_epilogue_fusions(ops):
    """bias + per-(n,co) temb term + residual + concat of two sources + activation."""
    g = torch.Generator().manual_seed(7)
    N, C0, C1, Cout, H = 3, 256, 128, 128, 16
    x0, x1 = bf(torch.randn(N, C0, H, H, generator=g)), bf(torch.randn(N, C1, H, H, generator=g))
    w = bf(torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1)))
    b = torch.randn(Cout, generator=g)
    tv = torch.randn(N, 5 * Cout, generator=g)  # wider row: exercises addvec_ld
    res = bf(torch.randn(N, Cout, H, H, generator=g))
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b, padding=1) + tv[:, Cout:2 * Cout, None, None] + res
    ref = F.leaky_relu(ref, 0.2)
    pw = ops.pack_conv_weight(w.to(DEV))
    tvd = tv.to(DEV)
    y = ops.conv2d(nhwc(x0), pw, in1=nhwc(x1), bias=b.to(DEV), addvec=tvd[:, Cout:2 * Cout], residual=nhwc(res),
                   act=ops.ACT_LEAKY02)
    got = nchw(y)
    assert rel_l2(got, ref) < 4e-3


def test_conv2d_k27_image_conv(ops):
    """3-channel NCHW fp32 image conv (conv_in / value conv1) through the K=27 im2col path."""
    g = torch.Generator().manual_seed(11)
    N, Cout, H = 5, 128, 32
    x = torch.randn(N, 3, H, H, generator=g)
    w = bf(torch.randn(Cout, 3, 3, 3, generator=g) / math.sqrt(27))
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(bf(x), w, b, padding=1)
    pw = ops.pack_conv_weight(w.to(DEV), k27=True)
    y = ops.conv2d(x.to(DEV), pw, bias=b.to(DEV))
    assert rel_l2(nchw(y), ref) < 4e-3


def test_conv2d_nchw_f32_out(ops):
    g = torch.Generator().manual_seed(12)
    N, Cin, H = 3, 128, 32
    x = bf(torch.randn(N, Cin, H, H, generator=g))
    w = bf(torch.randn(3, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(3, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    pw = ops.pack_conv_weight(w.to(DEV))
    y = ops.conv2d(nhwc(x), pw, bias=b.to(DEV), out_nchw_f32=True)
    assert y.shape == (N, 3, H, H) and y.dtype == torch.float32
    # fp32 output: only accumulation-order noise
    assert rel_l2(y.cpu(), ref) < 1e-5


def test_conv2d_dgrad_packing(ops):
    """transpose_flip packing turns the forward kernel into the data-gradient operator."""
    g = torch.Generator().manual_seed(13)
    N, Cin, Cout, H = 2, 128, 256, 16
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    gy = bf(torch.randn(N, Cout, H, H, generator=g))
    x = torch.zeros(N, Cin, H, H, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(gy)
    pw = ops.pack_conv_weight(w.to(DEV), transpose_flip=True)
    gx = ops.conv2d(nhwc(gy), pw)
    assert rel_l2(nchw(gx), x.grad) < 4e-3


GN_CASES = [
    # N, C0, C1, H, silu, eps
    (3, 128, 0, 32, True, 1e-6),
    (2, 256, 0, 16, True, 1e-6),
    (2, 256, 0, 16, False, 1e-6),
    (2, 256, 128, 32, True, 1e-6),   # concat, 12 channels/group straddling the boundary
    (2, 256, 256, 16, True, 1e-6),
    (3, 128, 128, 32, True, 1e-6),
    (5, 256, 256, 4, True, 1e-6),
    (2, 256, 0, 8, True, 1e-5),
    (1, 192, 0, 64, True, 1e-5),     # EDM level-0 shape: 6 channels/group -> dispatched to the generic two-kernel path
]


@pytest.mark.parametrize("N,C0,C1,H,silu,eps", GN_CASES)
def test_groupnorm_silu(ops, N, C0, C1, H, silu, eps):
    C = C0 + C1
    g = torch.Generator().manual_seed(99 + C + H)
    x = bf(torch.randn(N, C, H, H, generator=g) * 2.0 + 0.5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    x0 = nhwc(x[:, :C0])
    x1 = nhwc(x[:, C0:]) if C1 else None
    y = ops.groupnorm_silu(x0, gamma.to(DEV), beta.to(DEV), in1=x1, eps=eps, silu=silu)
    got = nchw(y)
    assert (got - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert rel_l2(got, ref) < 4e-3


@pytest.mark.parametrize("N,T,C,heads", [(2, 256, 256, 1), (3, 16, 256, 1), (2, 64, 128, 2), (1, 1024, 128, 2), (2, 200, 64, 1)])
def test_attention(ops, N, T, C, heads):
    g = torch.Generator().manual_seed(5 + T + C)
    qkv = bf(torch.randn(N, T, 3 * C, generator=g))
    D = C // heads
    scale = 1.0 / math.sqrt(D)
    q, k, v = qkv.split(C, dim=2)
    q = q.view(N, T, heads, D).transpose(1, 2)
    k = k.view(N, T, heads, D).transpose(1, 2)
    v = v.view(N, T, heads, D).transpose(1, 2)
    w = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    ref = (w @ v).transpose(1, 2).reshape(N, T, C)
    y = ops.attention(qkv.to(torch.bfloat16).to(DEV), heads, scale)
    got = y.float().cpu()
    # P is rounded to bf16 before PV: ~2^-9 relative per term
    assert rel_l2(got, ref) < 8e-3, rel_l2(got, ref)


@pytest.mark.parametrize("order", [0, 1])
def test_timestep_embedding(ops, order):
    t = torch.tensor([616.734131, 28.2926369, 1.50696171e-4, 0.0, 999.0])
    dim = 128
    half = dim // 2
    if order == 0:
        emb = math.log(10000) / (half - 1)
        freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
        args = t[:, None] * freqs[None]
        ref = torch.cat([torch.sin(args), torch.cos(args)], 1)
    else:
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        args = t[:, None] * freqs[None]
        ref = torch.cat([torch.cos(args), torch.sin(args)], 1)
    got = ops.timestep_embedding(t.to(DEV), dim, order).cpu()
    # arguments reach ~1e3 rad: fp32 argument rounding dominates
    assert (got - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("P,K,M", [(7, 128, 512), (256, 512, 4992), (300, 512, 512), (4, 256, 1)])
def test_linear(ops, P, K, M):
    g = torch.Generator().manual_seed(P + K + M)
    x = torch.randn(P, K, generator=g)
    w = bf(torch.randn(M, K, generator=g) / math.sqrt(K))
    b = torch.randn(M, generator=g)
    ref = F.silu(F.linear(bf(F.silu(x)), w, b))
    pw = ops.pack_conv_weight(w.to(DEV))
    got = ops.linear(x.to(DEV), pw, b.to(DEV), pre_act=ops.ACT_SILU, post_act=ops.ACT_SILU).cpu()
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)


@pytest.mark.parametrize("P,K,M,bias", [(16, 30720, 768, False), (100, 8192, 192, True), (5, 4096, 64, False)])
def test_linear_long_k_split(ops, P, K, M, bias):
    """Skinny products with a long K (the data gradient of the ADM nets' concatenated emb_layers: 16 x 30 k -> 768) take the split-K
    form (dxmi_linear_splitk: slices of K in parallel, summed in slice order): against fp32 on the same bf16 operands, bitwise
    reproducible, and a shape that does not split is untouched."""
    from dxmi_hip import _lib
    S = _lib.load().dxmi_linear_splitk_slices(P, K, M)
    assert S > 1 and _lib.load().dxmi_linear_splitk_slices(256, 512, 4992) == 1 and _lib.load().dxmi_linear_splitk_slices(16, 30720, 770) == 0
    g = torch.Generator().manual_seed(P + K + M)
    x = torch.randn(P, K, generator=g)
    w = bf(torch.randn(M, K, generator=g) / math.sqrt(K))
    b = torch.randn(M, generator=g) if bias else None
    ref = F.linear(bf(x), w, b)
    pw = ops.pack_conv_weight(w.to(DEV))
    got = ops.linear(x.to(DEV), pw, b.to(DEV) if bias else None)
    assert rel_l2(got.cpu(), ref) < 2e-3, rel_l2(got.cpu(), ref)
    assert torch.equal(got, ops.linear(x.to(DEV), pw, b.to(DEV) if bias else None))


def test_var_step_and_gather(ops):
    g = torch.Generator().manual_seed(3)
    N, T = 6, 10
    x, eps, z = (torch.randn(N, 3, 32, 32, generator=g) for _ in range(3))
    tabs = [torch.rand(T, generator=g) + 0.5 for _ in range(3)]
    lb = torch.randn(T, generator=g)
    t = torch.tensor([0, 9, 3, 3, 7, 1])
    tau, xm, cm, sg = ops.var_gather_sched(t.to(DEV), *[a.to(DEV) for a in tabs], lb.to(DEV))
    # integer index path: bit-exact gathers
    assert torch.equal(tau.cpu(), tabs[0][t]) and torch.equal(xm.cpu(), tabs[1][t]) and torch.equal(cm.cpu(), tabs[2][t])
    sigma = torch.exp(lb[t])
    assert torch.allclose(sg.cpu(), sigma, rtol=1e-6, atol=0)
    xs = x * tabs[1][t][:, None, None, None]
    control = tabs[2][t][:, None, None, None] * eps
    mean = xs + control
    xn = mean + sigma[:, None, None, None] * z
    logp = torch.distributions.Normal(mean, sigma[:, None, None, None]).log_prob(xn).mean(-1).mean(-1).mean(-1)
    out = ops.var_step(x.to(DEV), eps.to(DEV), z.to(DEV), xm, cm, sigma.to(DEV))
    assert torch.allclose(out[0].cpu(), xn, rtol=1e-6, atol=1e-6)
    assert torch.allclose(out[1].cpu(), mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(out[2].cpu(), control, rtol=1e-6, atol=1e-7)
    assert torch.allclose(out[3].cpu(), logp, rtol=1e-5, atol=1e-5)


def test_value_tail_ops(ops):
    g = torch.Generator().manual_seed(21)
    N, C, H = 3, 256, 8
    x = bf(torch.randn(N, C, H, H, generator=g))
    ref = F.leaky_relu(F.avg_pool2d(x, 2), 0.2)
    got = nchw(ops.pool_act(nhwc(x), True, ops.ACT_LEAKY02))
    assert rel_l2(got, ref) < 4e-3
    ref2 = F.leaky_relu(x, 0.2)
    assert rel_l2(nchw(ops.pool_act(nhwc(x), False, ops.ACT_LEAKY02)), ref2) < 4e-3
    w, b = torch.randn(C, generator=g), torch.randn(1, generator=g)
    head = (F.relu(x).flatten(2).sum(2) @ w + b) * 1.5 - 0.25
    got = ops.value_head(nhwc(x), w.to(DEV), b.to(DEV), torch.tensor([1.5], device=DEV), torch.tensor([-0.25], device=DEV)).cpu().flatten()
    assert torch.allclose(got, head, rtol=1e-4, atol=1e-3)


def test_layout_roundtrip(ops):
    x = torch.randn(3, 5, 8, 8)
    y = ops.nchw_f32_to_nhwc_bf16(x.to(DEV))
    assert torch.equal(y.float().cpu(), bf(x).permute(0, 2, 3, 1))
    assert torch.equal(ops.nhwc_bf16_to_nchw_f32(y).cpu(), bf(x))


def test_cpu_tensors_are_refused(ops):
    from dxmi_hip import DxmiError
    with pytest.raises(DxmiError):
        ops.pool_act(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16), False, 0)
