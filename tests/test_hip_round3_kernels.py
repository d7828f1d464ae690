"""Round-3 kernels through the C-ABI: GroupNorm block statistics from the conv epilogue (dxmi_conv_desc.gn_stats) and from
the statistics kernel, the streaming GroupNorm apply pass, and the U-Net forward on that path against the one-pass path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16).float()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module")
def ops():
    from dxmi_hip import ops as o
    o.device_check()
    return o


@pytest.fixture(scope="module", autouse=True)
def kernels_under_test(ops):
    """These cases put deliberately small grids (one tile, a handful of images) on the wave-specialised kernels: switch off the
    library's small-grid re-routing (include/dxmi_hip.h: dxmi_set_tuning) for this module; test_small_grid_routing covers it."""
    old = ops.set_tuning("conv_ws_min_tiles", 0), ops.set_tuning("conv_sm_mask", 1)
    yield
    ops.set_tuning("conv_ws_min_tiles", old[0])
    ops.set_tuning("conv_sm_mask", old[1])



def _torch_block_stats(y):
    """y NHWC bf16 -> [N, C/2, 2] fp64 (sum, sum of squares) of the stored values per channel pair."""
    N, H, W, C = y.shape
    f = y.double().view(N, H * W, C // 2, 2)
    return torch.stack([f.sum((1, 3)), (f * f).sum((1, 3))], -1)


# shapes that select conv_ws_kernel<32> / <16> (incl. upsample, concat, residual, half-empty last cout tile, uneven persistent loop)
STATS_CONV_CASES = [
    # N, C0, C1, Cout, H(out), upsample, residual
    (3, 128, 0, 128, 32, False, True),
    (2, 256, 128, 128, 32, False, False),
    (2, 256, 0, 256, 32, True, False),
    (5, 256, 0, 256, 16, False, True),
    (2, 256, 256, 256, 16, False, False),
    (2, 256, 0, 192, 32, False, False),      # Cout % 128 == 64
    (70, 128, 0, 128, 32, False, True),      # 280 tiles on 256 workgroups
]


@pytest.mark.parametrize("N,C0,C1,Cout,H,ups,res", STATS_CONV_CASES)
def test_conv_epilogue_block_stats(ops, N, C0, C1, Cout, H, ups, res):
    """The statistics the conv writes are those of the bf16 tensor it stores (exact sums in fp64 within fp32 accumulation
    error), the output is bitwise the one of a launch without statistics, and both are bitwise reproducible."""
    g = torch.Generator().manual_seed(7 + N + Cout)
    IH = H // 2 if ups else H
    x0 = torch.randn(N, IH, IH, C0, generator=g).to(torch.bfloat16).to(DEV)
    x1 = torch.randn(N, IH, IH, C1, generator=g).to(torch.bfloat16).to(DEV) if C1 else None
    pw = ops.pack_conv_weight((torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.03).to(DEV))
    bias = torch.randn(Cout, generator=g).to(DEV)
    temb = torch.randn(N, Cout, generator=g).to(DEV)
    r = torch.randn(N, H, H, Cout, generator=g).to(torch.bfloat16).to(DEV) if res else None
    kw = dict(in1=x1, bias=bias, addvec=temb, residual=r, upsample=ups)
    y, st = ops.conv2d(x0, pw, want_stats=True, **kw)
    assert st is not None and st.P == (2 if H == 16 else 8) and tuple(st.buf.shape) == (N, st.P, Cout // 2, 2)
    y_plain = ops.conv2d(x0, pw, **kw)
    assert torch.equal(y, y_plain)
    ref = _torch_block_stats(y)
    got = st.buf.double().sum(1)
    assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    y2, st2 = ops.conv2d(x0, pw, want_stats=True, **kw)
    assert torch.equal(st.buf, st2.buf) and torch.equal(y, y2)
    # the statistics kernel on the same tensor: same sums up to fp32 order
    st3 = ops.block_stats(y)
    assert (st3.buf.double().sum(1) - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # batch independence: image 0 alone gives bitwise the same statistics rows
    y0, st0 = ops.conv2d(x0[:1].contiguous(), pw, want_stats=True, in1=None if x1 is None else x1[:1].contiguous(), bias=bias,
                         addvec=temb[:1].contiguous(), residual=None if r is None else r[:1].contiguous(), upsample=ups)
    assert torch.equal(st0.buf[0], st.buf[0])


def test_conv_stats_unsupported_kernel_is_loud(ops):
    """Shapes whose kernel writes no statistics report P = 0 (ops.conv2d then returns None for them), and forcing the
    descriptor field anyway is an error, not a silent no-op."""
    import ctypes
    from dxmi_hip import _lib
    x = torch.randn(2, 4, 4, 256).to(torch.bfloat16).to(DEV)
    pw = ops.pack_conv_weight((torch.randn(256, 256, 3, 3) * 0.03).to(DEV))
    y, st = ops.conv2d(x, pw, want_stats=True)
    assert st is None
    d = _lib.ConvDesc()
    out = torch.empty_like(y)
    junk = torch.empty(2 * 64 * 2, device=DEV)
    d.in0, d.wpacked, d.out, d.gn_stats = x.data_ptr(), pw.buf.data_ptr(), out.data_ptr(), junk.data_ptr()
    d.N, d.IH, d.IW, d.C0, d.C1, d.OH, d.OW, d.Cout = 2, 4, 4, 256, 0, 4, 4, 256
    d.ksize, d.stride, d.pad = 3, 1, 1
    assert _lib.load().dxmi_conv2d_fwd(ctypes.byref(d), None) == -1
    assert b"statistics" in _lib.load().dxmi_last_error()


GN_APPLY_CASES = [
    # N, C0, C1, H, silu
    (3, 128, 0, 32, True),
    (2, 256, 0, 16, False),
    (2, 256, 128, 32, True),     # 12 channels per group, groups straddle the concat boundary
    (2, 256, 128, 16, True),
    (2, 256, 256, 16, True),
    (3, 128, 128, 32, True),
    (2, 256, 0, 8, True),        # one chunk, partial trip
    (1, 512, 0, 64, True),       # 4096 rows: 64 chunks per image
    (2, 64, 0, 5, True),         # HW = 25: ragged rows
    (2, 192, 0, 64, True),       # ImageNet-64 net: 6 channels per group
    (2, 384, 192, 32, True),     # 18 channels per group across a concat
]


@pytest.mark.parametrize("N,C0,C1,H,silu", GN_APPLY_CASES)
def test_groupnorm_apply_vs_torch_and_resident(ops, N, C0, C1, H, silu):
    C = C0 + C1
    g = torch.Generator().manual_seed(31 + C + H)
    x = bf(torch.randn(N, C, H, H, generator=g) * 2.0 + 0.5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    x0 = nhwc(x[:, :C0])
    x1 = nhwc(x[:, C0:]) if C1 else None
    s0 = ops.block_stats(x0)
    s1 = ops.block_stats(x1) if C1 else None
    assert s0.P == min(ops.load().dxmi_gn_block_stats_partials(H * H), ops.MAX_APPLY_PARTIALS) or s0.P == 1      # (folded when > 8 partials)
    ga, be = gamma.to(DEV), beta.to(DEV)
    y = ops.groupnorm_silu(x0, ga, be, in1=x1, eps=1e-6, silu=silu, stats=(s0, s1))
    got = nchw(y)
    assert (got - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert ((got - ref).norm() / ref.norm()).item() < 4e-3
    # against the exact result rounded once to bf16: all but a handful of values identical
    same = (y.float().cpu() == ref.permute(0, 2, 3, 1).to(torch.bfloat16).float()).float().mean().item()
    assert same > 0.999, same
    if ops.load().dxmi_groupnorm_silu_supported(C0, C1, H * H, 32):
        yr = ops.groupnorm_silu(x0, ga, be, in1=x1, eps=1e-6, silu=silu)
        assert (y == yr).float().mean().item() > 0.999
    y2 = ops.groupnorm_silu(x0, ga, be, in1=x1, eps=1e-6, silu=silu, stats=(s0, s1))
    assert torch.equal(y, y2)                                   # bitwise reproducible
    # batch independence
    s00 = ops.BlockStats(s0.buf[:1].contiguous(), s0.P)
    s10 = ops.BlockStats(s1.buf[:1].contiguous(), s1.P) if C1 else None
    y0 = ops.groupnorm_silu(x0[:1].contiguous(), ga, be, in1=None if x1 is None else x1[:1].contiguous(), eps=1e-6, silu=silu, stats=(s00, s10))
    assert torch.equal(y0[0], y[0])


def test_unet_forward_streaming_groupnorm_vs_one_pass():
    """The whole DDPM U-Net forward with the streaming GroupNorm path (default) and with the one-pass kernel everywhere:
    same function up to the bf16 noise floor of a re-ordered fp32 variance (a flipped rounding perturbs ~1e3
    downstream values), the default path bitwise reproducible and batch independent."""
    from models.DxMI.unet_small import Model
    from oracle.weights import formula_tensor
    net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32)
    net.load_state_dict({k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()})
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 3, 32, 32, generator=g).to(DEV)
    t = torch.tensor([0.5, 30.0, 200.0, 640.0, 999.0], device=DEV)
    with torch.no_grad():
        y = net(x, t).clone()
        y_again = net(x, t)
        y_one = net(x[2:3].contiguous(), t[2:3].contiguous())
        net.STREAM_GN_MIN_HW = 1 << 30
        y_res = net(x, t)
    assert torch.isfinite(y).all() and torch.equal(y, y_again)
    assert torch.equal(y_one[0], y[2])
    rel = ((y - y_res).norm() / y_res.norm()).item()
    assert rel < 1.2e-2, rel


# ------------------------------------------------------------------------------------------------ conv_sm_kernel (4x4 maps)
def _conv_ref(x0, x1, w, bias, temb, r, act):
    xin = torch.cat([x0, x1], 3) if x1 is not None else x0
    ref = F.conv2d(xin.float().permute(0, 3, 1, 2).cpu(), w.to(torch.bfloat16).float().cpu(), bias.cpu() if bias is not None else None, padding=1)
    if temb is not None:
        ref = ref + temb.cpu()[:, :, None, None]
    if r is not None:
        ref = ref + r.float().cpu().permute(0, 3, 1, 2)
    if act == 1:
        ref = F.leaky_relu(ref, 0.2)
    elif act == 2:
        ref = F.relu(ref)
    return ref.permute(0, 2, 3, 1)


SM_CASES = [
    # N, C0, C1, Cout, residual, temb, bias, act
    (256, 256, 0, 256, True, True, True, 0),       # one tile per CU
    (37, 256, 0, 256, True, True, True, 0),        # ragged last 8-image tile
    (3, 256, 256, 256, False, True, True, 0),      # concat, 16 chunks, fewer images than one tile
    (70, 160, 96, 64, True, False, False, 1),      # odd concat split, two cout tiles, leaky, no bias / temb
    (300, 256, 0, 256, True, True, True, 2),       # 304 tiles on 256 workgroups: the stream runs across tile ends
]


@pytest.mark.parametrize("N,C0,C1,Cout,res,temb,bias,act", SM_CASES)
def test_conv_sm_4x4(ops, N, C0, C1, Cout, res, temb, bias, act):
    import ctypes
    g = torch.Generator().manual_seed(5 + N + Cout)
    x0 = torch.randn(N, 4, 4, C0, generator=g).to(torch.bfloat16).to(DEV)
    x1 = torch.randn(N, 4, 4, C1, generator=g).to(torch.bfloat16).to(DEV) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.03).to(DEV)
    pw = ops.pack_conv_weight(w)
    bv = torch.randn(Cout, generator=g).to(DEV) if bias else None
    tv = torch.randn(N, Cout, generator=g).to(DEV) if temb else None
    r = torch.randn(N, 4, 4, Cout, generator=g).to(torch.bfloat16).to(DEV) if res else None
    kw = dict(in1=x1, bias=bv, addvec=tv, residual=r, act=act)
    y = ops.conv2d(x0, pw, **kw)
    ref = _conv_ref(x0, x1, w, bv, tv, r, act)
    rel = ((y.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 4e-3, rel
    # it IS the small-map kernel
    prof = ops.OpProfiler()
    ops.PROFILER = prof
    try:
        ops.conv2d(x0, pw, **kw)
    finally:
        ops.PROFILER = None
    torch.cuda.synchronize()
    assert [k[1] for k in prof.summary()] == [450432]          # conv_sm_kernel<2, 8, 32>
    # bitwise reproducible and independent of the batch an image rides in (first, middle, last image alone)
    assert torch.equal(y, ops.conv2d(x0, pw, **kw))
    for i in {0, N // 2, N - 1}:
        one = ops.conv2d(x0[i:i + 1].contiguous(), pw, in1=None if x1 is None else x1[i:i + 1].contiguous(), bias=bv,
                         addvec=None if tv is None else tv[i:i + 1].contiguous(), residual=None if r is None else r[i:i + 1].contiguous(), act=act)
        assert torch.equal(one[0], y[i]), i
    # a NaN input pixel poisons its 3x3 neighbourhood of every cout and nothing else
    xn = x0.clone()
    xn[N - 1, 1, 2, 7] = float("nan")
    yn = ops.conv2d(xn, pw, **kw)
    bad = torch.isnan(yn.float())
    assert bad[N - 1, 0:3, 1:4].all() and not bad[N - 1, 3].any() and not bad[N - 1, :, 0].any() and not bad[:N - 1].any()


def test_groupnorm_apply_scale_shift_and_fold(ops):
    """ADM scale-shift norm (models/cm/unet.py:252-256: GroupNorm32 -> * (1 + scale) + shift -> SiLU) on the streaming path, with
    a 64x64 map whose 16 row-chunk partials are folded first; against fp32 torch on the bf16-rounded input and against the
    generic two-kernel path."""
    N, C, H = 3, 192, 64
    g = torch.Generator().manual_seed(77)
    x = bf(torch.randn(N, C, H, H, generator=g) * 1.5 - 0.25)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ss = torch.randn(N, 2 * C, generator=g) * 0.5
    ref = F.group_norm(x, 32, gamma, beta, 1e-5) * (1 + ss[:, :C, None, None]) + ss[:, C:, None, None]
    ref = F.silu(ref)
    x0 = nhwc(x)
    st = ops.block_stats(x0)
    assert st.P == 1                                            # 16 partials -> folded
    y = ops.groupnorm_silu(x0, gamma.to(DEV), beta.to(DEV), eps=1e-5, silu=True, scale_shift=ss.to(DEV), stats=(st, None))
    got = nchw(y)
    assert ((got - ref).norm() / ref.norm()).item() < 4e-3
    yg = ops.groupnorm_silu(x0, gamma.to(DEV), beta.to(DEV), eps=1e-5, silu=True, scale_shift=ss.to(DEV))     # generic path
    assert (y == yg).float().mean().item() > 0.999
    # fold of a conv's many partials equals their plain sum
    raw = torch.randn(2, 512, 64, 2, generator=g).to(DEV)
    f = ops.fold_stats(ops.BlockStats(raw, 512))
    assert f.P == 1 and torch.allclose(f.buf[:, 0], raw.sum(1), rtol=1e-5, atol=1e-4)
    assert torch.equal(f.buf, ops.fold_stats(ops.BlockStats(raw, 512)).buf)


# -------------------------------------------------------------------------- GroupNorm of the conv's output from its own epilogue
@pytest.mark.parametrize("N,C0,C1,res,silu", [(37, 256, 0, False, True), (8, 256, 256, True, False), (300, 256, 0, True, True)])
def test_conv_sm_fused_groupnorm(ops, N, C0, C1, res, silu):
    """dxmi_conv_desc.gn_out: GroupNorm(+SiLU) of a 4x4 conv's output from conv_sm_kernel's epilogue (Normalize() + nonlinearity
    of the next layer, reference unet_small.py:119-126) against the separate GroupNorm launch and torch fp32."""
    Cout = 256
    g = torch.Generator().manual_seed(11 + N)
    x0 = torch.randn(N, 4, 4, C0, generator=g).to(torch.bfloat16).to(DEV)
    x1 = torch.randn(N, 4, 4, C1, generator=g).to(torch.bfloat16).to(DEV) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.03).to(DEV)
    pw = ops.pack_conv_weight(w)
    bv = torch.randn(Cout, generator=g).to(DEV)
    tv = torch.randn(N, Cout, generator=g).to(DEV)
    r = torch.randn(N, 4, 4, Cout, generator=g).to(torch.bfloat16).to(DEV) if res else None
    gamma = (1 + 0.3 * torch.randn(Cout, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(Cout, generator=g)).to(DEV)
    kw = dict(in1=x1, bias=bv, addvec=tv, residual=r)
    raw = ops.conv2d(x0, pw, **kw)
    out, y = ops.conv2d(x0, pw, fuse_gn=(gamma, beta, 32, 1e-6, silu, True), **kw)
    assert y is not None and torch.equal(out, raw)                  # the raw tensor is unchanged by the fusion
    none, y2 = ops.conv2d(x0, pw, fuse_gn=(gamma, beta, 32, 1e-6, silu, False), **kw)
    assert none is None and torch.equal(y, y2)
    # against the separate launch on the same stored tensor: same formula, statistics summed in another order -> bf16 ulps
    sep = ops.groupnorm_silu(raw, gamma, beta, groups=32, eps=1e-6, silu=silu)
    d = (y.float() - sep.float()).abs()
    assert d.max().item() <= 2 ** -6 * max(1.0, sep.float().abs().max().item()) and (d > 0).float().mean().item() < 0.02
    # against torch fp32 on the stored tensor
    ref = F.group_norm(raw.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    assert ((y.float() - ref).norm() / ref.norm()).item() < 3e-3
    # bitwise independent of the batch
    for i in {0, N // 2, N - 1}:
        _, one = ops.conv2d(x0[i:i + 1].contiguous(), pw, in1=None if x1 is None else x1[i:i + 1].contiguous(), bias=bv,
                            addvec=tv[i:i + 1].contiguous(), residual=None if r is None else r[i:i + 1].contiguous(),
                            fuse_gn=(gamma, beta, 32, 1e-6, silu, False))
        assert torch.equal(one[0], y[i]), i


def test_conv_fused_groupnorm_unsupported_shape(ops):
    """8x8 maps with the raw output kept (conv_ws8 has one output tile): ops returns y None; the raw C-ABI call is an error."""
    import ctypes
    from dxmi_hip import _lib
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 8, 8, 256, generator=g).to(torch.bfloat16).to(DEV)
    pw = ops.pack_conv_weight((torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(DEV))
    gamma, beta = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    out, y = ops.conv2d(x, pw, fuse_gn=(gamma, beta, 32, 1e-6, True, True))      # 8x8: only INSTEAD of the raw output
    assert y is None and out is not None
    d = _lib.ConvDesc()
    yb = torch.empty_like(out)
    d.in0, d.wpacked, d.out = x.data_ptr(), pw.buf.data_ptr(), out.data_ptr()
    d.N, d.IH, d.IW, d.C0, d.C1, d.OH, d.OW, d.Cout = 4, 8, 8, 256, 0, 8, 8, 256
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.gn_out, d.gn_gamma, d.gn_beta, d.gn_eps, d.gn_groups, d.gn_flags = yb.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-6, 32, 1
    lib = _lib.load()
    assert lib.dxmi_conv2d_gn_fuse_supported(ctypes.byref(d)) == 0
    assert lib.dxmi_conv2d_fwd(ctypes.byref(d), None) != 0
    assert b"GroupNorm" in lib.dxmi_last_error()
    d.gn_groups = 16          # 16 channels per group: not the fused layout either
    d.IH = d.IW = d.OH = d.OW = 4
    assert lib.dxmi_conv2d_gn_fuse_supported(ctypes.byref(d)) == 0
    d.gn_groups = 32
    assert lib.dxmi_conv2d_gn_fuse_supported(ctypes.byref(d)) == 1


# -------------------------------------------------------------------------- proj_out + residual fused behind the attention
@pytest.mark.parametrize("N", [1, 5])
def test_attention_proj_fused(ops, N):
    """dxmi_attention_proj_fwd (AttnBlock tail, reference unet_small.py:175-190) against the two launches it replaces and
    against torch fp32."""
    g = torch.Generator().manual_seed(21 + N)
    T = C = 256
    qkv = (torch.randn(N, T, 3 * C, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    x = torch.randn(N, T, C, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(C, C, 1, 1, generator=g) * 0.06).to(DEV)
    b = torch.randn(C, generator=g).to(DEV)
    scale = float(C) ** -0.5
    assert ops.attention_proj_supported(T, C, 1) and not ops.attention_proj_supported(16, C, 1)
    y = ops.attention_proj(qkv, ops.pack_attn_proj_weight(w), b, x, heads=1, scale=scale)
    # the unfused pair
    a = ops.attention(qkv, heads=1, scale=scale)
    sep = ops.conv2d(a.view(N, 16, 16, C), ops.pack_conv_weight(w), bias=b, residual=x.view(N, 16, 16, C)).view(N, T, C)
    d = (y.float() - sep.float()).abs()
    assert d.max().item() <= 2 ** -6 * max(1.0, sep.float().abs().max().item()), d.max().item()
    assert (d > 0).float().mean().item() < 0.02          # same operands; only the order inside a 16-channel MFMA step differs
    # torch fp32 on the same bf16 inputs
    q, k, v = (t.float() for t in qkv.split(C, dim=2))
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * scale, dim=2)
    o = torch.bmm(att, v)
    ref = x.float() + o @ w.view(C, C).t() + b
    assert ((y.float() - ref).norm() / ref.norm()).item() < 4e-3
    # block statistics of the stored output from the same launch (and the output unchanged by asking for them)
    y2, st = ops.attention_proj(qkv, ops.pack_attn_proj_weight(w), b, x, heads=1, scale=scale, want_stats=True)
    assert torch.equal(y, y2) and st.P == 8 and tuple(st.buf.shape) == (N, 8, C // 2, 2)
    want = _torch_block_stats(y.view(N, 16, 16, C)).to(DEV)
    got = st.buf.double().sum(1)
    assert ((got - want).abs() <= 1e-4 * (1 + want.abs())).all()
    want8 = torch.stack([_torch_block_stats(y[:, 32 * k:32 * k + 32].reshape(N, 2, 16, C)) for k in range(8)], 1).to(DEV)
    assert ((st.buf.double() - want8).abs() <= 1e-4 * (1 + want8.abs())).all()
    # bitwise reproducible and independent of the batch
    assert torch.equal(y, ops.attention_proj(qkv, ops.pack_attn_proj_weight(w), b, x, heads=1, scale=scale))
    i = N - 1
    one = ops.attention_proj(qkv[i:i + 1].contiguous(), ops.pack_attn_proj_weight(w), b, x[i:i + 1].contiguous(), heads=1, scale=scale)
    assert torch.equal(one[0], y[i])


@pytest.mark.parametrize("N,C0,C1,res,silu", [(37, 256, 0, False, True), (6, 256, 256, True, False), (256, 256, 0, False, True)])
def test_conv_ws8_fused_groupnorm(ops, N, C0, C1, res, silu):
    """8x8 maps: conv_ws8_kernel<true> writes GroupNorm(+SiLU) of its output instead of the raw output."""
    Cout = 256
    g = torch.Generator().manual_seed(17 + N)
    x0 = torch.randn(N, 8, 8, C0, generator=g).to(torch.bfloat16).to(DEV)
    x1 = torch.randn(N, 8, 8, C1, generator=g).to(torch.bfloat16).to(DEV) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) * 0.03).to(DEV)
    pw = ops.pack_conv_weight(w)
    bv = torch.randn(Cout, generator=g).to(DEV)
    tv = torch.randn(N, Cout, generator=g).to(DEV)
    r = torch.randn(N, 8, 8, Cout, generator=g).to(torch.bfloat16).to(DEV) if res else None
    gamma = (1 + 0.3 * torch.randn(Cout, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(Cout, generator=g)).to(DEV)
    kw = dict(in1=x1, bias=bv, addvec=tv, residual=r)
    raw = ops.conv2d(x0, pw, **kw)
    none, y = ops.conv2d(x0, pw, fuse_gn=(gamma, beta, 32, 1e-6, silu, False), **kw)
    assert none is None and y is not None
    sep = ops.groupnorm_silu(raw, gamma, beta, groups=32, eps=1e-6, silu=silu)
    d = (y.float() - sep.float()).abs()
    assert d.max().item() <= 2 ** -6 * max(1.0, sep.float().abs().max().item()) and (d > 0).float().mean().item() < 0.02
    ref = F.group_norm(raw.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    assert ((y.float() - ref.permute(0, 2, 3, 1)).norm() / ref.norm()).item() < 3e-3
    for i in {0, N // 2, N - 1}:
        _, one = ops.conv2d(x0[i:i + 1].contiguous(), pw, in1=None if x1 is None else x1[i:i + 1].contiguous(), bias=bv,
                            addvec=tv[i:i + 1].contiguous(), residual=None if r is None else r[i:i + 1].contiguous(),
                            fuse_gn=(gamma, beta, 32, 1e-6, silu, False))
        assert torch.equal(one[0], y[i]), i


def test_stem_and_stride2_conv_block_stats(ops):
    """conv_stem_kernel and the stride-2 conv_pipe_kernel (Downsample) emit the block statistics of what they store (one partial
    per tile), the output is bitwise that of a launch without statistics, and an image's rows do not depend on the batch."""
    g = torch.Generator().manual_seed(77)
    N = 5
    # stem: NCHW fp32 image -> NHWC bf16 [N,32,32,128]
    x = torch.randn(N, 3, 32, 32, generator=g).to(DEV)
    pw = ops.pack_conv_weight((torch.randn(128, 3, 3, 3, generator=g) * 0.2).to(DEV), k27=True)
    b = torch.randn(128, generator=g).to(DEV)
    y, st = ops.conv2d(x, pw, bias=b, want_stats=True)
    assert st is not None and st.P == 8 and torch.equal(y, ops.conv2d(x, pw, bias=b))
    ref = _torch_block_stats(y)
    assert (st.buf.double().sum(1) - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # a partial is one 4-row band of the image
    band = _torch_block_stats(y[:, 8:12])
    assert (st.buf[:, 2].double() - band).abs().max().item() <= 2e-6 * ref.abs().max().item()
    y1, st1 = ops.conv2d(x[2:3].contiguous(), pw, bias=b, want_stats=True)
    assert torch.equal(st1.buf[0], st.buf[2]) and torch.equal(y1[0], y[2])
    # Downsample: k3 s2, pad (0, 1): 32x32 -> 16x16, 128 channels
    x2 = torch.randn(N, 32, 32, 128, generator=g).to(torch.bfloat16).to(DEV)
    pw2 = ops.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) * 0.03).to(DEV))
    kw = dict(bias=b, stride=2, pad=0, pad_br=1)
    y2, st2 = ops.conv2d(x2, pw2, want_stats=True, **kw)
    assert st2 is not None and st2.P == 4 and tuple(y2.shape) == (N, 16, 16, 128) and torch.equal(y2, ops.conv2d(x2, pw2, **kw))
    ref2 = _torch_block_stats(y2)
    assert (st2.buf.double().sum(1) - ref2).abs().max().item() <= 2e-6 * ref2.abs().max().item()
    y3, st3 = ops.conv2d(x2[4:5].contiguous(), pw2, want_stats=True, **kw)
    assert torch.equal(st3.buf[0], st2.buf[4])
    # 8x8 output (a 64-pixel tile would hold one image, the kernel picked packs several): no statistics, None
    x4 = torch.randn(N, 16, 16, 128, generator=g).to(torch.bfloat16).to(DEV)
    y4, st4 = ops.conv2d(x4, pw2, want_stats=True, **kw)
    assert st4 is None or (st4.buf.double().sum(1) - _torch_block_stats(y4)).abs().max().item() <= 1e-3


def test_multi_tensor_pack_equals_per_weight_pack(ops):
    """dxmi_pack_conv_weights (one thread per fragment lane, all taps; contiguous-run fast path and guarded generic path) writes
    bitwise the images of dxmi_pack_conv_weight for 3x3 / 1x1 / stem / transposed-flipped / ragged shapes."""
    g = torch.Generator().manual_seed(3)
    cases = [((128, 128, 3, 3), False, False), ((256, 384, 3, 3), False, False), ((192, 192, 3, 3), False, False),
             ((768, 256, 1, 1), False, False), ((128, 3, 3, 3), False, True), ((256, 128, 3, 3), True, False),
             ((96, 72, 3, 3), False, False), ((40, 24, 1, 1), False, False), ((3, 128, 3, 3), False, False), ((128, 256, 1, 1), True, False)]
    ws = [(torch.randn(*shape, generator=g).to(DEV), tf, k27) for shape, tf, k27 in cases]
    single = [ops.pack_conv_weight(w, transpose_flip=tf, k27=k27) for w, tf, k27 in ws]
    with ops.pack_batch():
        batched = [ops.pack_conv_weight(w, transpose_flip=tf, k27=k27) for w, tf, k27 in ws]
    torch.cuda.synchronize()
    for (shape, tf, k27), a, b in zip(cases, single, batched):
        assert (a.Cout, a.Cin, a.ksize) == (b.Cout, b.Cin, b.ksize)
        assert torch.equal(a.buf, b.buf), (shape, tf, k27)


def test_colsum_f32(ops):
    g = torch.Generator().manual_seed(4)
    for B, N, C in ((2, 256, 128), (2, 7, 384), (1, 100, 40), (3, 16, 2304)):
        x = torch.randn(B, N, C, generator=g).to(DEV)
        got = ops.colsum_f32(x)
        assert tuple(got.shape) == (B, C)
        assert (got.double() - x.double().sum(1)).abs().max().item() <= 1e-5 * N ** 0.5 * 4
        assert torch.equal(got, ops.colsum_f32(x))
    x2 = torch.randn(33, 64, generator=g).to(DEV)
    assert (ops.colsum_f32(x2).double() - x2.double().sum(0)).abs().max().item() <= 1e-4


@pytest.mark.parametrize("N,C,Cout,H", [(5, 128, 128, 32), (3, 256, 128, 16), (260, 128, 128, 16), (37, 256, 256, 8), (19, 256, 256, 4)])
def test_conv_ws_activation_mask(ops, N, C, Cout, H):
    """Data gradient through a LeakyReLU on conv_ws_kernel: the mask source rides the residual tile's path
    (out = conv(x) * (mask > 0 ? 1 : slope); value net backward, models/modules.py:96-101)."""
    g = torch.Generator().manual_seed(41 + N)
    x = torch.randn(N, H, H, C, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.03).to(DEV)
    pw = ops.pack_conv_weight(w)
    m = torch.randn(N, H, H, Cout, generator=g).to(torch.bfloat16).to(DEV)
    prof = ops.OpProfiler()
    ops.PROFILER = prof
    try:
        y = ops.conv2d(x, pw, mask_src=m, mask_slope=0.2)
    finally:
        ops.PROFILER = None
    torch.cuda.synchronize()
    assert [k[1] for k in prof.summary()] == [{32: 400032, 16: 400016, 8: 400008, 4: 450432}[H]]      # conv_ws / conv_ws8 / conv_sm
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1)
    ref = (ref * torch.where(m.float().permute(0, 3, 1, 2) > 0, 1.0, 0.2)).permute(0, 2, 3, 1)
    assert ((y.float() - ref).norm() / ref.norm()).item() < 4e-3
    assert torch.equal(y, ops.conv2d(x, pw, mask_src=m, mask_slope=0.2))
    i = N - 1
    assert torch.equal(ops.conv2d(x[i:i + 1].contiguous(), pw, mask_src=m[i:i + 1].contiguous(), mask_slope=0.2)[0], y[i])
    # mask AND residual together stay on the pipelined kernel
    r = torch.randn(N, H, H, Cout, generator=g).to(torch.bfloat16).to(DEV)
    y2 = ops.conv2d(x, pw, mask_src=m, mask_slope=0.2, residual=r)
    ref2 = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1) + r.float().permute(0, 3, 1, 2)
    ref2 = (ref2 * torch.where(m.float().permute(0, 3, 1, 2) > 0, 1.0, 0.2)).permute(0, 2, 3, 1)
    assert ((y2.float() - ref2).norm() / ref2.norm()).item() < 4e-3
