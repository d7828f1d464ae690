"""Round-5 kernels through the C-ABI: the 16x16 AttnBlock as one launch (dxmi_attn_block_fwd: GroupNorm on load, folded
Wk^T Wq / Wproj Wv, the raw input tile as K, V and residual) against the reference block in torch fp32
(/root/reference models/DxMI/unet_small.py:167-191) and against the three launches it replaces."""
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


def _torch_block_stats(y):
    """y [N,H,W,C] bf16 -> fp64 [N, C/2, 2]: (sum, sum of squares) per channel pair over the image."""
    v = y.double().reshape(y.shape[0], -1, y.shape[-1] // 2, 2)
    return torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)


def _attn_block_params(g, C=256, wscale=0.06):
    p = {}
    for name in ("q", "k", "v", "proj"):
        p[name + "_w"] = (torch.randn(C, C, 1, 1, generator=g) * wscale).to(DEV)
        p[name + "_b"] = (torch.randn(C, generator=g) * 0.3).to(DEV)
    p["gamma"] = (1.0 + 0.3 * torch.randn(C, generator=g)).to(DEV)
    p["beta"] = (0.2 * torch.randn(C, generator=g)).to(DEV)
    return p


def _reference_block(x, p, eps=1e-6):
    """AttnBlock.forward in fp32 on the bf16 input (token-major [N,T,C])."""
    N, T, C = x.shape
    xf = x.float()
    hn = F.group_norm(xf.transpose(1, 2).reshape(N, C, 16, 16), 32, p["gamma"], p["beta"], eps).reshape(N, C, T).transpose(1, 2)
    q, k, v = (hn @ p[n + "_w"].view(C, C).t() + p[n + "_b"] for n in ("q", "k", "v"))
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * float(C) ** -0.5, dim=2)
    return xf + torch.bmm(att, v) @ p["proj_w"].view(C, C).t() + p["proj_b"]


def _three_launches(ops, x4, st, p):
    N, H, W, C = x4.shape
    hn = ops.groupnorm_silu(x4, p["gamma"], p["beta"], eps=1e-6, silu=False, stats=(st, None))
    qkv_w = ops.pack_conv_weight(torch.cat([p["q_w"], p["k_w"], p["v_w"]], 0))
    qkv = ops.conv2d(hn, qkv_w, bias=torch.cat([p["q_b"], p["k_b"], p["v_b"]], 0).contiguous())
    return ops.attention_proj(qkv.view(N, H * W, 3 * C), ops.pack_attn_proj_weight(p["proj_w"]), p["proj_b"], x4.view(N, H * W, C),
                              heads=1, scale=float(C) ** -0.5).view(N, H, W, C)


@pytest.mark.parametrize("N,wscale", [(1, 0.06), (5, 0.06), (3, 0.15)])
def test_attn_block_one_launch(ops, N, wscale):
    g = torch.Generator().manual_seed(500 + N)
    T = C = 256
    assert ops.attn_block_supported(T, C, 1) and not ops.attn_block_supported(64, C, 1) and not ops.attn_block_supported(T, 512, 1)
    # per-channel offsets and scales so that the GroupNorm has something to do
    x = (torch.randn(N, T, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + 0.7 * torch.randn(C, generator=g)).to(torch.bfloat16).to(DEV)
    x4 = x.view(N, 16, 16, C)
    p = _attn_block_params(g, C, wscale)
    st = ops.block_stats(x4)
    packed = ops.attn_block_pack(p["q_w"], p["q_b"], p["k_w"], p["v_w"], p["v_b"], p["proj_w"], p["proj_b"], float(C) ** -0.5)
    y, yst = ops.attn_block(x4, st, p["gamma"], p["beta"], packed, want_stats=True)
    torch.cuda.synchronize()
    assert y.shape == x4.shape and torch.isfinite(y.float()).all()
    ref = _reference_block(x, p).view(N, 16, 16, C)
    sep = _three_launches(ops, x4, st, p)
    branch = (ref - x4.float()).norm().item()          # the attention branch alone (the residual x dominates the output)
    err_one = (y.float() - ref).norm().item() / branch
    err_sep = (sep.float() - ref).norm().item() / branch
    print(f"attn_block N={N} wscale={wscale}: one launch {err_one:.3e}, three launches {err_sep:.3e} of the branch norm")
    # the output is rounded to bf16 once: ~2^-9 of |x + branch| per element is the floor of both forms
    floor = (2.0 ** -9) * ref.norm().item() / branch
    assert err_one < 1.5 * floor + 1e-2, (err_one, floor)
    assert err_one < 1.25 * err_sep + 2e-3, (err_one, err_sep)
    # block statistics of the stored output from the same launch
    assert yst.P == 8 and tuple(yst.buf.shape) == (N, 8, C // 2, 2)
    want = _torch_block_stats(y).to(DEV)
    assert ((yst.buf.double().sum(1) - want).abs() <= 1e-4 * (1 + want.abs())).all()
    want8 = torch.stack([_torch_block_stats(y.view(N, T, C)[:, 32 * k:32 * k + 32].reshape(N, 2, 16, C)) for k in range(8)], 1).to(DEV)
    assert ((yst.buf.double() - want8).abs() <= 1e-4 * (1 + want8.abs())).all()
    # the output does not change when no statistics are asked for; bitwise reproducible; independent of the batch it rides in
    assert torch.equal(y, ops.attn_block(x4, st, p["gamma"], p["beta"], packed))
    i = N - 1
    st1 = ops.block_stats(x4[i:i + 1].contiguous())
    one = ops.attn_block(x4[i:i + 1].contiguous(), st1, p["gamma"], p["beta"], packed)
    assert torch.equal(one[0], y[i])


def test_attn_block_key_bias_is_irrelevant_and_partials_fold(ops):
    """The k bias drops out of the softmax (constant along the key axis): the packed block ignores it by construction; check the
    reference agrees.  Statistics with more than 8 partials per image are folded first (the C-ABI takes at most 8)."""
    g = torch.Generator().manual_seed(77)
    N, T, C = 2, 256, 256
    x = torch.randn(N, T, C, generator=g).to(torch.bfloat16).to(DEV)
    p = _attn_block_params(g, C)
    r0 = _reference_block(x, p)
    p2 = dict(p)
    p2["k_b"] = p["k_b"] + 5.0
    assert ((r0 - _reference_block(x, p2)).abs().max() < 1e-3)
    x4 = x.view(N, 16, 16, C)
    st = ops.block_stats(x4)
    packed = ops.attn_block_pack(p["q_w"], p["q_b"], p["k_w"], p["v_w"], p["v_b"], p["proj_w"], p["proj_b"], float(C) ** -0.5)
    y = ops.attn_block(x4, st, p["gamma"], p["beta"], packed)
    # 16 partials per image (one per row of the map) carrying the same totals
    v = x4.float().reshape(N, 16, 16, C // 2, 2)
    buf = torch.stack([v.sum((2, 4)), (v * v).sum((2, 4))], -1).contiguous()
    y16 = ops.attn_block(x4, ops.BlockStats(buf, 16), p["gamma"], p["beta"], packed)
    assert ((y.float() - y16.float()).abs().max() <= 2 ** -6 * y.float().abs().max())
    assert ((y.float() - r0.view(N, 16, 16, C)).norm() / (r0 - x.float()).norm()).item() < 5e-2


def test_unet_forward_with_and_without_the_one_launch_attn_block():
    """The DDPM U-Net forward with the five 16x16 AttnBlocks as one launch each against the three-launch path (same weights, same
    input): both are bf16 pipelines of the same network, so they agree to the bf16 noise floor of a forward."""
    from models.DxMI.unet_small import Model
    from oracle.weights import formula_tensor
    torch.manual_seed(0)
    net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32)
    sd = {k: formula_tensor(k, v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 3, 32, 32, generator=g).to(DEV)
    t = torch.tensor([616.7, 1.5e-4, 300.0, 12.0, 900.0, 55.5], device=DEV)
    with torch.no_grad():
        net.FUSE_ATTN_BLOCK = True
        tr1 = []
        y1 = net.forward_inference(x, t, trace=tr1)
        net.FUSE_ATTN_BLOCK = False
        tr0 = []
        y0 = net.forward_inference(x, t, trace=tr0)
    worst = max(((a.float() - b.float()).norm() / b.float().norm()).item() for (_, a), (_, b) in zip(tr1, tr0) if a.dtype == torch.bfloat16)
    rel = ((y1 - y0).norm() / y0.norm()).item()
    print(f"unet forward one-launch vs three-launch attn blocks: rel-L2 {rel:.3e}, worst block {worst:.3e}")
    assert torch.isfinite(y1).all() and rel < 1.2e-2 and worst < 2e-2



def test_conv_ws_in_kernel_clock(ops):
    """dxmi_conv_ws_last_clock: the shader clock the chip held during the last conv_ws_kernel launch, from the kernel's own
    s_memtime / s_memrealtime stamps (bench.py roofline.clock_ghz).  MI355X boosts to 2.4 GHz and throttles under MFMA load."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 32, 32, 128, generator=g).to(torch.bfloat16).to(DEV)
    pw = ops.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) * 0.03).to(DEV))
    for _ in range(3):
        ops.conv2d(x, pw)
    torch.cuda.synchronize()
    c = ops.conv_ws_clock_ghz()
    print(f"conv_ws_kernel in-kernel clock: {c:.3f} GHz")
    assert c is not None and 0.8 < c < 2.6, c


def test_value_net_forward_pair_equals_two_forwards():
    """IGEBMEncoderV2.forward_pair (round 5: TD target + TD prediction of a train step in ONE forward, trainer.py:288-300): the two
    halves equal the no-grad forward and the autograd forward on their own, and the parameter gradients equal the gradients of
    the prediction half alone."""
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    torch.manual_seed(3)
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, keepdim=False, nh=128)).to(DEV)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(6, 3, 32, 32, generator=g).to(DEV)
    x1 = torch.randn(6, 3, 32, 32, generator=g).to(DEV)
    t = torch.zeros(6, dtype=torch.long, device=DEV)
    tgt, pred = v.forward_pair(x0, t + 1, x1, t)
    assert not tgt.requires_grad and pred.requires_grad and tgt.shape == pred.shape == (6, 1)
    (pred.squeeze() - tgt.squeeze()).pow(2).mean().backward()
    grads = [p.grad.clone() for p in v.parameters()]
    for p in v.parameters():
        p.grad = None
    with torch.no_grad():
        tgt2 = v(x0, t + 1)
    pred2 = v(x1, t)
    (pred2.squeeze() - tgt2.squeeze()).pow(2).mean().backward()
    assert torch.equal(tgt, tgt2) and torch.equal(pred.detach(), pred2.detach())
    for a, p in zip(grads, v.parameters()):
        assert torch.equal(a, p.grad), (a - p.grad).abs().max()
    # an input that asks for a gradient (the policy step) keeps the two-forward path
    x1g = x1.clone().requires_grad_(True)
    tgt3, pred3 = v.forward_pair(x0, t + 1, x1g, t)
    pred3.sum().backward()
    assert x1g.grad is not None and torch.equal(tgt3, tgt2)


# -------------------------------------------------------------------------- 1x1 convs on conv1x1_rw8_kernel (K = 384 / 512 / 576)
@pytest.mark.parametrize("N,H,C0,C1,Cout,res,rw8", [
    (20, 16, 576, 0, 1728, False, True),     # q|k|v of the 576-channel AttentionBlocks: 6.75 cout tiles (the last one three quarters full)
    (40, 16, 576, 0, 576, True, True),       # proj_out + residual: 2.25 cout tiles
    (20, 32, 384, 192, 384, False, True),    # skip_connection over a virtual concat whose boundary (384) is a chunk boundary
    (40, 16, 320, 256, 576, False, True),    # ... and one whose boundary is not
    (8, 64, 192, 192, 192, False, True),     # K = 384 with a cout count conv1x1_rw_kernel does not take
    (16, 32, 512, 0, 1536, False, True),     # K = 512, wide output (the LSUN net's q|k|v)
    (100, 32, 384, 0, 1152, False, False),   # K = 384, Cout % 128 == 0: conv1x1_rw_kernel keeps it
    (2, 16, 576, 0, 192, False, False),      # fewer pixel tiles than two per stream: the per-tile kernel
])
def test_conv1x1_rw8(ops, N, H, C0, C1, Cout, res, rw8):
    import ctypes
    from dxmi_hip import _lib
    g = torch.Generator().manual_seed(N * 7 + Cout)
    x0 = torch.randn(N, H, H, C0, generator=g).to(torch.bfloat16).to(DEV)
    x1 = torch.randn(N, H, H, C1, generator=g).to(torch.bfloat16).to(DEV) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 1, 1, generator=g) * 0.05).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    r = torch.randn(N, H, H, Cout, generator=g).to(torch.bfloat16).to(DEV) if res else None
    pw = ops.pack_conv_weight(w)
    y = ops.conv2d(x0, pw, in1=x1, bias=b, residual=r)
    d = _lib.ConvDesc()
    d.in0, d.in1, d.wpacked, d.bias, d.out = x0.data_ptr(), (x1.data_ptr() if C1 else None), pw.buf.data_ptr(), b.data_ptr(), y.data_ptr()
    d.residual = r.data_ptr() if res else None
    d.N, d.IH, d.IW, d.C0, d.C1, d.OH, d.OW, d.Cout = N, H, H, C0, C1, H, H, Cout
    d.ksize, d.stride, d.pad = 1, 1, 0
    kid = _lib.load().dxmi_conv2d_kernel_id(ctypes.byref(d))
    assert (550000 <= kid < 600000) == rw8, kid
    xin = torch.cat([x0, x1], 3) if C1 else x0
    ref = xin.float().reshape(-1, C0 + C1) @ w.view(Cout, -1).to(torch.bfloat16).float().t() + b
    if res:
        ref = ref + r.float().reshape(-1, Cout)
    err = ((y.float().reshape(-1, Cout) - ref).norm() / ref.norm()).item()
    assert torch.isfinite(y.float()).all() and err < 4e-3, err
    assert torch.equal(y, ops.conv2d(x0, pw, in1=x1, bias=b, residual=r))          # bitwise reproducible
    if rw8:
        # an image's result does not depend on the batch it rides in although a small batch takes the per-tile kernel: both kernels
        # run the K extent through the same MFMA in the same order and round once
        one = ops.conv2d(x0[:1].contiguous(), pw, in1=None if x1 is None else x1[:1].contiguous(), bias=b,
                         residual=None if r is None else r[:1].contiguous())
        d.N = 1
        assert not (550000 <= _lib.load().dxmi_conv2d_kernel_id(ctypes.byref(d)) < 600000)
        assert torch.equal(one[0], y[0])


def test_groupnorm_apply_split_is_bit_identical():
    """dxmi_groupnorm_apply_split (round 6: per-image finalize launch + prologue-free streaming pass) against dxmi_groupnorm_apply on
    a plain, a virtual-concat and a FiLM scale-shift case: same operations in the same order, so the outputs are equal bit for bit."""
    import torch
    from dxmi_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    for (N, H, C0, C1, ss) in [(5, 32, 128, 0, False), (3, 16, 256, 128, False), (4, 32, 192, 0, True)]:
        C = C0 + C1
        x0 = torch.randn(N, H, H, C0, generator=g).to(torch.bfloat16).to(dev)
        x1 = torch.randn(N, H, H, C1, generator=g).to(torch.bfloat16).to(dev) if C1 else None
        s0, s1 = ops.block_stats(x0), (ops.block_stats(x1) if C1 else None)
        ga, be = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
        sst = (torch.randn(N, 2 * C, generator=g) * 0.1).to(dev) if ss else None
        outs = []
        for split in (False, True):
            ops.GN_APPLY_SPLIT = split
            outs.append(ops.groupnorm_apply(x0, s0, ga, be, in1=x1, st1=s1, eps=1e-5, silu=True, scale_shift=sst))
        ops.GN_APPLY_SPLIT = False
        assert torch.equal(outs[0], outs[1]), (N, H, C0, C1, ss)


def test_blockstats_to_generic_feeds_the_generic_backward():
    """dxmi_gn_blockstats_to_generic (round 6): the block statistics the streaming forward normalised with, in the generic backward's
    partial format — per-group sums equal the generic forward's own partials added up (fp32 summation-order tolerance), and the
    backward run from them matches the backward that recomputes its statistics."""
    import torch
    from dxmi_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    for (N, H, C0, C1, ss) in [(5, 32, 192, 0, False), (3, 16, 384, 192, False), (4, 32, 192, 0, True), (2, 64, 192, 0, False)]:
        C = C0 + C1
        x0 = (torch.randn(N, H, H, C0, generator=g) + 0.3).to(torch.bfloat16).to(dev)
        x1 = torch.randn(N, H, H, C1, generator=g).to(torch.bfloat16).to(dev) if C1 else None
        ga, be = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
        sst = (torch.randn(N, 2 * C, generator=g) * 0.1).to(dev) if ss else None
        dy = torch.randn(N, H, H, C, generator=g).to(torch.bfloat16).to(dev)
        sv_gen, sv_str = [], []
        y_gen = ops.groupnorm_generic(x0, ga, be, in1=x1, eps=1e-5, silu=True, scale_shift=sst, saved=sv_gen)
        y_str = ops.groupnorm_silu(x0, ga, be, in1=x1, eps=1e-5, silu=True, scale_shift=sst, saved=sv_str,
                                   stats=(ops.block_stats(x0), ops.block_stats(x1) if C1 else None))
        assert len(sv_str) == 1 and sv_str[0].shape == sv_gen[0].shape
        a, b = sv_gen[0].view(N, -1, 32, 2).sum(1), sv_str[0].view(N, -1, 32, 2)
        assert float(b[:, 1:].abs().max()) == 0.0                       # the whole image in chunk 0
        torch.testing.assert_close(b[:, 0], a, rtol=2e-5, atol=1e-2)
        torch.testing.assert_close(y_str.float(), y_gen.float(), rtol=2e-2, atol=2e-2)
        ref = ops.groupnorm_generic_bwd(x0, dy, ga, be, in1=x1, eps=1e-5, silu=True, scale_shift=sst)
        got = ops.groupnorm_generic_bwd(x0, dy, ga, be, in1=x1, eps=1e-5, silu=True, scale_shift=sst, fwd_stats=sv_str[0])
        for r, t in zip(ref, got):
            assert (r is None) == (t is None)
            if r is not None:
                rel = float((r.float() - t.float()).norm() / r.float().norm().clamp_min(1e-9))
                assert rel < 2e-3, (N, H, C0, C1, ss, rel)


def test_generic_groupnorm_backward_one_launch_vs_two():
    """gn_gen_bwd_fused_kernel (round 6; knob gn_bwd_fused): rows kept in registers across an in-launch per-image hand-off.  Same
    arithmetic as the reduce + apply launches with the row sums grouped by work chunk instead of statistics chunk: dx within one
    bf16 step, the per-(image, channel) sums to fp32 summation-order tolerance; deterministic (two runs bitwise equal); repeated
    back to back (the counters are re-zeroed per call) and on the EDM nets' shapes incl. a virtual concat, FiLM, an additive
    gradient, ragged last work chunks and the 16-rows-per-thread form."""
    import torch
    from dxmi_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    cases = [(16, 64, 192, 0, False, True), (3, 64, 192, 192, False, False), (16, 32, 384, 0, True, True), (5, 32, 384, 384, False, True),
             (16, 16, 576, 0, True, False), (4, 16, 576, 576, False, True), (16, 8, 768, 0, True, True), (7, 8, 768, 768, False, False),
             (2, 24, 96, 0, False, True), (3, 5, 64, 0, False, False), (3, 4, 1024, 1024, False, True), (2, 8, 2048, 0, True, False)]
    for (N, H, C0, C1, ss, with_add) in cases:
        C = C0 + C1
        x0 = (torch.randn(N, H, H, C0, generator=g) * 1.3 + 0.2).to(torch.bfloat16).to(dev)
        x1 = torch.randn(N, H, H, C1, generator=g).to(torch.bfloat16).to(dev) if C1 else None
        ga, be = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
        sst = (torch.randn(N, 2 * C, generator=g) * 0.1).to(dev) if ss else None
        dy = torch.randn(N, H, H, C, generator=g).to(torch.bfloat16).to(dev)
        a0 = torch.randn(N, H, H, C0, generator=g).to(torch.bfloat16).to(dev) if with_add else None
        a1 = torch.randn(N, H, H, C1, generator=g).to(torch.bfloat16).to(dev) if with_add and C1 else None
        sv = []
        ops.groupnorm_generic(x0, ga, be, in1=x1, eps=1e-5, silu=True, scale_shift=sst, saved=sv)
        run = lambda st: ops.groupnorm_generic_bwd(x0, dy, ga, be, in1=x1, add0=a0, add1=a1, eps=1e-5, silu=True, scale_shift=sst, fwd_stats=st)
        old = ops.set_tuning("gn_bwd_fused", 0)
        try:
            ref = run(sv[0])
            ops.set_tuning("gn_bwd_fused", 2)        # 2: on every shape the plan fits (1, the default: maps of <= 256 pixels)
            got, again, fresh = run(sv[0]), run(sv[0]), run(None)
        finally:
            ops.set_tuning("gn_bwd_fused", old)
        for i, (r, t, t2, t3) in enumerate(zip(ref, got, again, fresh)):
            assert (r is None) == (t is None)
            if r is None:
                continue
            assert torch.isfinite(t.float()).all()
            assert torch.equal(t, t2) and torch.equal(t, t3), (N, H, C0, C1, i)
            if i < 2:       # dx0, dx1: bf16
                diff = (r.float() - t.float()).abs()
                assert float((diff > 0).float().mean()) < 0.05 and float((diff / r.float().abs().clamp_min(1e-2)).max()) < 2e-2, (N, H, C0, C1, i)
            else:
                torch.testing.assert_close(t, r, rtol=2e-4, atol=2e-3 * float(r.abs().max()))


def test_gn_ss_grads_vs_torch():
    """dxmi_gn_ss_grads (round 6): the FiLM GroupNorm's parameter / scale-shift gradients from the backward's per-image sums, against
    the torch expressions it replaced (autograd of models/cm/unet.py:252-256 on G0 / G1)."""
    import ctypes
    import torch
    from dxmi_hip import ops, _lib
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(2)
    for (N, C, pad) in [(16, 384, 0), (5, 1536, 64), (1, 192, 0), (19, 200, 8)]:
        g = torch.randn(2, N, C, generator=gen).to(dev)
        emb = torch.randn(N, 2 * C + pad, generator=gen).to(dev)
        ss = emb[:, pad // 2: pad // 2 + 2 * C]                      # a strided view, as the U-Net passes a slice of emb_all
        ga, be = torch.randn(C, generator=gen).to(dev), torch.randn(C, generator=gen).to(dev)
        d_ss = torch.empty(N, 2 * C, device=dev)
        dgb = torch.empty(2, C, device=dev)
        lib = _lib.load()
        ops.check(lib.dxmi_gn_ss_grads(g.data_ptr(), ss.data_ptr(), ss.stride(0), ga.data_ptr(), be.data_ptr(), d_ss.data_ptr(),
                                       dgb[0].data_ptr(), dgb[1].data_ptr(), N, C, None), "dxmi_gn_ss_grads")
        one_s = 1.0 + ss[:, :C]
        torch.testing.assert_close(d_ss, torch.cat([g[1] * ga + g[0] * be, g[0]], 1), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(dgb[0], (g[1] * one_s).sum(0), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dgb[1], (g[0] * one_s).sum(0), rtol=1e-5, atol=1e-5)


def test_attention_backward_with_the_forwards_log_sum_exp():
    """dxmi_attention_fwd_lse / dxmi_attention_bwd_lse (round 6): the forward's row log-sum-exp equals log2(sum exp(scale q.k)) of an fp32
    restatement, its output is bitwise dxmi_attention_fwd's, and the backward that takes it agrees with the backward that recomputes it
    (64-wide heads at the ADM nets' token counts: both forward kernels, a ragged T)."""
    import math
    import torch
    from dxmi_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(4)
    for (N, T, heads) in [(3, 1024, 6), (2, 256, 9), (5, 64, 12), (2, 100, 3)]:
        C = 64 * heads
        qkv = (torch.randn(N, T, 3 * C, generator=g) * 0.7).to(torch.bfloat16).to(dev)
        do = torch.randn(N, T, C, generator=g).to(torch.bfloat16).to(dev)
        scale = 1.0 / math.sqrt(64)
        o_plain = ops.attention(qkv, heads, scale)
        o, lse = ops.attention(qkv, heads, scale, want_lse=True)
        assert lse is not None and torch.equal(o, o_plain)
        q, k = qkv[:, :, :C].float().view(N, T, heads, 64), qkv[:, :, C:2 * C].float().view(N, T, heads, 64)
        s = torch.einsum("nthd,nshd->nhts", q, k) * scale
        ref = torch.logsumexp(s, dim=-1) / math.log(2.0)
        torch.testing.assert_close(lse, ref, rtol=1e-4, atol=2e-3)
        a = ops.attention_bwd(qkv, do, heads, scale, o=o)
        b = ops.attention_bwd(qkv, do, heads, scale, o=o, lse=lse)
        rel = float((a.float() - b.float()).norm() / a.float().norm())
        assert rel < 2e-3, (N, T, heads, rel)
    assert ops.attention(torch.zeros(1, 256, 768, dtype=torch.bfloat16, device=dev), 1, 1.0, want_lse=True)[1] is None     # attention256_kernel: no lse


def test_var_and_edm_step_backward_vs_torch_autograd():
    """dxmi_var_step_bwd / dxmi_edm_step_bwd (round 6) against torch autograd over the reference's elementwise expressions
    (var_sampler.py:357-408 incl. the log-prob with x' detached; openai_diffusion.py:71-94), every output contributing to the loss."""
    import math
    import torch
    from models.DxMI.openai_diffusion import _EdmStepFn
    from models.DxMI.var_sampler_train import _VarStepFn
    dev = "cuda:0"
    g = torch.Generator().manual_seed(17)
    B, shape = 6, (3, 32, 32)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    x, z = r(B, *shape), r(B, *shape)
    xm, cm = r(B).abs() + 0.5, -r(B).abs() - 0.1
    wn, wm, wc, wl = r(B, *shape), r(B, *shape), r(B, *shape), r(B)          # weights of the four outputs in the loss
    e = lambda v: v[:, None, None, None]

    def loss_of(xn, mean, control, logp):
        return (xn * wn).sum() + (mean * wm).sum() + (control * wc).sum() + (logp * wl).sum()
    eps1 = r(B, *shape).requires_grad_(True)
    lb1 = (r(B) * 0.3 - 1.0).requires_grad_(True)
    sg = torch.exp(lb1)
    mean = x * e(xm) + e(cm) * eps1
    xn = mean + e(sg) * z
    lp = (-((xn.detach() - mean) ** 2) / (2 * e(sg) ** 2) - torch.log(e(sg)) - math.log(math.sqrt(2 * math.pi))).mean((1, 2, 3))
    loss_of(xn, mean, e(cm) * eps1, lp).backward()
    eps2 = eps1.detach().clone().requires_grad_(True)
    lb2 = lb1.detach().clone().requires_grad_(True)
    out = _VarStepFn.apply(x, eps2, z, xm, cm, torch.exp(lb2))
    assert torch.allclose(out[0], xn.detach(), atol=1e-5) and torch.allclose(out[3], lp.detach(), atol=1e-5)
    loss_of(*out).backward()
    assert torch.allclose(eps2.grad, eps1.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(lb2.grad, lb1.grad, rtol=1e-4, atol=1e-4), (lb2.grad, lb1.grad)
    # EDM
    sd = 0.5
    sigma, sdn = r(B).abs() * 3 + 0.05, r(B).abs() * 0.5 + 0.01
    f1 = r(B, *shape).requires_grad_(True)
    u1 = (r(B) * 0.3 - 0.5).requires_grad_(True)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2) ** 0.5
    den = e(c_out) * f1 + e(c_skip) * x
    mu = x + (x - den) / e(sigma) * e(sdn - sigma)
    smp = mu + z * e(torch.exp(u1))
    ((smp * wn).sum() + (mu * wm).sum()).backward()
    f2 = f1.detach().clone().requires_grad_(True)
    u2 = u1.detach().clone().requires_grad_(True)
    s2, m2 = _EdmStepFn.apply(x, f2, z, sigma, sdn, torch.exp(u2), sd)
    assert torch.allclose(s2, smp.detach(), rtol=1e-5, atol=1e-4) and torch.allclose(m2, mu.detach(), rtol=1e-5, atol=1e-4)
    ((s2 * wn).sum() + (m2 * wm).sum()).backward()
    assert torch.allclose(f2.grad, f1.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(u2.grad, u1.grad, rtol=1e-4, atol=1e-3)
