"""Round-2 kernels at the shapes that select them (conv_ws_kernel, conv1x1_rw_kernel, attention256_kernel), against torch fp32
on the same bf16-rounded operands: persistent loops with uneven tile counts per workgroup, single-tile launches, channel
concat, upsample, every fused epilogue term, plus run-to-run determinism and batch independence.  The kernel each case runs is
asserted through dxmi_conv2d_kernel_id.  Tolerance: one bf16 rounding of the output (rel-L2 <= 4e-3)."""
import math

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

@pytest.fixture(scope="module", autouse=True)
def kernels_under_test(ops):
    """These cases put deliberately small grids (one tile, a handful of images) on the wave-specialised kernels: switch off the
    library's small-grid re-routing (include/dxmi_hip.h: dxmi_set_tuning) for this module; test_small_grid_routing covers it."""
    old = ops.set_tuning("conv_ws_min_tiles", 0), ops.set_tuning("conv_sm_mask", 1)
    yield
    ops.set_tuning("conv_ws_min_tiles", old[0])
    ops.set_tuning("conv_sm_mask", old[1])



def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def run_conv(ops, N, C0, C1, Cout, H, k, ups, fuse, seed=0):
    """-> (output NHWC bf16 on the device, fp32 reference NCHW, kernel id)"""
    g = torch.Generator().manual_seed(seed + N * 1000003 + C0 * 131 + C1 * 17 + Cout * 7 + H + k)
    Cin = C0 + C1
    x = bf(torch.randn(N, Cin, H, H, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g)
    xd = x.to(DEV)
    xi = F.interpolate(xd, scale_factor=2.0, mode="nearest") if ups else xd
    ref = F.conv2d(xi, w.to(DEV), b.to(DEV) if "bias" in fuse else None, padding=k // 2)
    kw = {}
    if "res" in fuse:
        res = bf(torch.randn(ref.shape, generator=g))
        ref = ref + res.to(DEV)
        kw["residual"] = nhwc(res)
    if "vec" in fuse:
        vec = torch.randn(N, Cout, generator=g)
        ref = ref + vec.to(DEV)[:, :, None, None]
        kw["addvec"] = vec.to(DEV)
    if "act" in fuse:
        ref = F.leaky_relu(ref, 0.2)
        kw["act"] = ops.ACT_LEAKY02
    prof = ops.OpProfiler()
    ops.PROFILER = prof
    try:
        y = ops.conv2d(nhwc(x[:, :C0]), ops.pack_conv_weight(w.to(DEV)), in1=nhwc(x[:, C0:]) if C1 else None,
                       bias=b.to(DEV) if "bias" in fuse else None, pad=k // 2, upsample=ups, **kw)
    finally:
        ops.PROFILER = None
    torch.cuda.synchronize()
    return y, ref.cpu(), prof.records[-1][1]


# (N, C0, C1, Cout, H, ups, fuse): 3x3 shapes in conv_ws_kernel's scope
WS_CASES = [
    (1, 128, 0, 128, 32, False, "none"),                    # 4 tiles on 256 CUs
    (1, 128, 0, 128, 16, False, "bias"),                    # ONE tile
    (3, 256, 128, 256, 16, False, "bias+res"),              # concat, 2 cout tiles, 6 tiles
    (67, 128, 128, 128, 32, False, "bias+res+vec+act"),     # 268 tiles: 12 workgroups walk two tiles, the rest one
    (150, 256, 256, 256, 16, False, "bias+vec"),            # 300 tiles over 256 workgroups, 144 steps per tile
    (40, 256, 0, 384, 32, False, "bias+res+vec"),           # three cout tiles, 480 tiles
    (5, 256, 0, 256, 16, True, "bias"),                     # nearest x2 upsample in front (32x32 output)
    (2, 128, 0, 128, 64, False, "bias+res"),                # 64x64 map: two 32-wide tile columns
    (9, 384, 0, 128, 32, False, "bias+vec"),                # 12 chunks
    (6, 192, 0, 192, 32, False, "bias+res+vec"),            # Cout % 128 == 64: half-empty last cout tile (EDM nets: 192, 576)
    (3, 128, 0, 64, 16, False, "bias+res"),                 # a single, half-empty cout tile
    (70, 384, 192, 576, 16, False, "bias+res+vec+act"),     # 4.5 cout tiles, 350 tiles: tile switches into and out of the half tile
]


@pytest.mark.parametrize("N,C0,C1,Cout,H,ups,fuse", WS_CASES)
def test_conv_ws(ops, N, C0, C1, Cout, H, ups, fuse):
    y, ref, kid = run_conv(ops, N, C0, C1, Cout, H, 3, ups, fuse)
    assert 400000 <= kid < 500000, f"expected conv_ws_kernel, got kernel id {kid}"
    assert nchw(y).shape == ref.shape
    assert rel_l2(nchw(y), ref) < 4e-3


# (N, C0, C1, Cout, fuse): 3x3 shapes on 8x8 maps in conv_ws8_kernel's scope (tiles of four images x 64 couts)
WS8_CASES = [
    (128, 256, 0, 256, "bias+res+vec"),                    # 128 tiles, one per workgroup
    (132, 256, 256, 256, "bias+vec+act"),                  # concat, 16 chunks
    (260, 256, 0, 256, "bias+res+vec"),                    # 260 tiles over 256 workgroups: four walk two tiles (tile switch)
    (520, 128, 0, 128, "bias+res"),                        # 4 chunks, two cout tiles, 260 tiles
    (64, 256, 0, 512, "none"),                             # eight cout tiles
    (1, 256, 0, 256, "bias+res+vec"),                      # one image: three quarters of the only tile masked
    (7, 256, 0, 256, "bias+res+vec"),                      # last tile holds three images
    (1031, 128, 0, 64, "bias+res"),                        # 258 tiles, the last one with three images, reached through a tile switch
]


@pytest.mark.parametrize("N,C0,C1,Cout,fuse", WS8_CASES)
def test_conv_ws8(ops, N, C0, C1, Cout, fuse):
    y, ref, kid = run_conv(ops, N, C0, C1, Cout, 8, 3, False, fuse)
    assert kid == 400008, f"expected conv_ws8_kernel, got kernel id {kid}"
    assert rel_l2(nchw(y), ref) < 4e-3
    y2, _, _ = run_conv(ops, N, C0, C1, Cout, 8, 3, False, fuse)
    assert torch.equal(y, y2)


# (N, C0, C1, Cout, H, fuse): 1x1 shapes in conv1x1_rw_kernel's scope (>= 32768 pixels)
RW_CASES = [
    (128, 256, 0, 768, 16, "bias"),                         # q|k|v: six cout tiles share a pixel stream
    (130, 256, 0, 768, 16, "bias"),                         # 520 tiles over 80 streams: uneven
    (128, 256, 0, 256, 16, "bias+res"),                     # proj_out
    (131, 256, 256, 256, 16, "bias+res"),                   # K = 512: one workgroup per CU, deeper ring
    (33, 128, 128, 128, 32, "bias+res"),                    # concat shortcut at 32x32
    (32, 128, 0, 256, 32, "none"),                          # K = 128: one chunk per tile
    (35, 256, 128, 384, 32, "bias+act"),                    # K = 384, three cout tiles
]


@pytest.mark.parametrize("N", [256, 5])
def test_conv_ws8_upsample(ops, N):
    """Upsample 4x4 -> 8x8 (nearest x2 in front of the 3x3 conv, unet_small.py:50-54) on conv_ws8_kernel, batch-independent."""
    y, ref, kid = run_conv(ops, N, 256, 0, 256, 4, 3, True, "bias")
    assert kid == 400008, f"expected conv_ws8_kernel, got kernel id {kid}"
    assert rel_l2(nchw(y), ref) < 4e-3
    y2, _, _ = run_conv(ops, N, 256, 0, 256, 4, 3, True, "bias")
    assert torch.equal(y, y2)


@pytest.mark.parametrize("N,C0,C1,Cout,H,fuse", RW_CASES)
def test_conv1x1_rw(ops, N, C0, C1, Cout, H, fuse):
    y, ref, kid = run_conv(ops, N, C0, C1, Cout, H, 1, False, fuse)
    assert kid >= 500000, f"expected conv1x1_rw_kernel, got kernel id {kid}"
    assert rel_l2(nchw(y), ref) < 4e-3


def test_small_grid_routing(ops):
    """Round 4: with the library's default knobs a 3x3 conv whose (256-pixel, 128-cout) grid leaves most CUs idle runs on a
    kernel with smaller tiles (the EDM nets at the train batch of 16), larger grids stay on the wave-specialised kernels; every
    route agrees with the fp32 reference."""
    old = ops.set_tuning("conv_ws_min_tiles", 96), ops.set_tuning("conv_sm_mask", 5)
    try:
        assert ops.get_tuning("conv_ws_min_tiles") == 96 and ops.get_tuning("conv_sm_mask") == 5
        for (N, C, Cout, H, want) in [(16, 576, 576, 16, "pipe"),     # 80 tiles
                                      (16, 768, 768, 8, "sm"),        # 24 conv_ws8 tiles -> 192 feed-tiled workgroups
                                      (16, 384, 384, 32, "ws"),       # 192 tiles
                                      (100, 768, 768, 8, "ws8"),      # 150 tiles
                                      (24, 576, 576, 16, "ws")]:      # 120 tiles
            y, ref, kid = run_conv(ops, N, C, 0, Cout, H, 3, False, "bias+res")
            got = "ws8" if kid == 400008 else "ws" if 400000 <= kid < 400100 else "sm" if 450000 <= kid < 460000 else "pipe" if kid < 400000 else kid
            assert got == want, (N, C, Cout, H, kid)
            assert rel_l2(nchw(y), ref) < 4e-3, (N, C, Cout, H)
        with pytest.raises(Exception, match="unknown knob"):
            ops.set_tuning("no_such_knob", 1)
    finally:
        ops.set_tuning("conv_ws_min_tiles", old[0])
        ops.set_tuning("conv_sm_mask", old[1])


def test_wide_8x8_layers_take_the_feed_tiled_kernel(ops):
    """Library defaults (knobs 0 / 9): the one shape rule that is on by default depends on the LAYER only — 8x8 maps with >= 1024
    channels in and out (the bottom of the LSUN-256 net) run on conv_sm_kernel at every batch size, so a sub-batch reproduces
    the rows of the full batch bit for bit; narrower 8x8 layers stay on conv_ws8_kernel."""
    old = ops.set_tuning("conv_ws_min_tiles", 0), ops.set_tuning("conv_sm_mask", 9)
    try:
        for (N, C, want) in [(16, 1024, "sm"), (3, 1024, "sm"), (16, 768, "ws8")]:
            y, ref, kid = run_conv(ops, N, C, 0, C, 8, 3, False, "bias+res")
            assert (450000 <= kid < 460000) == (want == "sm") and (kid == 400008) == (want == "ws8"), (N, C, kid)
            assert rel_l2(nchw(y), ref) < 4e-3
        g = torch.Generator().manual_seed(77)
        x = torch.randn(16, 8, 8, 1024, generator=g).to(torch.bfloat16).to(DEV)
        pw = ops.pack_conv_weight((torch.randn(1024, 1024, 3, 3, generator=g) * 0.01).to(DEV))
        full = ops.conv2d(x, pw)
        assert torch.equal(ops.conv2d(x[5:8].contiguous(), pw), full[5:8])
    finally:
        ops.set_tuning("conv_ws_min_tiles", old[0])
        ops.set_tuning("conv_sm_mask", old[1])


def test_round2_kernels_out_of_scope_shapes_fall_back(ops):
    """shapes just outside the new kernels' scope still run (on the round-1 kernels) and agree with the reference"""
    for (N, C0, C1, Cout, H, k, fuse) in [(2, 128, 0, 96, 32, 3, "bias"),       # Cout % 64 != 0
                                          (4, 128, 0, 256, 4, 3, "bias+res"),    # 4x4 map, four chunks (conv_sm_kernel takes >= 5)
                                          (2, 96, 0, 128, 32, 3, "bias"),        # odd chunk count
                                          (8, 256, 0, 256, 16, 1, "bias"),       # 1x1 with too few pixels
                                          (128, 192, 0, 384, 16, 1, "bias")]:    # 1x1 with K % 128 != 0
        y, ref, kid = run_conv(ops, N, C0, C1, Cout, H, k, False, fuse)
        assert kid < 400000, (kid, N, C0, Cout, H, k)
        assert rel_l2(nchw(y), ref) < 4e-3


@pytest.mark.parametrize("N,C0,C1,Cout,H,k,fuse", [(70, 128, 128, 128, 32, 3, "bias+res+vec"), (260, 256, 0, 768, 16, 1, "bias"),
                                                   (258, 256, 0, 256, 16, 1, "bias+res")])
def test_round2_convs_deterministic_and_batch_independent(ops, N, C0, C1, Cout, H, k, fuse):
    """bitwise: the same launch twice; and image i of a batch does not depend on the rest of the batch (persistent loops, DMA
    rings and LDS tiles carry no state across tiles)"""
    y1, _, _ = run_conv(ops, N, C0, C1, Cout, H, k, False, fuse)
    y2, _, _ = run_conv(ops, N, C0, C1, Cout, H, k, False, fuse)
    assert torch.equal(y1, y2)
    # same operands, batch cut to the first M images (run_conv seeds by N, so rebuild the operands by hand)
    g = torch.Generator().manual_seed(7)
    Cin = C0 + C1
    x = bf(torch.randn(N, Cin, H, H, generator=g))
    pw = ops.pack_conv_weight(bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(DEV))
    b = torch.randn(Cout, generator=g).to(DEV)
    res = nhwc(bf(torch.randn(N, Cout, H, H, generator=g)))
    x0, x1 = nhwc(x[:, :C0]), (nhwc(x[:, C0:]) if C1 else None)
    M = N // 2 + (1 if k == 3 else 0)      # the cut batch stays in the same kernel's scope
    full = ops.conv2d(x0, pw, in1=x1, bias=b, residual=res, pad=k // 2)
    part = ops.conv2d(x0[:M].contiguous(), pw, in1=x1[:M].contiguous() if C1 else None, bias=b, residual=res[:M].contiguous(), pad=k // 2)
    torch.cuda.synchronize()
    assert torch.equal(full[:M], part)


@pytest.mark.parametrize("N", [1, 5, 64])
def test_attention256(ops, N):
    T, C = 256, 256
    g = torch.Generator().manual_seed(11 + N)
    qkv = bf(torch.randn(N, T, 3 * C, generator=g) * 1.5)
    scale = 1.0 / math.sqrt(C)
    q, k, v = qkv.to(DEV).split(C, dim=2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v).cpu()
    y1 = ops.attention(qkv.to(torch.bfloat16).to(DEV), 1, scale)
    y2 = ops.attention(qkv.to(torch.bfloat16).to(DEV), 1, scale)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    # P is rounded to bf16 before PV: ~2^-9 relative per term
    assert rel_l2(y1.float().cpu(), ref) < 8e-3
    # batch independence: image 0 alone
    y0 = ops.attention(qkv[:1].to(torch.bfloat16).to(DEV), 1, scale)
    assert torch.equal(y0[0], y1[0])


@pytest.mark.parametrize("N,H,Cout", [(1, 32, 3), (7, 32, 3), (2, 64, 3), (3, 32, 6)])
def test_conv_head(ops, N, H, Cout):
    """conv_out (128 -> few channels, NCHW fp32 output): conv_head_kernel, borders and partial cout blocks included"""
    g = torch.Generator().manual_seed(3 + N + H + Cout)
    x = bf(torch.randn(N, 128, H, H, generator=g))
    w = bf(torch.randn(Cout, 128, 3, 3, generator=g) / math.sqrt(128 * 9))
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    prof = ops.OpProfiler()
    ops.PROFILER = prof
    try:
        y = ops.conv2d(nhwc(x), ops.pack_conv_weight(w.to(DEV)), bias=b.to(DEV), out_nchw_f32=True)
        y2 = ops.conv2d(nhwc(x), ops.pack_conv_weight(w.to(DEV)), bias=b.to(DEV), out_nchw_f32=True)
    finally:
        ops.PROFILER = None
    torch.cuda.synchronize()
    assert prof.records[-1][1] == 600000, prof.records[-1][1]
    assert y.dtype == torch.float32 and tuple(y.shape) == tuple(ref.shape)
    assert torch.equal(y, y2)
    assert rel_l2(y.cpu(), ref) < 2e-3      # fp32 output: only the bf16 operands' products, fp32 accumulation
