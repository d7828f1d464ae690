"""f4: the FID InceptionV3 extractor on the HIP kernels (pytorch_fid/inception.py).  PARITY UNPINNED — torchvision and the FID weight
file are absent from the image — so the HIP program is checked against the oracle's torch-CPU restatement of the published
architecture (oracle/inception.py) on formula weights; the host-side contract (reference class surface, state-dict keys, weight
loading by torchvision names) is checked on the CPU."""
import numpy as np
import pytest
import torch


def _tv_state_dict(model):
    """A torchvision-named state dict (the FID weight file's naming) with formula weights that keep activations O(1) through the
    ~45 conv layers: He-scaled conv weights, BatchNorm statistics near identity."""
    from oracle.weights import formula_tensor
    sd = {}
    for name, c in model._convs():
        w = formula_tensor(name + ".conv.weight", c.conv.weight.shape) * (6.0 ** 0.5)          # uniform(+-1/sqrt(fan_in)) -> variance 2 / fan_in
        n = c.bn.weight.numel()
        f = lambda k: formula_tensor(f"{name}.bn.{k}", (n,)) * (n ** 0.5)                      # uniform(+-1)
        sd[name + ".conv.weight"] = w
        sd[name + ".bn.weight"] = 1.0 + 0.2 * f("weight")
        sd[name + ".bn.bias"] = 0.1 * f("bias")
        sd[name + ".bn.running_mean"] = 0.1 * f("running_mean")
        sd[name + ".bn.running_var"] = 1.0 + 0.3 * f("running_var").abs()
        sd[name + ".bn.num_batches_tracked"] = torch.tensor(0)
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(1008, 2048), torch.zeros(1008)              # present in the FID file, ignored here
    return sd


def test_inception_surface_and_weight_loading_cpu():
    from pytorch_fid.inception import InceptionV3
    m = InceptionV3()
    assert InceptionV3.BLOCK_INDEX_BY_DIM == {64: 0, 192: 1, 768: 2, 2048: 3} and m.last_needed_block == 3 and len(m.blocks) == 4
    keys = list(m.state_dict().keys())
    assert keys[0] == "blocks.0.0.conv.weight" and "blocks.2.0.branch5x5_2.bn.running_var" in keys and "blocks.3.2.branch3x3dbl_3b.conv.weight" in keys
    assert sum(p.numel() for p in m.parameters()) == 21_785_568 and not any(p.requires_grad for p in m.parameters())
    shapes = {n: tuple(c.conv.weight.shape) for n, c in m._convs()}
    assert shapes["Conv2d_1a_3x3"] == (32, 3, 3, 3) and shapes["Mixed_5b.branch5x5_2"] == (64, 48, 5, 5)
    assert shapes["Mixed_6b.branch7x7_2"] == (128, 128, 1, 7) and shapes["Mixed_6e.branch7x7dbl_4"] == (192, 192, 7, 1)
    assert shapes["Mixed_7a.branch3x3_2"] == (320, 192, 3, 3) and shapes["Mixed_7c.branch3x3dbl_1"] == (448, 2048, 1, 1) and len(shapes) == 94
    sd = _tv_state_dict(m)
    m.load_fid_weights(sd)
    assert torch.equal(m.blocks[2][4].branch7x7_3.conv.weight, sd["Mixed_6b.branch7x7_3.conv.weight"])
    bad = dict(sd)
    del bad["Mixed_7b.branch_pool.bn.weight"]
    with pytest.raises(RuntimeError):
        InceptionV3().load_fid_weights(bad)
    with pytest.raises(Exception):
        m(torch.rand(1, 3, 32, 32))                    # no CPU path
    m2 = InceptionV3(output_blocks=[1])
    assert len(m2.blocks) == 2 and len(list(m2._convs())) == 5


def test_fid_inception_needs_the_weight_file(monkeypatch, tmp_path):
    """`--fid_extractor pytorch_fid.inception:FIDInceptionV3`: loud failure without DXMI_FID_WEIGHTS, the torchvision-named file
    loaded when it is there."""
    from dxmi_hip import DxmiError
    from pytorch_fid.fid_score import load_extractor
    from pytorch_fid.inception import InceptionV3
    monkeypatch.delenv("DXMI_FID_WEIGHTS", raising=False)
    with pytest.raises(DxmiError):
        load_extractor("pytorch_fid.inception:FIDInceptionV3")
    sd = _tv_state_dict(InceptionV3())
    path = tmp_path / "pt_inception.pth"
    torch.save(sd, path)
    monkeypatch.setenv("DXMI_FID_WEIGHTS", str(path))
    m = load_extractor("pytorch_fid.inception:FIDInceptionV3")
    assert m.output_blocks == [3] and torch.equal(m.blocks[3][2].branch_pool.conv.weight, sd["Mixed_7c.branch_pool.conv.weight"])


@pytest.mark.gpu
def test_generic_conv_and_pools_vs_torch():
    """dxmi_gconv_fwd on the kernel shapes of the net (1x1, 3x3 s2, 5x5, 1x7, 7x1, odd map sizes, cout padding, channel-window output),
    the two pools and the bilinear resize against torch fp32 on the bf16-rounded operands."""
    import torch.nn.functional as F
    from dxmi_hip import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    for (cin, cout, k, s, p, h) in [(16, 32, (3, 3), (2, 2), (0, 0), 37), (48, 64, (5, 5), (1, 1), (2, 2), 35), (64, 80, (1, 1), (1, 1), (0, 0), 19),
                                    (128, 128, (1, 7), (1, 1), (0, 3), 17), (160, 192, (7, 1), (1, 1), (3, 0), 17), (192, 320, (3, 3), (2, 2), (0, 0), 17)]:
        x = torch.randn(3, cin, h, h, generator=g)
        w = torch.randn(cout, cin, *k, generator=g) * (2.0 / (cin * k[0] * k[1])) ** 0.5
        bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
        xb = x.to(torch.bfloat16)
        pk = ops.gconv_pack(w.to(dev), tuple(t.to(dev) for t in bn), eps=1e-3)
        wide = torch.zeros(3, (h + 2 * p[0] - k[0]) // s[0] + 1, (h + 2 * p[1] - k[1]) // s[1] + 1, cout + 24, dtype=torch.bfloat16, device=dev)
        ops.gconv(xb.permute(0, 2, 3, 1).contiguous().to(dev), pk, stride=s, pad=p, out=wide, coff=8)
        scale = bn[0] / torch.sqrt(bn[3] + 1e-3)
        wf = (w * scale[:, None, None, None]).to(torch.bfloat16).float()
        ref = F.relu(F.conv2d(xb.float(), wf, None, stride=s, padding=p) + (bn[1] - bn[2] * scale)[None, :, None, None])
        got = wide[..., 8:8 + cout].float().cpu().permute(0, 3, 1, 2)
        assert float((got - ref).norm() / ref.norm()) < 6e-3, (cin, cout, k)
        assert float(wide[..., :8].abs().max()) == 0 and float(wide[..., 8 + cout:].abs().max()) == 0       # nothing outside the window
    x = torch.randn(2, 32, 17, 17, generator=g).to(torch.bfloat16)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    for (stride, pad, avg) in [(2, 0, False), (1, 1, False), (1, 1, True)]:
        got = ops.pool3x3(xd, stride, pad, avg_exclude_pad=avg).float().cpu().permute(0, 3, 1, 2)
        ref = F.avg_pool2d(x.float(), 3, stride, pad, count_include_pad=False) if avg else F.max_pool2d(x.float(), 3, stride, pad)
        assert torch.allclose(got, ref.to(torch.bfloat16).float(), atol=1e-2, rtol=1e-2)
    assert torch.allclose(ops.global_avgpool(xd).cpu(), x.float().mean((2, 3)), atol=1e-5)
    img = torch.rand(2, 3, 32, 32, generator=g)
    got = ops.resize_bilinear_nhwc16(img.to(dev), 299, 299, normalize=True).float().cpu()
    ref = 2 * F.interpolate(img, size=(299, 299), mode="bilinear", align_corners=False) - 1
    assert float(got[..., 3:].abs().max()) == 0
    assert torch.allclose(got[..., :3].permute(0, 3, 1, 2), ref, atol=8e-3)


@pytest.mark.gpu
def test_inception_forward_vs_oracle():
    """All four blocks of the HIP InceptionV3 (bf16 activations, BatchNorm folded into bf16 weights) against the fp32 oracle on formula
    weights, 32x32 inputs resized to 299x299 as the FID of CIFAR-10 does: relative L2 per block <= 3e-2 (a bf16 storage chain ~45
    layers deep), pool3 features of different images clearly apart (the extractor is not collapsing)."""
    from oracle import inception as oinc
    from pytorch_fid.inception import InceptionV3
    torch.set_num_threads(8)
    m = InceptionV3(output_blocks=[0, 1, 2, 3])
    sd = _tv_state_dict(m)
    m.load_fid_weights(sd)
    m = m.to("cuda:0")
    img = torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    outs = [o.cpu() for o in m(img.to("cuda:0"))]
    with torch.no_grad():
        refs = oinc.forward(sd, img)
    assert [tuple(o.shape) for o in outs] == [(3, 64, 73, 73), (3, 192, 35, 35), (3, 768, 17, 17), (3, 2048, 1, 1)]
    rel = [float((o - r).norm() / r.norm()) for o, r in zip(outs, refs)]
    print("inception blocks rel-L2 vs oracle fp32:", [f"{v:.2e}" for v in rel])
    assert all(v < 3e-2 for v in rel), rel
    f = refs[3].flatten(1)
    assert float((f[0] - f[1]).norm() / f[0].norm()) > 3 * rel[3]
    # the scripts' entry point: get_activations_from_tensor(model(batch)[0])
    from pytorch_fid.fid_score import get_activations_from_tensor
    m3 = InceptionV3(weights=sd).to("cuda:0")
    act = get_activations_from_tensor(img.to("cuda:0"), m3, batch_size=2, dims=2048, device="cuda:0")
    assert tuple(act.shape) == (3, 2048) and float((act.cpu() - refs[3].flatten(1)).norm() / refs[3].norm()) < 3e-2
