"""CLI surface: `--a.b.c v` overrides, YAML/_target_ config layer, checkpoint key fix-up (CPU)."""
import os

import pytest
import torch


def test_parse_arg_type():
    from cmd_utils import parse_arg_type as p
    assert p("3") == 3 and isinstance(p("3"), int)
    assert p("1e-6") == 1e-6 and p("0.5") == 0.5 and p("-2") == -2.0
    assert p("true") is True and p("False") is False
    assert p("null") is None and p("None") is None
    assert p("[1,2,2,2]") == [1, 2, 2, 2]
    assert p("fix_last") == "fix_last"


def test_unknown_and_nested_args():
    from cmd_utils import parse_nested_args, parse_unknown_args
    d = parse_unknown_args(["--training.lr", "1e-6", "--trainer.tau1", "0.1", "--run", "x"])
    assert d == {"training.lr": 1e-6, "trainer.tau1": 0.1, "run": "x"}
    assert parse_nested_args(d) == {"training": {"lr": 1e-6}, "trainer": {"tau1": 0.1}, "run": "x"}
    with pytest.raises(AssertionError):
        parse_unknown_args(["--a=1", "2"])


def test_config_merge_instantiate_roundtrip(tmp_path):
    import configs_builtin
    import dxmi_config
    cfg = configs_builtin.get("cifar10_T10")
    cfg = dxmi_config.merge(cfg, {"training": {"lr": 1e-6}, "sampler": {"n_timesteps": 4}})
    assert cfg.training.lr == 1e-6 and cfg.training.v_lr == 1e-5 and cfg.sampler.n_timesteps == 4
    assert cfg.training.get("fid_every", None) is None and cfg.energy is None
    path = os.path.join(tmp_path, "config.yaml")
    dxmi_config.save(cfg, path)
    cfg2 = dxmi_config.load(path)
    assert dxmi_config.to_container(cfg2) == dxmi_config.to_container(cfg)
    net = dxmi_config.instantiate(cfg2.sampler_net)
    sampler = dxmi_config.instantiate(cfg2.sampler, net=net)
    assert type(sampler).__name__ == "VARSampler" and sampler.n_timesteps == 4 and sampler.net is net
    v = dxmi_config.instantiate(cfg2.value)
    assert type(v).__name__ == "TimeIndependentValue" and type(v.net).__name__ == "IGEBMEncoderV2"
    assert dxmi_config.instantiate(cfg2.energy) is None


def test_fix_legacy_dict():
    from utils import fix_legacy_dict
    sd = {"module.a.weight": torch.zeros(1), "module.b.bias": torch.ones(1)}
    out = fix_legacy_dict({"model": sd})
    assert list(out.keys()) == ["a.weight", "b.bias"]
    assert list(fix_legacy_dict({"state_dict": {"x": 1, "y": 2}}).keys()) == ["x", "y"]


def test_png_writer_roundtrip(tmp_path):
    """The thread-pool PNG writer produces standard 8-bit RGB PNGs with save_image's rounding."""
    import numpy as np
    import torch
    from PIL import Image
    from utils import to_uint8_nhwc, write_png_batch
    g = torch.Generator().manual_seed(0)
    x = torch.rand(9, 3, 32, 32, generator=g)
    u8 = to_uint8_nhwc(x)
    ref = x.mul(255).add(0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    assert u8.shape == (9, 32, 32, 3) and np.array_equal(u8, ref)
    paths = [str(tmp_path / f"0_{i}.png") for i in range(9)]
    write_png_batch(u8, paths, workers=4)
    for i, p in enumerate(paths):
        im = Image.open(p)
        assert im.mode == "RGB" and im.size == (32, 32) and np.array_equal(np.asarray(im), u8[i])
