"""EDM / ADM path: pin oracle/edm.py against golden vectors produced by the reference itself
(tests/golden/make_golden.py --only edm) and check the host side of the product modules
(schedule tables, state-dict key parity).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import Precision, edm
from oracle.weights import formula_tensor

torch.set_num_threads(8)

TINY = dict(image_size=16, model_channels=64, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2))
VARIANTS = {"": edm.EDMConfig(**TINY),
            "_plain": edm.EDMConfig(num_classes=None, use_scale_shift_norm=False, resblock_updown=False, **TINY)}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def formula_sd(cfg):
    return {k: formula_tensor(k, s) for k, s in edm.state_dict_shapes(cfg).items()}


def oracle_model(sd, cfg, prec=None):
    return lambda x, t, **kw: edm.unet_forward(sd, cfg, x, t, prec=prec, **kw)


@pytest.mark.parametrize("tag", ["", "_plain"])
def test_state_dict_layout_matches_reference(golden_dir, tag):
    g = load(golden_dir, f"edm_unet_forward{tag}")
    sh = edm.state_dict_shapes(VARIANTS[tag])
    assert list(sh.keys()) == [str(k) for k in g["state_keys"]]
    for (k, s), ref in zip(sh.items(), g["state_shapes"]):
        assert list(s) == [int(v) for v in ref[:len(s)]], k
    assert sum(int(np.prod(s)) for s in sh.values()) == int(g["n_params"])


@pytest.mark.parametrize("tag", ["", "_plain"])
def test_unet_forward_and_denoise_match_reference(golden_dir, tag):
    cfg = VARIANTS[tag]
    g = load(golden_dir, f"edm_unet_forward{tag}")
    sd = formula_sd(cfg)
    x, t, y = (torch.from_numpy(g[k]) for k in ("x", "t", "y"))
    kw = {"y": y} if cfg.num_classes else {}
    with torch.no_grad():
        out = edm.unet_forward(sd, cfg, x, t, **kw)
        mo, den = edm.denoise(oracle_model(sd, cfg), x * 5, torch.from_numpy(g["sigma"]), **kw)
    # same ops in the same order on the same torch build: fp32 round-off only
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(mo.numpy(), g["model_output"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(den.numpy(), g["denoised"], rtol=0, atol=2e-5)


def test_schedule_tables_match_reference(golden_dir):
    g = load(golden_dir, "edm_schedule")
    for name, kw in (("T10", dict(n_timesteps=10)), ("T4", dict(n_timesteps=4, stochastic_last=True, rho=4.0))):
        s = edm.EDMSchedule(trainable_beta="fix_last", **kw)
        for k in ("sigmas", "sigma_up", "sigma_down", "log_betas"):
            np.testing.assert_array_equal(getattr(s, k).numpy(), g[f"{k}_{name}"], err_msg=f"{k}_{name}")
    # known answers: the ladder starts at sigma_max and ends at sigma_min (then the appended 0)
    s = edm.EDMSchedule(10)
    assert abs(float(s.sigmas[0]) - 80.0) < 1e-4 and abs(float(s.sigmas[9]) - 0.002) < 1e-7 and float(s.sigmas[10]) == 0.0


@pytest.mark.parametrize("tag", ["", "_plain"])
def test_sampling_and_sample_step_match_reference(golden_dir, tag):
    cfg = VARIANTS[tag]
    sd = formula_sd(cfg)
    sch = edm.EDMSchedule(4, stochastic_last=True, rho=4.0, trainable_beta="fix_last")
    g = load(golden_dir, f"edm_sampling_T4{tag}")
    torch.manual_seed(int(g["seed"]))
    x0 = torch.randn(2, 3, 16, 16) * 80.0
    zs = [torch.randn(2, 3, 16, 16) for _ in range(4)]
    kw = {"y": torch.tensor([7, 7])} if cfg.num_classes else {}
    with torch.no_grad():
        d = edm.sample(oracle_model(sd, cfg), sch, x0, zs, **kw)
    np.testing.assert_allclose(torch.stack(d["l_sample"]).numpy(), g["l_sample"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(torch.stack(d["mean"]).numpy(), g["mean"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(torch.stack(d["sigma"]).numpy(), g["sigma"])

    g = load(golden_dir, f"edm_sample_step_T4{tag}")
    torch.manual_seed(int(g["seed"]))
    z = torch.randn(2, 3, 16, 16)
    kw = {"y": torch.from_numpy(g["y"])} if cfg.num_classes else {}
    with torch.no_grad():
        ds = edm.sample_step(oracle_model(sd, cfg), sch, torch.from_numpy(g["x"]), torch.from_numpy(g["idx"]), z, **kw)
    for k in ("sample", "mean", "sigma"):
        np.testing.assert_allclose(ds[k].numpy(), g[k], rtol=0, atol=1e-5, err_msg=k)


def test_bf16_storage_model_stays_near_fp32(golden_dir):
    cfg = VARIANTS[""]
    g = load(golden_dir, "edm_unet_forward")
    sd = formula_sd(cfg)
    x, t, y = (torch.from_numpy(g[k]) for k in ("x", "t", "y"))
    with torch.no_grad():
        ob = edm.unet_forward(sd, cfg, x, t, y=y, prec=Precision("bf16"))
    ref = torch.from_numpy(g["out"])
    assert float((ob - ref).norm() / ref.norm()) < 1.5e-2


# ---------------------------------------------------------------------------- host side of the product
def test_product_unet_state_dict_keys_match_reference(golden_dir):
    from models.cm.script_util import create_model_and_diffusion
    for tag, over in (("", {}), ("_plain", dict(class_cond=False, use_scale_shift_norm=False, resblock_updown=False))):
        kw = dict(image_size=16, class_cond=True, learn_sigma=False, num_channels=64, num_res_blocks=1, channel_mult="1,2",
                  num_heads=4, num_head_channels=64, num_heads_upsample=-1, attention_resolutions="8", dropout=0.0,
                  use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=True, use_fp16=False,
                  use_new_attention_order=False, weight_schedule="uniform")
        kw.update(over)
        net, diffusion = create_model_and_diffusion(**kw)
        g = load(golden_dir, f"edm_unet_forward{tag}")
        sd = net.state_dict()
        assert list(sd.keys()) == [str(k) for k in g["state_keys"]]
        for v, ref in zip(sd.values(), g["state_shapes"]):
            assert list(v.shape) == [int(s) for s in ref[:v.dim()]]
        assert diffusion.sigma_data == 0.5


def test_product_sampler_tables_match_reference(golden_dir):
    from models.cm.script_util import create_model_and_diffusion
    from models.DxMI.openai_diffusion import OpenAIDiffusion
    g = load(golden_dir, "edm_schedule")
    for name, kw in (("T10", dict(n_timesteps=10)), ("T4", dict(n_timesteps=4, stochastic_last=True, rho=4.0))):
        net, diffusion = create_model_and_diffusion(
            image_size=16, class_cond=True, learn_sigma=False, num_channels=64, num_res_blocks=1, channel_mult="1,2",
            num_heads=4, num_head_channels=64, num_heads_upsample=-1, attention_resolutions="8", dropout=0.0,
            use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=True, use_fp16=False,
            use_new_attention_order=False, weight_schedule="uniform")
        s = OpenAIDiffusion(net, diffusion, sample_shape=(3, 16, 16), class_cond=True, num_classes=1000,
                            trainable_beta="fix_last", **kw)
        np.testing.assert_array_equal(s.sigmas.numpy(), g[f"sigmas_{name}"])
        np.testing.assert_array_equal(s.sigma_up.numpy(), g[f"sigma_up_{name}"])
        np.testing.assert_array_equal(s.sigma_down.numpy(), g[f"sigma_down_{name}"])
        np.testing.assert_array_equal(net.log_betas.detach().numpy(), g[f"log_betas_{name}"])
        assert "log_betas" in dict(net.named_parameters())


def test_product_refuses_cpu_tensors():
    from dxmi_hip._lib import DxmiError
    from models.cm.unet import UNetModel
    net = UNetModel(image_size=16, in_channels=3, model_channels=64, out_channels=3, num_res_blocks=1,
                    attention_resolutions=(2,), channel_mult=(1, 2), num_head_channels=64)
    with pytest.raises(DxmiError):
        net(torch.zeros(1, 3, 16, 16), torch.zeros(1))


@pytest.mark.parametrize("tag,tb", [("fixlast3", "fix_last3"), ("allbeta", True)])
def test_edm_sample_step_learnable_sigma_variants(golden_dir, tag, tb):
    """OpenAIDiffusion(trainable_beta in {'fix_last3', True}) (openai_diffusion.py:76-84) at T = 6, log_betas off their initial
    values: the last three steps keep the ancestral sigma_up under 'fix_last3', none does under True."""
    cfg = VARIANTS[""]
    g = load(golden_dir, f"edm_sample_step_T6_{tag}")
    sd = formula_sd(cfg)
    sch = edm.EDMSchedule(6, stochastic_last=True, rho=4.0, trainable_beta=tb)
    lb = torch.from_numpy(g["log_betas"])
    idx = torch.from_numpy(g["idx"])
    with torch.no_grad():
        ds = edm.sample_step(oracle_model(sd, cfg), sch, torch.from_numpy(g["x"]), idx, torch.from_numpy(g["z"]), log_betas=lb,
                             y=torch.from_numpy(g["y"]))
    for k in ("sample", "mean", "sigma"):
        np.testing.assert_allclose(ds[k].numpy(), g[k], rtol=0, atol=1e-5, err_msg=k)
    learned = torch.exp(lb)[idx].clamp(1e-4, None).numpy()
    fixed = sch.sigma_up[idx].clamp(1e-4, None).numpy()
    uses_learned = np.isclose(g["sigma"], learned, rtol=1e-6)
    want = (idx.numpy() < 3) if tb == "fix_last3" else np.ones(len(idx), bool)
    assert (uses_learned == want).all() and np.isclose(g["sigma"][~want], fixed[~want], rtol=1e-6).all()
