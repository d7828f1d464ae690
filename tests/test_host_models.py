"""Host-side contract of the reference-mirroring modules (CPU, no kernels): state-dict keys, parameter
counts, schedule tables (bit-exact vs the reference's golden tables), integer helpers, and that the
product refuses CPU tensors instead of silently falling back."""
import os

import numpy as np
import pytest
import torch

UNET_KW = dict(ch=128, out_ch=3, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
               in_channels=3, resolution=32)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("T", [10, 4])
def test_sampler_tables_and_keys(golden_dir, T):
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler, VAR_get_params
    g = load(golden_dir, "schedule")
    net = Model(**UNET_KW)  # ch_mult arrives as a list from YAML loaders
    s = VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last")
    assert sorted(net.state_dict().keys()) == list(g[f"T{T}_state_dict_keys"])
    assert len(net.state_dict()) == 330
    np.testing.assert_array_equal(s.user_defined_eta, g[f"T{T}_user_defined_eta"])
    for k in ("continuous_steps", "Gamma_bar", "x_prev_multiplier", "theta_multiplier", "std"):
        np.testing.assert_array_equal(getattr(s, k).numpy(), g[f"T{T}_{k}"], err_msg=k)
    np.testing.assert_array_equal(net.log_betas.detach().numpy(), g[f"T{T}_log_betas"])
    np.testing.assert_array_equal(net.std.numpy(), g[f"T{T}_std"])
    np.testing.assert_array_equal(s.diffusion_steps_list.numpy(), g[f"T{T}_continuous_steps"])
    xm, cm, std, dsl = VAR_get_params(s.diffusion_hyperparams, s.user_defined_eta, s.kappa, s.continuous_steps)
    np.testing.assert_array_equal(xm.numpy(), g[f"T{T}_x_prev_multiplier"])
    assert s.n_timesteps == T and s.trainable_beta == "fix_last" and list(s.sample_shape) == [3, 32, 32]
    assert "log_betas" in dict(net.named_parameters())  # optimizer param-group split by name (train_cifar10.py:286-290)


def test_ddpm_hyperparams(golden_dir):
    from models.DxMI.var_sampler import calc_diffusion_hyperparams, diffusion_config
    g = load(golden_dir, "schedule")
    dh = calc_diffusion_hyperparams(**diffusion_config)
    np.testing.assert_array_equal(dh["Beta"].numpy(), g["ddpm_Beta"])
    np.testing.assert_array_equal(dh["Alpha_bar"].numpy(), g["ddpm_Alpha_bar"])


def test_param_counts_and_init_parity():
    from models.DxMI.unet_small import Model
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    net = Model(**UNET_KW)
    assert sum(p.numel() for p in net.parameters()) == 35746307
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    assert sum(p.numel() for p in v.parameters()) == 5134595
    assert len(v.state_dict()) == 33
    assert "net.blocks.2.skip.0.weight" in v.state_dict() and "net.out_scale.bias" in v.state_dict()


def test_value_keys_match_reference(golden_dir):
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    assert sorted(v.state_dict().keys()) == list(load(golden_dir, "value_forward")["keys"])


def test_unsupported_value_configs_fail_loudly():
    from models.modules import IGEBMEncoderV2
    with pytest.raises(NotImplementedError):
        IGEBMEncoderV2(use_spectral_norm=True, keepdim=False)
    with pytest.raises(NotImplementedError):
        IGEBMEncoderV2(keepdim=True)


def test_integer_helpers():
    from models.diffusion import extract, make_beta_schedule
    from models.modules import process_single_t
    x = torch.zeros(5, 3, 4, 4)
    assert torch.equal(process_single_t(x, 3), torch.full((5,), 3, dtype=torch.long))
    assert torch.equal(process_single_t(x, torch.tensor(2)), torch.full((5,), 2, dtype=torch.long))
    tt = torch.tensor([4, 0, 1, 1, 3])
    assert process_single_t(x, tt) is tt
    betas = torch.arange(10, dtype=torch.float32) * 0.5
    out = extract(betas, torch.tensor([9, 0, 3, 3, 7]), x)
    assert out.shape == (5, 1, 1, 1) and torch.equal(out.flatten(), betas[[9, 0, 3, 3, 7]])
    assert torch.equal(make_beta_schedule("constant", 4, 1.0, 1.0), torch.ones(4))
    assert make_beta_schedule("linear", 10).shape == (10,)


def test_cpu_tensors_are_refused():
    from dxmi_hip import DxmiError
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    net = Model(**UNET_KW)
    s = VARSampler(net, 10, [3, 32, 32], trainable_beta="fix_last")
    with pytest.raises(DxmiError):
        with torch.no_grad():
            net(torch.zeros(1, 3, 32, 32), torch.zeros(1))
    with pytest.raises(DxmiError):
        s.sample(2, device="cpu")
