"""Pin the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; these run in the driver's `-m "not gpu"` pass."""
import os

import numpy as np
import pytest
import torch

from oracle import Precision
from oracle import schedule as osched
from oracle import unet_small as ounet
from oracle import value as ovalue
from oracle import var_sampler as ovs
from oracle.weights import formula_tensor

torch.set_num_threads(8)

UNET_SHAPES = None


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def unet_state_dict(keys, T):
    """Formula weights for the reference key list; shapes come from our own host module."""
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
                in_channels=3, resolution=32)
    VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last")
    sd = net.state_dict()
    assert sorted(sd.keys()) == list(keys)
    return {k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in sd.items()}


# ---------------------------------------------------------------------------- schedule (a1)
@pytest.mark.parametrize("T", [10, 4])
def test_schedule_tables_match_reference(golden_dir, T):
    g = load(golden_dir, "schedule")
    s = osched.var_schedule(T)
    np.testing.assert_allclose(s["user_defined_eta"], g[f"T{T}_user_defined_eta"], rtol=0, atol=0)
    for k in ("continuous_steps", "Gamma_bar", "x_prev_multiplier", "theta_multiplier", "std"):
        assert s[k].dtype == np.float32
        np.testing.assert_array_equal(s[k], g[f"T{T}_{k}"], err_msg=k)  # bit-exact float32 tables
    # log() goes through a different libm: 1 ulp
    np.testing.assert_allclose(s["log_betas"], g[f"T{T}_log_betas"], rtol=2e-7, atol=0)


def test_schedule_known_answers():
    """The only known answers the reference itself prints: eta for T=10, models/DxMI/trainer.py:148-149."""
    eta = osched.var_noise(10)
    ref = [1.0e-4, 1.1025e-2, 4.0e-2, 8.7025e-2, 1.521e-1, 2.35225e-1, 3.364e-1, 4.55625e-1, 5.929e-1, 7.48225e-1]
    np.testing.assert_allclose(eta, ref, rtol=1e-5)


def test_ddpm_tables(golden_dir):
    g = load(golden_dir, "schedule")
    beta, _, abar = osched.ddpm_tables()
    np.testing.assert_array_equal(beta, g["ddpm_Beta"])
    np.testing.assert_array_equal(abar, g["ddpm_Alpha_bar"])


# ---------------------------------------------------------------------------- U-Net (a5)
def test_unet_forward_matches_reference(golden_dir):
    g = load(golden_dir, "unet_small_forward")
    keys = load(golden_dir, "schedule")["T10_state_dict_keys"]
    sd = unet_state_dict(keys, 10)
    assert sum(v.numel() for k, v in sd.items() if k not in ("log_betas", "std")) == int(g["n_params"]) == 35746307
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    np.testing.assert_allclose(ounet.timestep_embedding_sincos(t, 128).numpy(), g["temb_sinusoid"], rtol=0, atol=1e-6)
    with torch.no_grad():
        y = ounet.forward(sd, ounet.UNetSmallConfig(), x, t)
    ref = torch.from_numpy(g["y"])
    rel = ((y - ref).norm() / ref.norm()).item()
    assert rel < 1e-5, rel
    # and the bf16 storage model stays within the tolerance stated for the HIP path
    with torch.no_grad():
        yb = ounet.forward(sd, ounet.UNetSmallConfig(), x, t, Precision("bf16"))
    relb = ((yb - ref).norm() / ref.norm()).item()
    assert relb < 1e-2, relb


# ---------------------------------------------------------------------------- sampler (a2, a3)
def _sched_t(T):
    s = osched.var_schedule(T)
    return {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in s.items() if k != "user_defined_eta"}


@pytest.mark.parametrize("T", [10, 4])
def test_var_sampling_matches_reference(golden_dir, T):
    g = load(golden_dir, f"var_sampling_T{T}")
    keys = load(golden_dir, "schedule")[f"T{T}_state_dict_keys"]
    sd = unet_state_dict(keys, T)
    sched = _sched_t(T)
    B = int(g["B"])
    noise = list(torch.from_numpy(g["noise"]))      # the reference run's own draws, stored with the fixture
    cfg = ounet.UNetSmallConfig()
    with torch.no_grad():
        d = ovs.sample(lambda x, t: ounet.forward(sd, cfg, x, t), sched, sched["log_betas"], noise)
    for key in ("l_sample", "mean", "control", "sigma", "logp"):
        got = torch.stack(d[key]).numpy()
        ref = g[key]
        assert got.shape == ref.shape, key
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 2e-5, (key, rel)
    np.testing.assert_allclose(d["sample"].numpy(), g["sample"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("name,tb", [("sample_step_T10", "fix_last"), ("sample_step_T10_fixedbeta", False)])
def test_sample_step_matches_reference(golden_dir, name, tb):
    g = load(golden_dir, name)
    keys = load(golden_dir, "schedule")["T10_state_dict_keys"]
    sd = unet_state_dict(keys, 10)
    sched = _sched_t(10)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    z = torch.from_numpy(g["z"])                    # the reference run's own draw, stored with the fixture
    cfg = ounet.UNetSmallConfig()
    with torch.no_grad():
        d = ovs.sample_step(lambda xx, tt: ounet.forward(sd, cfg, xx, tt), sched, sched["log_betas"], x, t, z,
                            trainable_beta=tb)
    for key in ("sample", "mean", "control", "sigma", "entropy", "logp"):
        got, ref = d[key].numpy(), g[key]
        assert got.shape == ref.shape, key
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 2e-5, (key, rel)


# ---------------------------------------------------------------------------- value net (a9)
def value_state_dict(keys):
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    sd = v.state_dict()
    assert sorted(sd.keys()) == list(keys)
    return {k: formula_tensor(k, t.shape) for k, t in sd.items()}


def test_value_forward_and_grads_match_reference(golden_dir):
    g = load(golden_dir, "value_forward")
    sd = value_state_dict(g["keys"])
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"]) == 5134595
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    out = ovalue.forward(leaves, x)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=2e-5, atol=1e-5)
    out.sum().backward()
    checks = {"grad_x": x.grad, "grad_conv1_w": leaves["net.conv1.weight"].grad,
              "grad_b5_conv2_w": leaves["net.blocks.5.conv2.weight"].grad[:4],
              "grad_b2_skip_w": leaves["net.blocks.2.skip.0.weight"].grad,
              "grad_linear_w": leaves["net.linear.weight"].grad,
              "grad_out_scale_w": leaves["net.out_scale.weight"].grad, "grad_out_scale_b": leaves["net.out_scale.bias"].grad}
    for k, got in checks.items():
        ref = g[k]
        rel = np.linalg.norm(got.numpy() - ref) / np.linalg.norm(ref)
        assert rel < 2e-5, (k, rel)
    g64 = load(golden_dir, "value_forward_64")
    with torch.no_grad():
        out64 = ovalue.forward(sd, torch.from_numpy(g64["x"]))
    np.testing.assert_allclose(out64.numpy(), g64["out"], rtol=2e-5, atol=1e-5)


# ---------------------------------------------------------------------------- log_prob_step (a4)
def test_log_prob_step_matches_reference(golden_dir):
    """VARSampler.log_prob_step (var_sampler.py:431-444): value, and the gradient THROUGH the net that the reference's
    un-detached call builds (w.r.t. x_prev and a spread of parameters)."""
    torch.set_num_threads(8)
    g = load(golden_dir, "log_prob_step_T10")
    keys = load(golden_dir, "schedule")["T10_state_dict_keys"]
    sd = {k: v.clone().requires_grad_(k not in ("std", "log_betas")) for k, v in unet_state_dict(keys, 10).items()}
    s = osched.var_schedule(10)
    sched = {k: torch.from_numpy(v) for k, v in s.items() if k != "user_defined_eta"}
    x_prev = torch.from_numpy(g["x_prev"]).requires_grad_(True)
    lp = ovs.log_prob_step(lambda x, t: ounet.forward(sd, ounet.UNetSmallConfig(), x, t), sched, x_prev,
                           torch.from_numpy(g["x_next"]), torch.from_numpy(g["t"]))
    np.testing.assert_allclose(lp.detach().numpy(), g["log_prob"], rtol=2e-5)
    lp.sum().backward()
    ref = g["grad_x_prev"]
    assert np.linalg.norm(x_prev.grad.numpy() - ref) <= 2e-4 * np.linalg.norm(ref)
    for i, (k, rows) in enumerate(zip(g["grad_keys"], g["grad_rows"])):
        got = sd[str(k)].grad.numpy()
        got = got if rows < 0 else got[:rows]
        ref = g[f"grad_{i}"]
        assert np.linalg.norm(got - ref) <= 3e-4 * np.linalg.norm(ref), k


def test_sample_step_all_learnable_sigma_matches_reference(golden_dir):
    """VARSampler(trainable_beta=True): every step's sigma from net.log_betas, the last one included (var_sampler.py:383-394);
    the fixture's log_betas are off their initial values."""
    g = load(golden_dir, "sample_step_T10_allbeta")
    keys = load(golden_dir, "schedule")["T10_state_dict_keys"]
    sd = unet_state_dict(keys, 10)
    sched = _sched_t(10)
    lb = torch.from_numpy(g["log_betas"])
    assert not torch.allclose(lb, sched["log_betas"])
    cfg = ounet.UNetSmallConfig()
    with torch.no_grad():
        d = ovs.sample_step(lambda xx, tt: ounet.forward(sd, cfg, xx, tt), sched, lb, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]),
                            torch.from_numpy(g["z"]), trainable_beta=True)
    for key in ("sample", "mean", "control", "sigma", "entropy", "logp"):
        got, ref = d[key].numpy(), g[key]
        assert got.shape == ref.shape, key
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-5, key
    np.testing.assert_allclose(d["sigma"].flatten().numpy(), np.exp(g["log_betas"])[g["t"]], rtol=1e-6)
