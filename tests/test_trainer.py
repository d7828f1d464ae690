"""DxMI train step (rows a10-a12): oracle pinned to the reference's golden step (CPU), the product's
host-side buffer / index logic against the reference's literal expressions (CPU), and the HIP train
step against the same golden step (GPU)."""
import os

import numpy as np
import pytest
import torch

UNET_KW = dict(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
               in_channels=3, resolution=32)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def golden_logs(g, which):
    return dict(zip([str(k) for k in g[f"{which}_keys"]], [float(v) for v in g[f"{which}_vals"]]))


def build_models():
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    from oracle.weights import formula_tensor
    net = Model(**UNET_KW)
    sampler = VARSampler(net, 10, [3, 32, 32], trainable_beta="fix_last")
    net.load_state_dict({k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()})
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    v.load_state_dict({k: formula_tensor(k, t.shape) for k, t in v.state_dict().items()})
    return net, sampler, v


def test_oracle_trainer_step_matches_reference(golden_dir):
    from oracle import schedule as osched
    from oracle.trainer import OracleDxMI
    torch.set_num_threads(8)
    g = load(golden_dir, "trainer_step")
    B, T = int(g["B"]), int(g["T"])
    net, sampler, v = build_models()
    s = osched.var_schedule(T)
    sched = {k: torch.from_numpy(np.asarray(val, dtype=np.float32)) for k, val in s.items() if k != "user_defined_eta"}
    o = OracleDxMI({k: t.detach() for k, t in net.state_dict().items()}, {k: t.detach() for k, t in v.state_dict().items()},
                   sched, B, T, eta=s["user_defined_eta"])
    img = torch.from_numpy(g["img"])
    torch.manual_seed(int(g["seed"]))
    noise = [torch.randn(B, 3, 32, 32) for _ in range(T + 1)]
    d = o.sample(noise)
    buf = o.append_buffer(o.reset_buffer(), d)
    assert torch.equal(buf["timestep"], torch.from_numpy(g["buffer_timestep"]))       # INT path bit-exact
    assert abs(buf["state"].double().sum().item() - float(g["buffer_state_sum"])) < 1e-2
    le = o.update_f_v(img, d, buf)
    z = None
    perm_state = torch.get_rng_state()
    # update_sampler draws randperm then randn_like: reproduce the same order
    torch.set_rng_state(perm_state)
    ls = None

    class _Z:
        pass
    # draw z after the permutation, exactly as sample_step's randn_like does
    def run_sampler():
        import oracle.var_sampler as ovs
        orig = ovs.sample_step
        def patched(net_fn, sched_, lb, x, t, zz, **kw):
            return orig(net_fn, sched_, lb, x, t, torch.randn_like(x), **kw)
        ovs.sample_step = patched
        try:
            return o.update_sampler(buf, None)
        finally:
            ovs.sample_step = orig
    ls = run_sampler()
    ge, gs = golden_logs(g, "energy"), golden_logs(g, "sampler")
    assert list(le.keys()) == list(ge.keys()) and list(ls.keys()) == list(gs.keys())
    for k in ge:
        assert abs(le[k] - ge[k]) <= 2e-4 * max(1.0, abs(ge[k])), (k, le[k], ge[k])
    for k in gs:
        assert abs(ls[k] - gs[k]) <= 2e-4 * max(1.0, abs(gs[k])), (k, ls[k], gs[k])
    np.testing.assert_allclose(o.betas_for_q.numpy(), g["betas_for_q"], rtol=1e-5)
    np.testing.assert_allclose(o.net["log_betas"].detach().numpy(), g["log_betas_after"], rtol=1e-5, atol=1e-6)


def test_buffer_and_td_indexing_match_reference_expressions():
    """Product host logic (CPU): append_buffer layout and the single-gather TD row selection equal the
    reference's torch.cat-per-step layout and `buf[key][indices][train_indices]` double index."""
    from models.DxMI.trainer import append_buffer, reset_buffer
    from oracle.trainer import OracleDxMI
    g = torch.Generator().manual_seed(1)
    B, T = 5, 4
    d = {"l_sample": [torch.randn(B, 3, 4, 4, generator=g) for _ in range(T + 1)],
         "logp": [torch.randn(B, generator=g) for _ in range(T)], "control": [torch.randn(B, 3, 4, 4, generator=g) for _ in range(T)],
         "mean": [torch.randn(B, 3, 4, 4, generator=g) for _ in range(T)], "sigma": [torch.rand(B, 1, 1, 1, generator=g) for _ in range(T)]}
    buf = reset_buffer("cpu")
    for _ in range(2):   # two appends: rows accumulate
        buf = append_buffer(buf, d)
    ref = OracleDxMI.reset_buffer()
    for _ in range(2):
        ref = OracleDxMI.append_buffer(ref, d)
    for k in ("state", "next_state", "timestep", "logp", "control", "mean", "sigma"):
        assert torch.equal(buf[k], ref[k]), k
    assert buf["final"].shape == buf["state"].shape and torch.equal(buf["final"][:B], d["l_sample"][-1])
    torch.manual_seed(3)
    perm = torch.randperm(B * T)
    indices = perm + (buf["state"].shape[0] - B * T)
    ts_perm = buf["timestep"][indices]
    for update_t in range(T):
        train_indices = torch.nonzero(buf["timestep"][indices] == update_t).flatten()
        rows = indices[torch.nonzero(ts_perm == update_t).flatten()]
        assert torch.equal(buf["state"][rows], buf["state"][indices][train_indices])
        assert torch.equal(buf["sigma"][rows], buf["sigma"][indices][train_indices])
        assert torch.equal(buf["timestep"][rows], buf["timestep"][indices][train_indices])


@pytest.mark.gpu
def test_hip_trainer_step_vs_reference(golden_dir):
    """Full HIP train step at the reference's golden configuration.  bf16 activations/gradients against
    the reference's fp32: scalar statistics within 5e-2 relative (|.|<1: absolute), integer paths exact,
    parameter updates (Adam normalises the gradient, so an update is +-lr per element) in direction."""
    from models.DxMI.trainer import DxMI_Trainer, append_buffer, reset_buffer
    DEV = "cuda:0"
    g = load(golden_dir, "trainer_step")
    B, T = int(g["B"]), int(g["T"])
    net, sampler, v = build_models()
    sampler, v = sampler.to(DEV), v.to(DEV)
    params_not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
    opt = torch.optim.Adam([{"params": net.log_betas, "lr": 1e-5}, {"params": params_not_beta, "lr": 1e-7}])
    opt_v = torch.optim.Adam(v.parameters(), lr=1e-5)
    trainer = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99,
                           entropy_in_value=None, velocity_in_value=None, time_cost_sig=True, n_timesteps=T)
    trainer.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
    img = torch.from_numpy(g["img"]).to(DEV)
    torch.manual_seed(int(g["seed"]))
    noise = [torch.randn(B, 3, 32, 32) for _ in range(T + 1)]      # the reference's CPU draws, in its order
    sampler.eval()
    d = sampler.sample(B, device=DEV, noise=noise)
    buf = append_buffer(reset_buffer(DEV), d)
    assert torch.equal(buf["timestep"].cpu(), torch.from_numpy(g["buffer_timestep"]))
    assert [list(buf[k].shape) for k in ("state", "sigma", "logp")] == [[T * B, 3, 32, 32], [T * B, 1, 1, 1], [T * B]]
    v0 = v.net.conv1.weight.detach().clone()
    le = trainer.update_f_v(img, d, buf)
    orig = sampler.sample_step
    sampler.sample_step = lambda x, t, y=None: orig(x, t, noise=torch.randn(x.shape).to(x.device))  # CPU draw, reference order
    ls = trainer.update_sampler(buf, 1)
    ge, gs = golden_logs(g, "energy"), golden_logs(g, "sampler")
    assert list(le.keys()) == list(ge.keys()) and list(ls.keys()) == list(gs.keys())
    bad = []
    for got, ref in ((le, ge), (ls, gs)):
        for k in ref:
            if abs(got[k] - ref[k]) > 5e-2 * max(1.0, abs(ref[k])):
                bad.append((k, got[k], ref[k]))
    assert not bad, bad
    np.testing.assert_allclose(trainer.betas_for_q.cpu().numpy(), g["betas_for_q"], rtol=2e-3)
    # parameter updates: sign agreement of the Adam steps with the reference's
    dv = (v.net.conv1.weight.detach() - v0).cpu().numpy()
    ref_dv = g["value_conv1_w_delta"]
    agree = np.mean(np.sign(dv[np.abs(ref_dv) > 5e-5]) == np.sign(ref_dv[np.abs(ref_dv) > 5e-5]))
    assert agree > 0.9, agree
    lb = net.log_betas.detach().cpu().numpy()
    assert np.allclose(lb, g["log_betas_after"], atol=3e-5), (lb, g["log_betas_after"])


def test_oracle_sample_guidance_matches_reference(golden_dir):
    from oracle import schedule as osched
    from oracle.trainer import OracleDxMI
    torch.set_num_threads(8)
    g = load(golden_dir, "sample_guidance_T10")
    T, B = 10, 2
    net, sampler, v = build_models()
    s = osched.var_schedule(T)
    sched = {k: torch.from_numpy(np.asarray(val, dtype=np.float32)) for k, val in s.items() if k != "user_defined_eta"}
    o = OracleDxMI({k: t.detach() for k, t in net.state_dict().items()}, {k: t.detach() for k, t in v.state_dict().items()},
                   sched, B, T, eta=s["user_defined_eta"])
    torch.manual_seed(int(g["seed"]))
    zs = [torch.randn(B, 3, 32, 32) for _ in range(T)]
    d = o.sample_guidance(torch.from_numpy(g["x0"]), zs, float(g["scale"]))
    np.testing.assert_allclose(torch.stack(d["l_sample"]).numpy(), g["l_sample"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(torch.stack(d["guidance"]).numpy(), g["guidance"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(torch.stack(d["logp_on"]).numpy(), g["logp_on"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_hip_sample_guidance_vs_reference(golden_dir):
    """Value-guided sampling on the HIP path (fused transitions + value-net input gradient) against the reference."""
    from models.DxMI.trainer import DxMI_Trainer
    DEV = "cuda:0"
    g = load(golden_dir, "sample_guidance_T10")
    T, B = 10, 2
    net, sampler, v = build_models()
    sampler, v = sampler.to(DEV), v.to(DEV)
    trainer = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, n_timesteps=T)
    trainer.set_models(f=None, v=v, sampler=sampler, optimizer=None, optimizer_fstar=None, optimizer_v=None)
    torch.manual_seed(int(g["seed"]))
    zs = [torch.randn(B, 3, 32, 32).to(DEV) for _ in range(T)]
    sampler.eval()
    with torch.no_grad():
        d = trainer.sample_guidance(B, DEV, x0=torch.from_numpy(g["x0"]), guidance_scale=float(g["scale"]), noise=zs)
    ls = torch.stack(d["l_sample"]).cpu().numpy()
    assert ls.shape == g["l_sample"].shape
    rel = np.linalg.norm(ls - g["l_sample"]) / np.linalg.norm(g["l_sample"])
    gd = torch.stack(d["guidance"]).cpu().numpy()
    relg = np.linalg.norm(gd - g["guidance"]) / np.linalg.norm(g["guidance"])
    assert rel < 3e-2 and relg < 1e-1, (rel, relg)
    assert torch.stack(d["logp_on"]).shape == (T, B) and d["logp_on_traj"].shape == (B,)
