"""Config-level coverage on the GPU: every BASELINE.json configuration runs through the HIP path at its named size.

  C2  CIFAR-10 DDPM T=10, batch 256: one full DxMI train step (finite, INT buffer exact, reproducible bit for bit)
  C3  CIFAR-10 "DDGAN protocol" T=4, 128 images / rank (512 over 4 GPUs), value_resample=True: full train step at size
      (the reference-golden parity of this branch is tests/test_trainer.py::test_hip_trainer_step_vs_reference[..T4_resample])
  C4  ImageNet-64 EDM T=10, batch 100 / rank, class-conditional: OpenAIDiffusion.sample at full network size
  C5  LSUN-256 EDM T=4, batch 16 / rank: OpenAIDiffusion.sample at full size + the first 256x256 level against the oracle
  a4  VARSampler.log_prob_step against the reference's golden values and gradients
Size-independent properties used where the CPU oracle cannot follow: determinism, batch independence
(an image's trajectory does not depend on its batch mates), INT paths exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _cifar(T, dropout=0.1):
    import configs_builtin
    import dxmi_config
    cfg = configs_builtin.get("cifar10_T10")
    cfg.sampler_net["dropout"] = dropout
    cfg.sampler["n_timesteps"] = T
    torch.manual_seed(0)
    net = dxmi_config.instantiate(cfg.sampler_net)
    sampler = dxmi_config.instantiate(cfg.sampler, net=net).to(DEV)
    v = dxmi_config.instantiate(cfg.value).to(DEV)
    return net, sampler, v


def _trainer(net, sampler, v, B, T, **kw):
    from dxmi_hip.optim import Adam
    from models.DxMI.trainer import DxMI_Trainer
    not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
    opt = Adam([{"params": net.log_betas, "lr": 1e-5}, {"params": not_beta, "lr": 1e-7}])
    opt_v = Adam(v.parameters(), lr=1e-5)
    tr = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99,
                      entropy_in_value=None, velocity_in_value=None, time_cost_sig=1, n_timesteps=T, **kw)
    tr.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
    return tr


def _full_step(B, T, **kw):
    """train_cifar10.py:162-193 (n_critic = n_generator = 1) with injected noise; returns logs + a parameter checksum."""
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import append_buffer
    net, sampler, v = _cifar(T)
    net.dropout_seed = 99
    tr = _trainer(net, sampler, v, B, T, **kw)
    g = torch.Generator(device=DEV).manual_seed(17)
    img = torch.rand(B, 3, 32, 32, device=DEV, generator=g) * 2 - 1
    noise = [torch.randn(B, 3, 32, 32, device=DEV, generator=g) for _ in range(T + 1)]
    zs = [torch.randn(B, 3, 32, 32, device=DEV, generator=g) for _ in range(T + 1)]
    ring = TransitionRing(1, T, B, (3, 32, 32), DEV)
    sampler.eval()
    d = sampler.sample(B, device=DEV, noise=noise, out=ring.next_slot())
    buf = append_buffer(ring, d)
    assert torch.equal(buf.gather("timestep", torch.arange(T * B, device=DEV)), torch.arange(T, device=DEV).repeat_interleave(B))
    torch.manual_seed(5)                        # randperm: torch's CPU generator, as in the reference
    orig, k = sampler.sample_step, [0]

    def step_with_noise(x, t, y=None):
        k[0] += 1
        return orig(x, t, noise=zs[(k[0] - 1) % len(zs)])
    sampler.sample_step = step_with_noise
    le = tr.update_f_v(img, d, buf)
    ls = tr.update_sampler(buf, 1)
    assert k[0] == (T + 1 if kw.get("value_resample") else 1)
    chk = sum(p.detach().double().sum().item() for p in list(net.parameters()) + list(v.parameters()))
    return le, ls, chk


def test_c2_train_step_batch256_T10():
    """BASELINE configs[1] train leg at its size: B=256, T=10, dropout 0.1 live in the policy step."""
    le, ls, chk = _full_step(256, 10)
    assert all(np.isfinite(x) for x in list(le.values()) + list(ls.values()))
    assert len([k for k in le if k.startswith("value/step_")]) == 10 and len([k for k in ls if k.startswith("sigma/")]) == 10
    le2, ls2, chk2 = _full_step(256, 10)
    assert le == le2 and ls == ls2 and chk == chk2            # reproducible bit for bit (fixed-order reductions)


def test_c3_train_step_batch128_T4_value_resample():
    """BASELINE configs[2] protocol (configs/cifar10/T4_ddgan.yaml: T=4, batch 512 / 4 GPUs = 128 per rank,
    value_resample: True) on the DDPM backbone (models.ddgan is absent from the reference snapshot)."""
    le, ls, chk = _full_step(128, 4, value_resample=True)
    assert all(np.isfinite(x) for x in list(le.values()) + list(ls.values()))
    assert [k for k in le if k.startswith("value/step_")] == [f"value/step_{t}_" for t in (3, 2, 1, 0)]
    le2, ls2, chk2 = _full_step(128, 4, value_resample=True)
    assert le == le2 and ls == ls2 and chk == chk2


def test_a4_log_prob_step_vs_reference(golden_dir):
    """VARSampler.log_prob_step: value against the reference golden, and the gradient that flows THROUGH the net
    (the reference does not detach it, var_sampler.py:189-200) w.r.t. x_prev and a spread of parameters."""
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    from oracle.weights import formula_tensor
    g = np.load(os.path.join(golden_dir, "log_prob_step_T10.npz"))
    net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1, in_channels=3, resolution=32)
    sampler = VARSampler(net, 10, [3, 32, 32], trainable_beta="fix_last")
    net.load_state_dict({k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()})
    sampler = sampler.to(DEV).eval()
    x_prev = torch.from_numpy(g["x_prev"]).to(DEV).requires_grad_(True)
    lp = sampler.log_prob_step(x_prev, torch.from_numpy(g["x_next"]).to(DEV), torch.from_numpy(g["t"]).to(DEV))
    assert lp.requires_grad and lp.shape == (6,)
    # log N(x'; mean, sigma) with sigma down to 1e-3 amplifies the bf16 error of eps by c/sigma: compare on the scale of |lp|
    np.testing.assert_allclose(lp.detach().cpu().numpy(), g["log_prob"], rtol=3e-2, atol=0.5)
    lp.sum().backward()
    assert _cos(x_prev.grad.cpu().numpy(), g["grad_x_prev"]) > 0.99
    named = dict(net.named_parameters())
    lp_err = np.abs(lp.detach().cpu().numpy() - g["log_prob"]) / np.maximum(np.abs(g["log_prob"]), 0.5 / 3e-2)
    print(f"log_prob_step: worst sample {int(lp_err.argmax())} at {lp_err.max():.3e} of max(|lp|, 16.7) (bound 3e-2); "
          f"grad x_prev cosine {_cos(x_prev.grad.cpu().numpy(), g['grad_x_prev']):.5f}")
    worst = (None, 2.0)
    for i, (k, rows) in enumerate(zip(g["grad_keys"], g["grad_rows"])):
        got = named[str(k)].grad.cpu().numpy()
        got = got if rows < 0 else got[:rows]
        c = _cos(got, g[f"grad_{i}"])
        worst = (str(k), c) if c < worst[1] else worst
        assert c > 0.99, (k, c)
    print(f"log_prob_step WORST parameter gradient: {worst[0]} cosine {worst[1]:.5f} (bound 0.99)")
    with pytest.raises(IndexError):
        sampler.log_prob_step(x_prev.detach(), x_prev.detach(), torch.tensor([10] * 6, device=DEV))
    with pytest.raises(IndexError):
        sampler.sample_step(x_prev.detach(), torch.tensor([0, 1, 2, 3, 4, 10]))          # host-resident t: checked eagerly
    with torch.no_grad():                                                                 # device t: poisoned, not OOB
        bad = sampler.sample_step(x_prev.detach(), torch.tensor([0, 1, 2, 3, 4, 10], device=DEV))
    assert torch.isnan(bad["sample"][5]).all() and torch.isfinite(bad["sample"][:5]).all()


def _edm(name):
    import configs_builtin
    from models.cm.script_util import create_model_and_diffusion
    from models.DxMI.openai_diffusion import OpenAIDiffusion
    from oracle.weights import formula_tensor
    cfg = configs_builtin.CONFIGS[name]
    net, diffusion = create_model_and_diffusion(**dict(cfg["diffusion"]))
    sd = {k: formula_tensor(k, v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    s = OpenAIDiffusion(net, diffusion, **dict(cfg["sampler"]))
    net.to(DEV).eval()
    return net, s, sd


def test_c4_imagenet64_T10_batch100_sample():
    """BASELINE configs[3]: full-size ImageNet-64 EDM net (295.9M parameters), T=10, 100 images per rank, class-conditional,
    through OpenAIDiffusion.sample: finite, reproducible, BITWISE batch-independent (round 3: the GroupNorm statistics of an
    image no longer depend on the batch — block statistics per 128-pixel half tile from the conv epilogue, batch-independent
    chunking in the generic path), sigma ladder exact."""
    net, s, _ = _edm("imagenet64_T10")
    assert sum(p.numel() for p in net.parameters()) == 295_899_267 + 10       # + log_betas[T]
    B, T = 100, 10
    g = torch.Generator(device=DEV).manual_seed(64)
    noise = torch.randn(T + 1, B, 3, 64, 64, device=DEV, generator=g)
    y = torch.randint(0, 1000, (B,), device=DEV, generator=g)
    with torch.no_grad():
        d = s.sample(B, device=DEV, i_class=y, noise=noise)
        d2 = s.sample(B, device=DEV, i_class=y, noise=noise)
        part = s.sample(7, device=DEV, i_class=y[40:47].contiguous(), noise=noise[:, 40:47].contiguous())
    assert len(d["l_sample"]) == T + 1 and d["sample"].shape == (B, 3, 64, 64) and torch.isfinite(d["sample"]).all()
    assert torch.equal(d["sample"], d2["sample"])
    assert torch.equal(part["sample"], d["sample"][40:47])
    assert torch.equal(d["y"], y)
    np.testing.assert_allclose(torch.stack(d["sigma"])[:, 0].cpu().numpy()[:-1],
                               np.array([35.9629364, 18.3089771, 8.63025856, 3.69350600, 1.39556587, 0.446370661, 0.112989359,
                                         0.0201192517, 0.00199039816], np.float32), rtol=2e-6)     # SURVEY 8c sigma_up table


def test_c5_lsun256_T4_batch16_sample_and_first_level_vs_oracle():
    """BASELINE configs[4]: full-size LSUN-256 EDM net (526.3M parameters; configs/lsun/T4.yaml: 2 res blocks, additive
    embedding, unconditional), T=4 rho=4 stochastic_last, 16 images per rank at 256x256, through OpenAIDiffusion.sample.
    Oracle comparison where the CPU can follow: the first resolution level (input conv + the two 256x256 ResBlocks + the
    down ResBlock, 256 channels) of two of the 16 images against the oracle's bf16 storage model."""
    from oracle import Precision, edm
    net, s, sd = _edm("lsun_bedroom_T4")
    assert sum(p.numel() for p in net.parameters()) == 526_304_771 + 4
    B, T = 16, 4
    g = torch.Generator(device=DEV).manual_seed(256)
    noise = torch.randn(T + 1, B, 3, 256, 256, device=DEV, generator=g)
    with torch.no_grad():
        d = s.sample(B, device=DEV, noise=noise)
        d2 = s.sample(B, device=DEV, noise=noise)
        part = s.sample(2, device=DEV, noise=noise[:, 5:7].contiguous())
    assert d["sample"].shape == (B, 3, 256, 256) and torch.isfinite(d["sample"]).all() and d["y"] is None
    assert torch.equal(d["sample"], d2["sample"])
    assert torch.equal(part["sample"], d["sample"][5:7])                       # bitwise batch-independent
    np.testing.assert_allclose(torch.stack(d["sigma"])[:, 0].cpu().numpy(),
                               np.array([26.0551224, 6.38497114, 0.671041727, 1.99999753e-3], np.float32), rtol=2e-6)
    # first level vs oracle (B=2 of the batch, network input of step 0: c_in * x_T)
    x = (noise[0, 5:7] * 80.0)
    sig = torch.full((2,), 80.0000076, device=DEV)
    c_in = 1 / (sig ** 2 + 0.25) ** 0.5
    t = 250 * torch.log(sig)
    trace = []
    with torch.no_grad():
        net.forward_inference(x * c_in[:, None, None, None], t, None, trace=trace)
    got = {n: v for n, v in trace}
    cfg = edm.EDMConfig(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions=(8, 16, 32),
                        channel_mult=(1, 1, 2, 2, 4, 4), num_classes=None, use_scale_shift_norm=False, resblock_updown=True)
    inp, _, _, _ = edm.plan(cfg)
    prec = Precision("bf16")
    xc, tc = (x * c_in[:, None, None, None]).cpu(), t.cpu()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    with torch.no_grad():
        emb = edm.timestep_embedding(tc, 256)
        emb = torch.nn.functional.linear(prec.act(emb), prec.w(sd["time_embed.0.weight"]), sd["time_embed.0.bias"])
        emb = torch.nn.functional.linear(prec.act(torch.nn.functional.silu(emb)), prec.w(sd["time_embed.2.weight"]), sd["time_embed.2.bias"])
        h = xc
        for pre, layers in inp[:4]:
            h = edm._layers(sd, pre, layers, h, emb, cfg, prec)
            r = rel_l2(got[pre].float().cpu().permute(0, 3, 1, 2), h)
            print(f"lsun256 {pre}: {tuple(h.shape)} rel-L2 vs oracle bf16-model {r:.3e}")
            assert r < 1.2e-2, (pre, r)
