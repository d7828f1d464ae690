"""End-to-end parity of the HIP path (through the C-ABI) against (a) golden vectors produced by the
reference, (b) the pinned oracle on the same seeded inputs, and (c) size-independent properties at
BASELINE.json's full batch (256).

Tolerances (stated per SURVEY 7): the HIP path stores activations/weights in bf16 and accumulates in
fp32; the reference is fp32.
  * vs reference golden (fp32):       rel-L2 <= 1e-2 per U-Net forward, <= 3e-2 on `sample` after T steps
  * vs oracle with the bf16 storage model (same rounding points): per op >= 99.9 % of bf16 outputs
    bit-identical; whole net: the bf16 noise floor (see test_unet_forward_vs_reference_and_oracle)
  * integer index path (schedule gathers): bit-exact
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
UNET_KW = dict(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
               in_channels=3, resolution=32)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def make_sampler(T, trainable_beta="fix_last"):
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    from oracle.weights import formula_tensor
    net = Model(**UNET_KW)
    s = VARSampler(net, T, [3, 32, 32], trainable_beta=trainable_beta)
    sd = {k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    return s.to(DEV).eval(), sd


@pytest.fixture(scope="module")
def sampler10():
    return make_sampler(10)


def test_unet_forward_vs_reference_and_oracle(golden_dir, sampler10):
    """Full forward.  bf16 storage makes ANY two implementations of this 35-layer net decorrelate to
    the bf16 noise floor after a few layers (a 1-ulp fp32 accumulation-order difference flips a bf16
    rounding with p ~ 1e-3, and each flip perturbs ~1e3 downstream outputs), so the full-net bound
    is the floor itself (measured 8.7e-3 for the oracle's bf16 model vs fp32 too); the sharp checks
    are per op and per block, below."""
    from oracle import Precision
    from oracle import unet_small as ounet
    s, sd = sampler10
    g = load(golden_dir, "unet_small_forward")
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    trace_h, trace_o = [], []
    with torch.no_grad():
        y = s.net.forward_inference(x.to(DEV), t.to(DEV), trace=trace_h).cpu()
        yo = ounet.forward(sd, ounet.UNetSmallConfig(), x, t, Precision("bf16"), trace=trace_o)
    assert y.shape == (2, 3, 32, 32) and y.dtype == torch.float32
    r_ref, r_orc = rel_l2(y, g["y"]), rel_l2(y, yo)
    print(f"unet forward: rel-L2 vs reference fp32 {r_ref:.3e}, vs oracle bf16-model {r_orc:.3e}")
    assert r_ref < 1e-2, r_ref             # the bound the docstring / SURVEY 7 state (measured 8.7e-3)
    assert r_orc < 1e-2, r_orc
    th, to = dict(trace_h), dict(trace_o)
    nchw = lambda a: a.float().cpu().permute(0, 3, 1, 2)
    # temb MLP (fp32 rows): accumulation-order noise only
    assert rel_l2(th["s_temb"].cpu(), to["s_temb"]) < 1e-5
    # image conv: every bf16 output bit-identical to the bf16 model
    assert torch.equal(nchw(th["conv_in"]), to["conv_in"])
    # one ResnetBlock (2 GN, 2 conv, temb add, residual) and the first downsample
    assert rel_l2(nchw(th["down.0.block.0"]), to["down.0.block.0"]) < 1.5e-3
    assert rel_l2(nchw(th["down.0.block.1"]), to["down.0.block.1"]) < 3e-3


def test_resnet_and_attn_block_ops_vs_bf16_model(sampler10):
    """Each fused op of a ResnetBlock / AttnBlock, fed the SAME bf16 input as the oracle's bf16 model:
    >= 99.9 % of the bf16 outputs bit-identical, rel-L2 <= 1e-4 (attention: P is rounded against a
    running max instead of the final max, so 5e-3)."""
    import torch.nn.functional as F
    from oracle import Precision
    from oracle import unet_small as ounet
    from dxmi_hip import ops
    s, sd = sampler10
    net, prec = s.net, Precision("bf16")
    pk = net.packed()
    nhwc = lambda a: a.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    nchw = lambda a: a.float().cpu().permute(0, 3, 1, 2).contiguous()

    def check(name, got, ref, min_same=0.999, max_rel=1e-4):
        same = (got == ref).float().mean().item()
        r = rel_l2(got, ref)
        print(f"  {name}: identical {100 * same:.3f}% rel-L2 {r:.2e}")
        assert same >= min_same and r <= max_rel, (name, same, r)

    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        blk, pre = net.down[1].block[0], "down.1.block.0"      # 128 -> 256 @16x16, has nin_shortcut
        x = prec.act(torch.randn(3, 128, 16, 16, generator=g))
        tp = torch.randn(3, 256, generator=g)
        a1 = prec.act(ounet.swish(ounet._gn(sd, pre + ".norm1", x)))
        check("norm1+silu", nchw(ops.groupnorm_silu(nhwc(x), blk.norm1.weight, blk.norm1.bias)), a1)
        h1 = prec.act(ounet._conv(sd, pre + ".conv1", a1, prec, padding=1) + tp[:, :, None, None])
        check("conv1+bias+temb", nchw(ops.conv2d(nhwc(a1), pk[id(blk), "conv1"], bias=blk.conv1.bias, addvec=tp.to(DEV))), h1)
        sc = prec.act(ounet._conv(sd, pre + ".nin_shortcut", x, prec))
        check("nin_shortcut", nchw(ops.conv2d(nhwc(x), pk[id(blk), "short"], bias=blk.nin_shortcut.bias)), sc)
        a2 = prec.act(ounet.swish(ounet._gn(sd, pre + ".norm2", h1)))
        out = prec.act(sc + ounet._conv(sd, pre + ".conv2", a2, prec, padding=1))
        check("conv2+bias+residual", nchw(ops.conv2d(nhwc(a2), pk[id(blk), "conv2"], bias=blk.conv2.bias, residual=nhwc(sc))), out)
        # attention block @16x16, C=256
        att, pre = net.down[1].attn[0], "down.1.attn.0"
        xa = prec.act(torch.randn(2, 256, 16, 16, generator=g))
        ref = ounet.attn_block(sd, pre, xa, prec)
        got = nchw(net._attn(pk, att, nhwc(xa))[0])          # (_attn returns (out, block statistics))
        check("attn block", got, ref, min_same=0.9, max_rel=5e-3)
        # stride-2 downsample and upsample convs
        ds = net.down[0].downsample
        xd = prec.act(torch.randn(2, 128, 32, 32, generator=g))
        refd = prec.act(ounet._conv(sd, "down.0.downsample.conv", F.pad(xd, (0, 1, 0, 1)), prec, stride=2))
        check("downsample", nchw(ops.conv2d(nhwc(xd), pk[id(ds), "conv"], bias=ds.conv.bias, stride=2, pad=0, pad_br=1)), refd)
        us = net.up[1].upsample
        xu = prec.act(torch.randn(2, 256, 16, 16, generator=g))
        refu = prec.act(ounet._conv(sd, "up.1.upsample.conv", F.interpolate(xu, scale_factor=2.0, mode="nearest"), prec, padding=1))
        check("upsample", nchw(ops.conv2d(nhwc(xu), pk[id(us), "conv"], bias=us.conv.bias, upsample=True)), refu)


@pytest.mark.parametrize("T", [10, 4])
def test_var_sampling_vs_reference(golden_dir, T):
    s, sd = make_sampler(T)
    g = load(golden_dir, f"var_sampling_T{T}")
    B = int(g["B"])
    noise = list(torch.from_numpy(g["noise"]))      # the reference run's own draws, stored with the fixture
    d = s.sample(B, device=DEV, noise=noise)
    assert len(d["l_sample"]) == T + 1 and len(d["mean"]) == T and d["sigma"][0].shape == (B, 1, 1, 1)
    errs = {k: rel_l2(torch.stack(d[k]).cpu(), g[k]) for k in ("l_sample", "mean", "control", "sigma", "logp")}
    errs["sample"] = rel_l2(d["sample"].cpu(), g["sample"])
    print(f"VAR_sampling T={T}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["sample"] < 3e-2 and errs["l_sample"] < 3e-2 and errs["mean"] < 3e-2
    assert errs["sigma"] < 1e-6          # exp(log_betas): fp32 table path
    assert errs["logp"] < 1e-4           # depends only on sigma and z (x' - mean = sigma z)
    assert torch.equal(d["logp_terminal"].cpu(), torch.zeros(B))


def test_sample_timestep_branch_table_is_bitwise_neutral():
    """sample() evaluates the U-Net's timestep branch (embedding, dense layers, every temb_proj) once on the T distinct
    timesteps and hands each step its row (row stride 0 over the batch) instead of recomputing B identical rows per step:
    the trajectory must be bitwise the one the per-step evaluation gives."""
    s, _ = make_sampler(10)
    B = 6
    torch.manual_seed(3)
    noise = [torch.randn(B, 3, 32, 32) for _ in range(11)]
    assert s.temb_table
    a = s.sample(B, device=DEV, noise=noise)
    s.temb_table = False
    b = s.sample(B, device=DEV, noise=noise)
    s.temb_table = True
    for k in ("l_sample", "mean", "control", "logp"):
        assert torch.equal(torch.stack(a[k]), torch.stack(b[k])), k
    # and the table rows are the rows forward() computes for a batch at that timestep
    net = s._bare_net()
    tau = s.continuous_steps[:10].to(DEV).float().contiguous()
    table = net.temb_table(tau)
    x = noise[0].to(DEV)
    e1 = net(x, tau[3].expand(B).contiguous())
    e2 = net(x, tau[3].expand(B).contiguous(), temb_rows=table[3:4])
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("name,tb", [("sample_step_T10", "fix_last"), ("sample_step_T10_fixedbeta", False)])
def test_sample_step_vs_reference(golden_dir, name, tb):
    s, sd = make_sampler(10, tb)
    g = load(golden_dir, name)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    z = torch.from_numpy(g["z"])                    # the reference run's own draw, stored with the fixture
    with torch.no_grad():
        d = s.sample_step(x.to(DEV), t.to(DEV), noise=z.to(DEV))
    for k in ("sample", "mean", "control"):
        assert d[k].shape == g[k].shape
        assert rel_l2(d[k].cpu(), g[k]) < 1e-2, k
    for k in ("sigma", "entropy"):
        assert d[k].shape == g[k].shape
        assert rel_l2(d[k].cpu(), g[k]) < 1e-6, k   # integer-gathered tables, fp32
    assert rel_l2(d["logp"].cpu(), g["logp"]) < 1e-4
    # scalar t is broadcast (models/modules.py:183-186)
    with torch.no_grad():
        d2 = s.sample_step(x.to(DEV), 3, noise=z.to(DEV))
    assert torch.allclose(d2["sigma"].flatten(), d["sigma"].flatten()[2].expand(6))


def test_sample_step_all_learnable_sigma_vs_reference(golden_dir):
    """VARSampler(trainable_beta=True): every sigma from net.log_betas, the last step included (reference var_sampler.py:383-394),
    with log_betas moved off their initial values — both the fused no-grad transition and the autograd one."""
    s, _ = make_sampler(10, True)
    g = load(golden_dir, "sample_step_T10_allbeta")
    with torch.no_grad():
        s.net.log_betas.copy_(torch.from_numpy(g["log_betas"]))
    x, t, z = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "z"))
    with torch.no_grad():
        d = s.sample_step(x, t, noise=z)
    dg = s.sample_step(x, t, noise=z)              # parameters ask for gradients: the differentiable transition
    for dd in (d, dg):
        for k in ("sample", "mean", "control"):
            assert rel_l2(dd[k].detach().cpu(), g[k]) < 1e-2, k
        for k in ("sigma", "entropy"):
            assert dd[k].shape == g[k].shape and rel_l2(dd[k].detach().cpu(), g[k]) < 1e-6, k
        assert rel_l2(dd["logp"].detach().cpu(), g["logp"]) < 1e-4
    dg["entropy"].sum().backward()
    want = torch.bincount(t.cpu(), minlength=10).float()          # d sum(log sigma_t) / d log_betas = visits per step, last one too
    assert torch.allclose(s.net.log_betas.grad.cpu(), want)


def test_host_schedule_tables_bit_exact(golden_dir, sampler10):
    s, _ = sampler10
    g = load(golden_dir, "schedule")
    for k in ("continuous_steps", "Gamma_bar", "x_prev_multiplier", "theta_multiplier", "std"):
        np.testing.assert_array_equal(getattr(s, k).cpu().numpy(), g[f"T10_{k}"], err_msg=k)


def test_value_forward_vs_reference_and_oracle(golden_dir):
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    from oracle import Precision
    from oracle import value as ovalue
    from oracle.weights import formula_tensor
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    sd = {k: formula_tensor(k, t.shape) for k, t in v.state_dict().items()}
    v.load_state_dict(sd)
    v = v.to(DEV).eval()
    for name in ("value_forward", "value_forward_64"):
        g = load(golden_dir, name)
        x = torch.from_numpy(g["x"])
        with torch.no_grad():
            out = v(x.to(DEV), torch.zeros(len(x), dtype=torch.long, device=DEV)).cpu()
            oo = ovalue.forward(sd, x, Precision("bf16"))
        assert out.shape == g["out"].shape == (len(x), 1)
        r_ref, r_orc = rel_l2(out, g["out"]), rel_l2(out, oo)
        print(f"{name}: rel-L2 vs reference {r_ref:.3e}, vs oracle bf16-model {r_orc:.3e}")
        assert r_ref < 1e-2 and r_orc < 5e-3


def test_full_batch_properties(sampler10):
    """BASELINE configs[1] size (B=256): per-image results do not depend on the batch they ride in
    (bit-exact: every output pixel has a fixed accumulation order), runs are reproducible bit for
    bit, and the per-sample-t path equals the constant-t path."""
    s, _ = sampler10
    g = torch.Generator().manual_seed(5)
    B = 256
    x = torch.randn(B, 3, 32, 32, generator=g).to(DEV)
    t = torch.full((B,), 170.300507, device=DEV)
    with torch.no_grad():
        y = s.net(x, t)
        y2 = s.net(x, t)
        ysub = s.net(x[40:44].contiguous(), t[:4])
    assert torch.equal(y, y2)
    assert torch.equal(ysub, y[40:44])
    assert torch.isfinite(y).all()
    z = torch.randn(B, 3, 32, 32, generator=g).to(DEV)
    ti = torch.full((B,), 5, dtype=torch.long, device=DEV)
    with torch.no_grad():
        d = s.sample_step(x, ti, noise=z)
        d_sub = s.sample_step(x[:8].contiguous(), ti[:8], noise=z[:8].contiguous())
    assert torch.equal(d["sample"][:8], d_sub["sample"]) and torch.equal(d["logp"][:8], d_sub["logp"])


def test_graft_smoke():
    import __graft_entry__ as ge
    ge.smoke()


def test_cli_train_then_generate(tmp_path):
    """Drop-in CLI surface end to end on one GPU: train_cifar10.py (3 synthetic iterations, writes
    config.yaml + sampler_last.pth with reference key names) then generate_cifar10.py from that log dir."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-by-maxentirl_amd")
    env = dict(os.environ, LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, "train_cifar10.py", "--config", "builtin:cifar10_T10", "--dataset", "builtin",
                        "--run", "t", "--synthetic_data", "--max_iters", "3", "--training.batchsize", "8",
                        "--training.n_epochs", "1", "--training.log_every", "1"], cwd=pkg, env=dict(env, PWD=pkg),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    logdir = os.path.join(pkg, "results", "cifar10", "cifar10_T10", "t")
    ck = torch.load(os.path.join(logdir, "sampler_last.pth"), map_location="cpu")
    assert len(ck["state_dict"]) == 330 and "log_betas" in ck["state_dict"] and ck["iter"] == 3
    r2 = subprocess.run([sys.executable, "generate_cifar10.py", "--log_dir", logdir, "-n", "8", "--batchsize", "8",
                         "--epoch", "last", "--skip_fid"], cwd=pkg, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    pngs = [f for f in os.listdir(os.path.join(logdir, "generated")) if f.endswith(".png")]
    assert len(pngs) == 8
    import shutil
    shutil.rmtree(os.path.join(pkg, "results"), ignore_errors=True)
