"""Train-step tail kernels on the GPU (csrc/optim.hip) through the C-ABI: multi-tensor Adam / RAdam against
torch.optim on the same device (bit-compare), device-side gradient-norm clip against torch.nn.utils.clip_grad_norm_,
replay-buffer row gather (INT path, bit-exact), and the ring replay buffer against the reference's expressions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# odd sizes on purpose: scalar tail, unaligned views, > one chunk, > DXMI_MT_MAX tensors
SHAPES = [(3,), (10,), (128, 3, 3, 3), (256, 128, 3, 3), (4097,), (1, 1), (513, 7)] + [(17 + i,) for i in range(70)]


def _params(seed, shapes=SHAPES):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(DEV) * (0.1 + 0.01 * i) for i, s in enumerate(shapes)]


def _clone_opt(cls_ref, cls_new, kw_ref, kw_new, seed=0, groups=False):
    a = [p.clone().requires_grad_(True) for p in _params(seed)]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    if groups:
        mk = lambda ps: [{"params": ps[:1], "lr": 1e-3}, {"params": ps[1:], "lr": 1e-4}]
        return a, b, cls_ref(mk(a), **kw_ref), cls_new(mk(b), **kw_new)
    return a, b, cls_ref(a, lr=1e-3, **kw_ref), cls_new(b, lr=1e-3, **kw_new)


def _set_grads(ps, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        p.grad = (torch.randn(p.shape, generator=g) * scale).to(DEV)


def _max_ulp(a, b):
    ia, ib = a.contiguous().view(torch.int32).long(), b.contiguous().view(torch.int32).long()
    return int((ia - ib).abs().max())


@pytest.mark.parametrize("groups", [False, True])
def test_fused_adam_bit_exact_vs_torch(groups):
    """dxmi_adam_step vs torch.optim.Adam (its default foreach implementation on the device): parameters and both
    moments BIT-identical over 5 steps, including the lr-per-group split of train_cifar10.py:283-296."""
    from dxmi_hip.optim import Adam
    a, b, ref, new = _clone_opt(torch.optim.Adam, Adam, {}, {}, groups=groups)
    for step in range(5):
        _set_grads(a, 100 + step, 10.0 ** (-step))
        _set_grads(b, 100 + step, 10.0 ** (-step))
        ref.step()
        new.step()
        worst = max(_max_ulp(x, y) for x, y in zip(a, b))
        assert worst == 0, (step, worst)
        for x, y in zip(a, b):
            assert torch.equal(ref.state[x]["exp_avg"], new.state[y]["exp_avg"])
            assert torch.equal(ref.state[x]["exp_avg_sq"], new.state[y]["exp_avg_sq"])
    # interchangeable state dicts
    sd = new.state_dict()
    ref.load_state_dict(sd)
    assert float(ref.state[a[0]]["step"]) == 5.0


def test_fused_radam_vs_torch():
    """dxmi_radam_step vs torch.optim.RAdam: bit-identical to the single-tensor implementation (foreach=False) whose
    arithmetic it follows, through the un-rectified phase (steps 1-5) and the rectified one; within 2 ulp of the foreach
    implementation (different but equivalent operation order)."""
    from dxmi_hip.optim import RAdam
    a, b, ref, new = _clone_opt(torch.optim.RAdam, RAdam, {"foreach": False}, {})
    c = [p.detach().clone().requires_grad_(True) for p in a]
    ref_fe = torch.optim.RAdam(c, lr=1e-3, foreach=True)
    for step in range(8):
        for ps in (a, b, c):
            _set_grads(ps, 200 + step)
        ref.step()
        new.step()
        ref_fe.step()
        assert max(_max_ulp(x, y) for x, y in zip(a, b)) == 0, step
        for x, y in zip(b, c):
            assert torch.allclose(x, y, rtol=1e-6, atol=1e-9)


def test_fused_radam_found_inf_skips_on_device():
    from dxmi_hip.optim import RAdam
    ps = [p.clone().requires_grad_(True) for p in _params(3)[:5]]
    before = [p.detach().clone() for p in ps]
    opt = RAdam(ps, lr=1e-3)
    _set_grads(ps, 9)
    flag = torch.ones(1, device=DEV)
    opt.step(found_inf=flag)
    assert all(torch.equal(p.detach(), q) for p, q in zip(ps, before))
    opt.rollback_step()
    assert float(opt.state[ps[0]]["step"]) == 0.0
    opt.step(found_inf=torch.zeros(1, device=DEV), grad_scale=torch.full((1,), 0.5, device=DEV))
    assert not torch.equal(ps[0].detach(), before[0])


def test_mixed_precision_trainer_sliced_step_is_bitwise_the_flat_path():
    """MixedPrecisionTrainer on the HIP path (round 4): the model parameters are the slices of their flat masters and
    `optimize` steps the masters tensor by tensor on the model gradients (no flatten / mul_ / copy-back passes).  Parameters,
    both moment tensors, lg_loss_scale and the logged norms must equal the reference-shaped flat path (fp16_util.py:204-223:
    flatten, norms, grad.mul_(1 / scale), opt.step on the flat masters) bit for bit over rectified and un-rectified steps, with
    an overflow step in the middle (skipped on the device, step counters rolled back, loss scale lowered) and a parameter
    that never receives a gradient."""
    import copy
    from dxmi_hip.optim import RAdam
    from models.cm.fp16_util import MixedPrecisionTrainer
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.GroupNorm(4, 16), torch.nn.Conv2d(16, 8, 1), torch.nn.Linear(5, 7)).to(DEV)
    net.register_parameter("log_betas", torch.nn.Parameter(torch.linspace(-2, 0, 4, device=DEV)))
    nets = [net, copy.deepcopy(net)]
    mps = [MixedPrecisionTrainer(model=m, use_fp16=True, initial_lg_loss_scale=12, special_key="log_betas") for m in nets]
    assert all(mp._aliased for mp in mps)
    for mp, m in zip(mps, nets):          # every model parameter IS its master's slice
        for master, (group, _) in zip(mp.master_params, mp.param_groups_and_shapes):
            off = 0
            for _, q in group:
                assert q.data_ptr() == master.data_ptr() + 4 * off
                off += q.numel()
    opts = [RAdam([{"params": mp.master_params[1:], "lr": 1e-3}, {"params": mp.master_params[0:1], "lr": 1e-2}]) for mp in mps]
    mps[1]._aliased = False               # second trainer: the flat path (its copy-back finds nothing to copy, only bumps versions)
    g = torch.Generator(device="cpu").manual_seed(11)
    for step in range(9):
        grads = [torch.randn(q.shape, generator=g).to(DEV) * 4096.0 for q in nets[0].parameters()]
        if step == 4:
            grads[2].view(-1)[3] = float("inf")
        for m in nets:
            for i, (q, gr) in enumerate(zip(m.parameters(), grads)):
                q.grad = None if i == 5 else gr.clone()         # parameter 5 never gets a gradient (zeros, as the reference)
        v0 = [q._version for q in nets[0].parameters()]
        ok = [mp.optimize(opt) for mp, opt in zip(mps, opts)]
        assert ok[0] == ok[1] == (step != 4)
        assert mps[0].lg_loss_scale == mps[1].lg_loss_scale
        for a, b in zip(nets[0].parameters(), nets[1].parameters()):
            assert torch.equal(a.detach(), b.detach()), step
        for Ma, Mb in zip(mps[0].master_params, mps[1].master_params):
            sa, sb = opts[0].state[Ma], opts[1].state[Mb]
            assert float(sa["step"]) == float(sb["step"]) == (step + 1 if step < 4 else step)
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        if ok[0]:
            assert mps[0].log["grad_norm"] == pytest.approx(mps[1].log["grad_norm"], rel=2e-6)
            assert mps[0].log["param_norm"] == pytest.approx(mps[1].log["param_norm"], rel=2e-6)
            assert all(q._version > v for q, v in zip(nets[0].parameters(), v0))       # packed-weight caches see the update
        assert all(q.grad is None for q in nets[0].parameters())
    assert abs(mps[0].lg_loss_scale - (12 - 1 + 8 * 1e-3)) < 1e-9


@pytest.mark.parametrize("max_norm", [0.1, 1e6])
def test_gradnorm_clip_vs_torch(max_norm):
    from dxmi_hip import ops
    a = [p.clone().requires_grad_(True) for p in _params(1)]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    _set_grads(a, 5)
    _set_grads(b, 5)
    ref_norm = torch.nn.utils.clip_grad_norm_(a, max_norm)
    out = ops.gradnorm_clip([p.grad for p in b], max_norm)
    torch.cuda.synchronize()
    assert abs(out[0].item() - ref_norm.item()) <= 2e-6 * ref_norm.item()
    assert out[2].item() == 0.0
    for x, y in zip(a, b):
        assert torch.allclose(x.grad, y.grad, rtol=3e-6, atol=0)
    # reproducible bit for bit, and inf/nan is flagged
    out2 = ops.gradnorm_clip([p.grad.clone() for p in b], 0.0)
    out3 = ops.gradnorm_clip([p.grad.clone() for p in b], 0.0)
    assert torch.equal(out2, out3)
    b[3].grad[0, 0, 0, 0] = float("inf")
    assert ops.gradnorm_clip([p.grad for p in b], 0.0)[2].item() == 1.0


def test_gather_rows_bit_exact():
    """INT path: every row width class (16-byte vectors, 8-byte and 4-byte elements), negative indices, repeated rows."""
    from dxmi_hip import ops
    g = torch.Generator().manual_seed(8)
    idx = torch.tensor([5, 0, 39, 39, -1, 17, 2], dtype=torch.int64)
    for src in (torch.randn(40, 3, 32, 32, generator=g), torch.randn(40, 1, 1, 1, generator=g), torch.randn(40, generator=g),
                torch.arange(40, dtype=torch.int64) * 7, torch.randn(40, 6, generator=g), torch.randn(40, 3, 64, 64, generator=g)):
        got = ops.gather_rows(src.to(DEV), idx.to(DEV))
        assert got.dtype == src.dtype and torch.equal(got.cpu(), src[idx]), src.shape
    bad = ops.gather_rows(torch.zeros(4, 8, device=DEV), torch.tensor([1, 9], device=DEV))
    assert torch.equal(bad[0].cpu(), torch.zeros(8)) and torch.isnan(bad[1]).all()     # out of range: poisoned, not OOB


def _cifar_models(T):
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    from oracle.weights import formula_tensor
    net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.0, in_channels=3, resolution=32)
    sampler = VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last")
    net.load_state_dict({k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()})
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    v.load_state_dict({k: formula_tensor(k, t.shape) for k, t in v.state_dict().items()})
    return net, sampler.to(DEV), v.to(DEV)


def test_ring_buffer_gathers_match_reference_double_index():
    """f1: trajectories generated IN PLACE in the ring; every field gathered by dxmi_gather_rows equals the reference's
    `state_dict[key][indices][train_indices]` double fancy index on the torch.cat buffer, bit for bit (trainer.py:278-289)."""
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import append_buffer, reset_buffer
    B, T = 6, 4
    net, sampler, _ = _cifar_models(T)
    sampler.eval()
    ring = TransitionRing(2, T, B, (3, 32, 32), DEV)
    ref = reset_buffer(DEV)
    for rep in range(2):
        g = torch.Generator().manual_seed(40 + rep)
        noise = [torch.randn(B, 3, 32, 32, generator=g) for _ in range(T + 1)]
        d_ring = sampler.sample(B, device=DEV, noise=noise, out=ring.next_slot())
        assert d_ring["l_sample"][1].data_ptr() == ring.traj[rep, 1].data_ptr()          # written in place: no copy on append
        append_buffer(ring, d_ring)
        ref = append_buffer(ref, sampler.sample(B, device=DEV, noise=noise))              # reference-style dict (torch.cat)
    sd = ring.as_state_dict()
    for k in ("state", "next_state", "timestep", "logp", "control", "mean", "sigma"):
        assert torch.equal(sd[k], ref[k]), k
    torch.manual_seed(11)
    indices = (torch.randperm(B * T) + (ring.n_rows - B * T)).to(DEV)
    for update_t in range(T):
        train_indices = torch.nonzero(ref["timestep"][indices] == update_t).flatten()
        rows = indices[train_indices]
        for k in ("state", "next_state", "mean", "control", "sigma", "logp", "timestep", "final"):
            assert torch.equal(ring.gather(k, rows), ref[k][indices][train_indices]), (k, update_t)


def test_trainer_step_ring_equals_dict_buffer():
    """The whole HIP train step is bit-identical whether transitions live in the ring (in-place sampling, gather
    kernel, one stable sort) or in the reference-style dict (torch.cat, fancy index, per-step nonzero)."""
    from dxmi_hip.optim import Adam
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import DxMI_Trainer, append_buffer, reset_buffer
    B, T = 4, 4
    logs, finals = [], []
    for use_ring in (False, True):
        net, sampler, v = _cifar_models(T)
        not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
        opt = Adam([{"params": net.log_betas, "lr": 1e-5}, {"params": not_beta, "lr": 1e-7}])
        opt_v = Adam(v.parameters(), lr=1e-5)
        tr = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99,
                          time_cost_sig=1, n_timesteps=T, entropy_in_value=0)
        tr.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
        g = torch.Generator().manual_seed(3)
        img = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(DEV)
        noise = [torch.randn(B, 3, 32, 32, generator=g) for _ in range(T + 1)]
        zs = torch.randn(B, 3, 32, 32, generator=g).to(DEV)
        sampler.eval()
        if use_ring:
            buf = TransitionRing(1, T, B, (3, 32, 32), DEV)
            d = sampler.sample(B, device=DEV, noise=noise, out=buf.next_slot())
        else:
            buf = reset_buffer(DEV)
            d = sampler.sample(B, device=DEV, noise=noise)
        buf = append_buffer(buf, d)
        torch.manual_seed(77)
        le = tr.update_f_v(img, d, buf)
        orig = sampler.sample_step
        sampler.sample_step = lambda x, t, y=None: orig(x, t, noise=zs)
        ls = tr.update_sampler(buf, 1)
        logs.append((le, ls))
        finals.append([p.detach().clone() for p in list(v.parameters()) + list(net.parameters())])
    assert logs[0] == logs[1]
    assert all(torch.equal(a, b) for a, b in zip(*finals))


# ------------------------------------------------------------------------------------------------ round-3 regressions
def _small_unet():
    from models.DxMI.unet_small import Model
    from oracle.weights import formula_tensor
    net = Model(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=[8], dropout=0.0, in_channels=3, resolution=16)
    net.load_state_dict({k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()},
                        strict=False)
    return net.to(DEV)


def test_fused_adam_step_invalidates_packed_weights():
    """The fused optimisers write parameters through raw pointers; every packed bf16 weight cache is keyed on
    (data_ptr, _version).  Two train steps with dxmi_hip.optim.Adam and with torch.optim.Adam (lr large enough that a
    stale pack would be obvious) must leave the two nets computing the same function, and the pack must be rebuilt."""
    from dxmi_hip.optim import Adam
    torch.manual_seed(0)
    x = torch.randn(4, 3, 16, 16, device=DEV)
    t = torch.tensor([3.0, 50.0, 400.0, 900.0], device=DEV)
    outs = []
    for cls in (torch.optim.Adam, Adam):
        net = _small_unet()
        opt = cls([p for n, p in net.named_parameters() if n != "log_betas"], lr=1e-3)
        with torch.no_grad():
            y0 = net(x, t).clone()
        key0 = net._packed_key
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            net(x, t).square().mean().backward()
            opt.step()
        assert net._param_key() != key0, "optimizer.step() must move the parameters' version counters"
        with torch.no_grad():
            y2 = net(x, t).clone()
        assert net._packed_key != key0                                   # the fragments were re-packed
        assert (y2 - y0).norm() > 1e-2 * y0.norm(), "two lr=1e-3 Adam steps must change the network's output"
        outs.append(y2)
    rel = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
    assert rel < 2e-2, rel                                                # same function up to the bf16 noise of the backward


def test_fused_adam_lagging_step_counters_and_reloaded_state():
    """A parameter that receives its first gradient later has a lagging step counter (torch.optim handles it per tensor):
    the fused step forms one launch series per step count instead of raising.  load_state_dict() replaces the moment
    tensors: the cached pointer arrays must follow."""
    from dxmi_hip.optim import Adam
    a = [p.clone().requires_grad_(True) for p in _params(7)[:6]]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    ref, new = torch.optim.Adam(a, lr=1e-3), Adam(b, lr=1e-3)
    for step in range(4):
        _set_grads(a, 300 + step)
        _set_grads(b, 300 + step)
        if step < 2:                       # the last two tensors join at step 2
            for ps in (a, b):
                ps[4].grad = ps[5].grad = None
        ref.step()
        new.step()
        assert max(_max_ulp(x, y) for x, y in zip(a, b)) == 0, step
    assert float(new.state[b[5]]["step"]) == 2.0 and float(new.state[b[0]]["step"]) == 4.0
    # reload: new moment tensors at new addresses
    import copy
    new.load_state_dict(copy.deepcopy(ref.state_dict()))      # (load_state_dict aliases same-device tensors: copy first)
    _set_grads(a, 400)
    _set_grads(b, 400)
    ref.step()
    new.step()
    assert max(_max_ulp(x, y) for x, y in zip(a, b)) == 0
    for x, y in zip(a, b):
        assert torch.equal(ref.state[x]["exp_avg_sq"], new.state[y]["exp_avg_sq"])
    # p.data replaced: same python object, new storage
    with torch.no_grad():
        for ps in (a, b):
            ps[0].data = ps[0].data.clone()
    _set_grads(a, 401)
    _set_grads(b, 401)
    ref.step()
    new.step()
    assert max(_max_ulp(x, y) for x, y in zip(a, b)) == 0


# (Cin, Cout, H, k, batch, out_nchw_f32): one shape per forward conv kernel family
NAN_CASES = {"ws<32>": (128, 128, 32, 3, 8, False), "ws<16>": (256, 256, 16, 3, 16, False), "ws8": (256, 256, 8, 3, 8, False),
             "pipe": (256, 256, 4, 3, 32, False), "1x1_rw": (256, 768, 16, 1, 128, False), "1x1_stream": (256, 256, 4, 1, 8, False),
             "head": (128, 3, 32, 3, 4, True)}


@pytest.mark.parametrize("case", list(NAN_CASES))
def test_conv_epilogue_keeps_nan(case):
    """A NaN activation must come out of every conv kernel family as NaN (act none / leaky / relu), so that the non-finite
    checks downstream (gradient-norm flag, MixedPrecisionTrainer overflow test, NaN poisoning of bad timesteps) can fire."""
    from dxmi_hip import ops
    cin, cout, hw, k, n, nchw = NAN_CASES[case]
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, hw, hw, cin, generator=g).to(DEV).to(torch.bfloat16)
    x[0, hw // 2, hw // 2, 5] = float("nan")
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(DEV)
    pw = ops.pack_conv_weight(w)
    bias = torch.zeros(cout, device=DEV)
    for act in (ops.ACT_NONE, ops.ACT_LEAKY02, ops.ACT_RELU):
        y = ops.conv2d(x, pw, bias=bias, act=act, out_nchw_f32=nchw)
        y = y.permute(0, 2, 3, 1) if nchw else y
        assert torch.isnan(y[0, hw // 2, hw // 2].float()).all(), (case, act)
        assert torch.isfinite(y[1:].float()).all(), (case, act)
