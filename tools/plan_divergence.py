"""GPU box: does re-packing by plan replay change a train step?  Two identically seeded EDM train steps, ops.PACK_PLAN_REPLAY on / off;
prints a checksum of the master parameters after every optimiser step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
from dxmi_hip.optim import Adam, RAdam
import configs_builtin, dxmi_config
from models.cm.fp16_util import MixedPrecisionTrainer
from models.cm.script_util import create_model_and_diffusion
from models.DxMI.openai_diffusion import OpenAIDiffusion
from models.DxMI.trainer import append_buffer, reset_buffer

dev, B = "cuda:0", 4
cfg = configs_builtin.get("imagenet64_T10")


TRACE = []
_WRAPPED = {}


def _cs(o):
    if torch.is_tensor(o):
        return float(o.detach().double().sum()) if o.numel() else 0.0
    if isinstance(o, (tuple, list)):
        return tuple(_cs(x) for x in o if torch.is_tensor(x) or isinstance(x, (tuple, list)) or hasattr(x, "buf"))
    if hasattr(o, "buf") and torch.is_tensor(getattr(o, "buf")):
        return float(o.buf.double().sum()) if o.buf.dtype != torch.uint8 else float(o.buf.long().sum())
    return None


def trace_ops(on):
    names = ["conv2d", "groupnorm_silu", "groupnorm_generic", "groupnorm_apply", "groupnorm_generic_bwd", "attention", "attention_bwd", "conv2d_wgrad",
             "linear", "linear_bwd", "upsample2x", "pool_act", "block_stats", "colsum_per_image", "colsum", "gradnorm_clip", "stem_conv_wgrad"]
    for n in names:
        if not hasattr(ops, n):
            continue
        if on and n not in _WRAPPED:
            f = getattr(ops, n)
            _WRAPPED[n] = f

            def w(*a, _f=f, _n=n, **k):
                r = _f(*a, **k)
                shp = tuple(a[0].shape) if a and torch.is_tensor(a[0]) else None
                TRACE.append((_n, shp, _cs(r)))
                return r
            setattr(ops, n, w)
        elif not on and n in _WRAPPED:
            setattr(ops, n, _WRAPPED.pop(n))


def run(replay):
    ops.PACK_PLAN_REPLAY = replay
    torch.manual_seed(0)
    unet, diffusion = create_model_and_diffusion(**cfg.diffusion)
    for p in unet.parameters():
        if p.abs().max() == 0:
            torch.nn.init.normal_(p, std=0.02)
    sampler = OpenAIDiffusion(unet, diffusion, **cfg.sampler)
    unet.to(dev)
    v = dxmi_config.instantiate(cfg.value).to(dev)
    mp = MixedPrecisionTrainer(model=unet, use_fp16=True, initial_lg_loss_scale=20, special_key="log_betas")
    opt = RAdam([{"params": mp.master_params[1:], "lr": 1e-8}, {"params": mp.master_params[0:1], "lr": 1e-6}])
    opt_v = Adam(v.parameters(), lr=1e-5)
    trainer = dxmi_config.instantiate(cfg.trainer, batchsize=B)
    trainer.set_models(v=v, sampler=sampler, optimizer=opt, optimizer_v=opt_v)
    sums = []
    orig = mp.optimize

    state = {"n": 0}

    def spy(o):
        r = orig(o)
        sums.append(float(torch.stack([m.detach().double().sum() for m in mp.master_params]).sum()))
        state["n"] += 1
        if state["n"] == 1:
            TRACE.clear()
            trace_ops(True)          # trace the SECOND sampler iteration
        elif state["n"] == 2:
            trace_ops(False)
        return r
    mp.optimize = spy
    g = torch.Generator(device=dev).manual_seed(1)
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    for _ in range(2):
        data = torch.rand(B, 3, 64, 64, device=dev, generator=g) * 2 - 1
        y = torch.randint(0, 1000, (B,), device=dev, generator=g)
        sampler.eval()
        d = sampler.sample(B, device=dev, i_class=y)
        sums.append(float(d["sample"].double().sum()))
        buf = append_buffer(reset_buffer(dev), d)
        le = trainer.update_f_v(data, d, buf, y=y)
        sums.append(le["ebm/v_loss_"])
        trainer.update_sampler_mixed_precision(buf, mp_trainer=mp)
    return sums


a = run(True)
ta = list(TRACE)
b = run(False)
tb = list(TRACE)
for i, (x, z) in enumerate(zip(a[:6], b[:6])):
    print(i, repr(x), repr(z), "" if x == z else "   <-- differs")
print("traced ops", len(ta), len(tb))
shown = 0
for i, (x, z) in enumerate(zip(ta, tb)):
    if x != z:
        print("first differing op", i, x, z)
        for j in range(max(0, i - 3), i):
            print("   before:", j, ta[j])
        shown += 1
        if shown >= 3:
            break


def compare_packs():
    """replayed vs fresh packs of the full net under MixedPrecisionTrainer aliasing, after one in-place update"""
    from models.cm.unet_train import _pack_t
    ops.PACK_PLAN_REPLAY = True
    torch.manual_seed(0)
    unet, diffusion = create_model_and_diffusion(**cfg.diffusion)
    for p in unet.parameters():
        if p.abs().max() == 0:
            torch.nn.init.normal_(p, std=0.02)
    OpenAIDiffusion(unet, diffusion, **cfg.sampler)          # adds log_betas to the net
    unet.to(dev)
    mp = MixedPrecisionTrainer(model=unet, use_fp16=True, initial_lg_loss_scale=20, special_key="log_betas")
    names = {id(m): n for n, m in unet.named_modules()}

    def snap():
        out = {}
        for tag, d in (("f", unet.packed()), ("t", _pack_t(unet))):
            for k, v in d.items():
                key = (tag, k if isinstance(k, str) else (names[k[0]], k[1]))
                if isinstance(v, ops.PackedConvWeight):
                    out[key] = v.buf.clone()
                elif torch.is_tensor(v):
                    out[key] = v.clone()
        return out
    snap()
    with torch.no_grad():
        for m in mp.master_params:
            m.add_(torch.randn_like(m) * 1e-3)
        for p in unet.parameters():
            p._version if False else None
    # versions of the aliased model parameters must change for the key to change
    for p in unet.parameters():
        p.data.add_(0)
    a = snap()
    print("replayed:", unet._pack_plan is not None, "fp32_params", unet._fp32_params)
    unet._pack_plan = unet._pack_t_plan = None
    unet._packed = unet._packed_t = None
    b = snap()
    bad = [k for k in b if not torch.equal(a[k], b[k])]
    print("pack entries", len(b), "differing", len(bad), bad[:10])

