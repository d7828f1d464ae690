"""Oracle: ADM/EDM U-Net + Karras preconditioning + Euler-ancestral few-step sampler on the CPU
(TEST INFRASTRUCTURE — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline import this).

Restates, as pure functions over a reference-keyed state dict:
  models/cm/unet.py:554-790   UNetModel (constructor plan + forward)
  models/cm/unet.py:147-260   ResBlock (scale-shift norm, resblock up/down)
  models/cm/unet.py:263-333   AttentionBlock with QKVAttentionLegacy (:401-441; layout "(three h d)")
  models/cm/nn.py:119-137     timestep_embedding ([cos | sin], freq = exp(-ln(P) i / half))
  models/cm/karras_diffusion.py:64-68, :337-351   get_scalings / denoise
  models/cm/karras_diffusion.py:423-429           get_sigmas_karras
  models/DxMI/openai_diffusion.py:11-129          OpenAIDiffusion (schedule, sample_step, sample)
Precision("fp32") is the reference arithmetic with the network in fp32 (the legacy attention still
rounds q,k,v to fp16 exactly as unet.py:421 does); Precision("bf16") mirrors the storage points of the
HIP pipeline (bf16 activations between kernels, bf16 weight operands, fp32 accumulation).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from .precision import Precision


@dataclass
class EDMConfig:
    image_size: int = 64
    in_channels: int = 3
    model_channels: int = 192
    out_channels: int = 3
    num_res_blocks: int = 3
    attention_resolutions: tuple = (2, 4, 8)   # downsample rates, script_util.py:137-139
    channel_mult: tuple = (1, 2, 3, 4)
    num_classes: int = 1000                    # None: unconditional
    num_heads: int = 4
    num_head_channels: int = 64
    num_heads_upsample: int = -1
    use_scale_shift_norm: bool = True
    resblock_updown: bool = True
    conv_resample: bool = True


def plan(cfg):
    """Block list of UNetModel.__init__ (unet.py:607-743): (prefix, [layer descriptors])."""
    mc = cfg.model_channels
    nhu = cfg.num_heads if cfg.num_heads_upsample == -1 else cfg.num_heads_upsample
    heads = lambda ch, nh: nh if cfg.num_head_channels == -1 else ch // cfg.num_head_channels
    ch = int(cfg.channel_mult[0] * mc)
    inp = [("input_blocks.0", [("conv", cfg.in_channels, ch)])]
    chans, ds = [ch], 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, int(mult * mc), None)]
            ch = int(mult * mc)
            if ds in cfg.attention_resolutions:
                layers.append(("attn", ch, heads(ch, cfg.num_heads)))
            inp.append((f"input_blocks.{len(inp)}", layers))
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            layers = [("res", ch, ch, "down")] if cfg.resblock_updown else [("downsample", ch, cfg.conv_resample)]
            inp.append((f"input_blocks.{len(inp)}", layers))
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, None), ("attn", ch, heads(ch, cfg.num_heads)), ("res", ch, ch, None)]
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, int(mc * mult), None)]
            ch = int(mc * mult)
            if ds in cfg.attention_resolutions:
                layers.append(("attn", ch, heads(ch, nhu)))
            if level and i == cfg.num_res_blocks:
                layers.append(("res", ch, ch, "up") if cfg.resblock_updown else ("upsample", ch, cfg.conv_resample))
                ds //= 2
            out.append((f"output_blocks.{len(out)}", layers))
    return inp, mid, out, ch


def state_dict_shapes(cfg, log_betas_len=None):
    """name -> shape in the reference's registration order (what net.state_dict() lists)."""
    inp, mid, out, ch_out = plan(cfg)
    ted = cfg.model_channels * 4
    sh = {"time_embed.0.weight": (ted, cfg.model_channels), "time_embed.0.bias": (ted,),
          "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}
    if cfg.num_classes is not None:
        sh["label_emb.weight"] = (cfg.num_classes, ted)

    def conv(name, co, ci, k):
        sh[name + ".weight"], sh[name + ".bias"] = (co, ci, k, k), (co,)

    def norm(name, c):
        sh[name + ".weight"], sh[name + ".bias"] = (c,), (c,)

    def layers(pre, ls):
        for j, L in enumerate(ls):
            p = f"{pre}.{j}"
            if L[0] == "conv":
                conv(p, L[2], L[1], 3)
            elif L[0] == "res":
                norm(p + ".in_layers.0", L[1])
                conv(p + ".in_layers.2", L[2], L[1], 3)
                eo = 2 * L[2] if cfg.use_scale_shift_norm else L[2]
                sh[p + ".emb_layers.1.weight"], sh[p + ".emb_layers.1.bias"] = (eo, ted), (eo,)
                norm(p + ".out_layers.0", L[2])
                conv(p + ".out_layers.3", L[2], L[2], 3)
                if L[1] != L[2]:
                    conv(p + ".skip_connection", L[2], L[1], 1)
            elif L[0] == "attn":
                norm(p + ".norm", L[1])
                conv(p + ".qkv", 3 * L[1], L[1], 1)
                conv(p + ".proj_out", L[1], L[1], 1)
            elif L[0] == "downsample" and L[2]:
                conv(p + ".op", L[1], L[1], 3)
            elif L[0] == "upsample" and L[2]:
                conv(p + ".conv", L[1], L[1], 3)

    for pre, ls in inp:
        layers(pre, ls)
    layers("middle_block", mid)
    for pre, ls in out:
        layers(pre, ls)
    norm("out.0", ch_out)
    conv("out.2", cfg.out_channels, int(cfg.channel_mult[0] * cfg.model_channels), 3)
    if log_betas_len is not None:
        sh["log_betas"] = (log_betas_len,)
    return sh


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _conv(sd, name, x, prec, stride=1, padding=0):
    w = sd[name + ".weight"]
    if w.dim() == 3:
        w = w[..., None]
    return F.conv2d(x, prec.w(w), sd[name + ".bias"], stride=stride, padding=padding)


def _gn(sd, name, x):
    return F.group_norm(x.float(), 32, sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def res_block(sd, pre, x, emb, cin, cout, updown, cfg, prec):
    """ResBlock._forward, unet.py:240-260."""
    h = prec.act(F.silu(_gn(sd, pre + ".in_layers.0", x)))
    if updown == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif updown == "down":
        h = prec.act(F.avg_pool2d(h, 2, 2))
        x = prec.act(F.avg_pool2d(x, 2, 2))
    h = _conv(sd, pre + ".in_layers.2", h, prec, padding=1)
    emb_out = F.linear(prec.act(F.silu(emb)), prec.w(sd[pre + ".emb_layers.1.weight"]), sd[pre + ".emb_layers.1.bias"])
    emb_out = emb_out[:, :, None, None]
    if cfg.use_scale_shift_norm:
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _gn(sd, pre + ".out_layers.0", prec.act(h)) * (1 + scale) + shift
        h = prec.act(F.silu(h))
    else:
        h = prec.act(h + emb_out)
        h = prec.act(F.silu(_gn(sd, pre + ".out_layers.0", h)))
    h = _conv(sd, pre + ".out_layers.3", h, prec, padding=1)
    if cin != cout:
        x = prec.act(_conv(sd, pre + ".skip_connection", x, prec))
    return prec.act(x + h)


def attention_block(sd, pre, x, heads, prec):
    """AttentionBlock._forward (unet.py:320-333) with QKVAttentionLegacy.forward (:413-441)."""
    b, c, hh, ww = x.shape
    length = hh * ww
    qkv = prec.act(_conv(sd, pre + ".qkv", prec.act(_gn(sd, pre + ".norm", x)), prec)).reshape(b, 3 * c, length)
    ch = c // heads
    scale = 1 / math.sqrt(math.sqrt(ch))
    if prec.mode == "fp32":
        qkv = qkv.half()   # unet.py:421 — the legacy attention always runs its einsums in fp16
    q, k, v = [z.reshape(b * heads, ch, length) for z in qkv.reshape(b, 3, heads, ch, length).unbind(1)]
    if prec.mode == "fp32":
        weight = torch.einsum("bct,bcs->bts", q * scale, k * scale)
        weight = torch.softmax(weight, dim=-1).type(weight.dtype)
        a = torch.einsum("bts,bcs->bct", weight, v).float()
    else:
        # HIP kernel: S = (q.k) * ch^-0.5 in fp32, un-normalised exp() rounded to bf16 for the PV MFMA
        s = torch.einsum("bct,bcs->bts", q, k) * (scale * scale)
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        a = torch.einsum("bts,bcs->bct", prec.p(e), v) / e.sum(dim=-1)[:, None, :]
    a = prec.act(a.reshape(b, c, hh, ww))
    return prec.act(x + _conv(sd, pre + ".proj_out", a, prec))


def _layers(sd, pre, layers, h, emb, cfg, prec):
    for j, L in enumerate(layers):
        p = f"{pre}.{j}"
        if L[0] == "conv":
            h = prec.act(_conv(sd, p, prec.act(h), prec, padding=1))
        elif L[0] == "res":
            h = res_block(sd, p, h, emb, L[1], L[2], L[3], cfg, prec)
        elif L[0] == "attn":
            h = attention_block(sd, p, h, L[2], prec)
        elif L[0] == "downsample":
            h = prec.act(_conv(sd, p + ".op", h, prec, stride=2, padding=1)) if L[2] else prec.act(F.avg_pool2d(h, 2, 2))
        elif L[0] == "upsample":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if L[2]:
                h = prec.act(_conv(sd, p + ".conv", h, prec, padding=1))
    return h


def unet_forward(sd, cfg, x, timesteps, y=None, prec=None, trace=None):
    """UNetModel.forward, unet.py:761-790."""
    prec = prec or Precision("fp32")
    tr = (lambda n, v: trace.append((n, v.clone()))) if trace is not None else (lambda n, v: None)
    assert (y is not None) == (cfg.num_classes is not None)
    inp, mid, out, _ = plan(cfg)
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(prec.act(emb), prec.w(sd["time_embed.0.weight"]), sd["time_embed.0.bias"])
    emb = F.linear(prec.act(F.silu(emb)), prec.w(sd["time_embed.2.weight"]), sd["time_embed.2.bias"])
    if cfg.num_classes is not None:
        emb = emb + sd["label_emb.weight"][y]
    tr("emb", emb)
    hs, h = [], x
    for pre, layers in inp:
        h = _layers(sd, pre, layers, h, emb, cfg, prec)
        tr(pre, h)
        hs.append(h)
    h = _layers(sd, "middle_block", mid, h, emb, cfg, prec)
    tr("middle_block", h)
    for pre, layers in out:
        h = _layers(sd, pre, layers, torch.cat([h, hs.pop()], dim=1), emb, cfg, prec)
        tr(pre, h)
    h = prec.act(F.silu(_gn(sd, "out.0", h)))
    return _conv(sd, "out.2", h, prec, padding=1)


# --------------------------------------------------------------------------- Karras / EDM
def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat([sigmas, sigmas.new_zeros([1])])


def get_scalings(sigma, sigma_data=0.5):
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2) ** 0.5
    c_in = 1 / (sigma ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out, c_in


def denoise(model, x_t, sigmas, **kw):
    c_skip, c_out, c_in = [s[:, None, None, None] for s in get_scalings(sigmas)]
    rescaled_t = 1000 * 0.25 * torch.log(sigmas + 1e-44)
    model_output = model(c_in * x_t, rescaled_t, **kw)
    return model_output, c_out * model_output + c_skip * x_t


class EDMSchedule:
    """OpenAIDiffusion.__init__ (openai_diffusion.py:11-47) — host tables."""

    def __init__(self, n_timesteps, sigma_min=0.002, sigma_max=80.0, stochastic_last=False, rho=7.0, trainable_beta=False):
        self.n_timesteps, self.sigma_max, self.trainable_beta = n_timesteps, sigma_max, trainable_beta
        if stochastic_last:
            self.sigmas = get_sigmas_karras(n_timesteps + 1, sigma_min, sigma_max, rho=rho)[:-1]
        else:
            self.sigmas = get_sigmas_karras(n_timesteps, sigma_min, sigma_max, rho=rho)
        sf, st = self.sigmas[:-1], self.sigmas[1:]
        self.sigma_up = (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** 0.5
        self.sigma_down = (st ** 2 - self.sigma_up ** 2) ** 0.5
        self.log_betas = torch.log(self.sigma_up.clamp(1e-3)) if trainable_beta else torch.log(self.sigma_up)


def sample_step(model, sch, x, indices, z, log_betas=None, **kw):
    """OpenAIDiffusion.sample_step (openai_diffusion.py:65-100); z replaces torch.randn_like."""
    sigma = sch.sigmas[indices]
    _, denoised = denoise(model, x, sigma, **kw)
    sigma_down, sigma_up = sch.sigma_down[indices], sch.sigma_up[indices]
    d = (x - denoised) / sigma[:, None, None, None]
    dt = (sigma_down - sigma)[:, None, None, None]
    mu = x + d * dt
    if sch.trainable_beta:
        lb = sch.log_betas if log_betas is None else log_betas
        s = torch.exp(lb[indices])
        if sch.trainable_beta == "fix_last":
            terminal = indices == sch.n_timesteps - 1
            s = s * ~terminal + sigma_up * terminal
        elif sch.trainable_beta == "fix_last3":
            non_terminal = indices < sch.n_timesteps - 3
            s = s * non_terminal + sigma_up * (~non_terminal)
        sigma_up = s
    return {"sample": mu + z * sigma_up[:, None, None, None], "mean": mu, "sigma": sigma_up.clamp(1e-4, None)}


def sample(model, sch, x0, noises, log_betas=None, **kw):
    """OpenAIDiffusion.sample (openai_diffusion.py:102-129): x0 = randn * sigma_max supplied by the caller."""
    x = x0
    l_x, l_mean, l_sigma = [x], [], []
    for i in range(sch.n_timesteps):
        d = sample_step(model, sch, x, i * torch.ones(len(x), dtype=torch.long), noises[i], log_betas=log_betas, **kw)
        x = d["sample"]
        l_x.append(x)
        l_mean.append(d["mean"])
        l_sigma.append(d["sigma"])
    return {"sample": x, "l_sample": l_x, "mean": l_mean, "sigma": l_sigma}
