"""Oracle: DDPM U-Net forward as a functional torch-CPU program (TEST INFRASTRUCTURE).

Restates models/DxMI/unet_small.py (reference root) over a flat state dict `sd`
(name -> tensor, reference key names), so no reference code is needed at run time:
  timestep embedding  unet_small.py:9-27     temb MLP            :296-299
  ResnetBlock         :117-136               AttnBlock           :167-191
  Downsample          :69-76                 Upsample            :50-54
  Model.forward       :292-332
`prec` (oracle.Precision) chooses reference fp32 arithmetic or the bf16 storage model of the HIP
pipeline.  `dropout_masks` is unused: parity fixtures run the net in eval mode / dropout 0.
"""
import math

import torch
import torch.nn.functional as F

from .precision import Precision


class UNetSmallConfig:
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,),
                 dropout=0.0, resamp_with_conv=True, in_channels=3, resolution=32):
        self.ch, self.out_ch, self.ch_mult = ch, out_ch, tuple(ch_mult)
        self.num_res_blocks, self.attn_resolutions = num_res_blocks, tuple(attn_resolutions)
        self.dropout, self.resamp_with_conv = dropout, resamp_with_conv
        self.in_channels, self.resolution = in_channels, resolution


def timestep_embedding_sincos(t, dim):
    """unet_small.py:9-27 — [sin | cos], frequencies exp(-i*ln(1e4)/(half-1)), fp32."""
    half = dim // 2
    step = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    arg = t.float()[:, None] * freqs[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def swish(x):
    return x * torch.sigmoid(x)


def _conv(sd, name, x, prec, stride=1, padding=0):
    return F.conv2d(x, prec.w(sd[name + ".weight"]), sd[name + ".bias"], stride=stride, padding=padding)


def _gn(sd, name, x, eps=1e-6):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps)


def resnet_block(sd, pre, x, s_temb, cin, cout, prec):
    """unet_small.py:117-136.  `s_temb` = swish(temb) (shared by all blocks).  x is a stored
    activation (already prec.act-rounded)."""
    h = prec.act(swish(_gn(sd, pre + ".norm1", x)))
    h = _conv(sd, pre + ".conv1", h, prec, padding=1)
    tproj = F.linear(prec.act(s_temb), prec.w(sd[pre + ".temb_proj.weight"]), sd[pre + ".temb_proj.bias"])
    h = prec.act(h + tproj[:, :, None, None])
    h = prec.act(swish(_gn(sd, pre + ".norm2", h)))
    h = _conv(sd, pre + ".conv2", h, prec, padding=1)
    if cin != cout:
        x = prec.act(_conv(sd, pre + ".nin_shortcut", x, prec))
    return prec.act(x + h)


def attn_block(sd, pre, x, prec):
    """unet_small.py:167-191 — single head over H*W tokens, scale C^-0.5."""
    b, c, hh, ww = x.shape
    hn = prec.act(_gn(sd, pre + ".norm", x))
    q = prec.act(_conv(sd, pre + ".q", hn, prec)).reshape(b, c, hh * ww).permute(0, 2, 1)
    k = prec.act(_conv(sd, pre + ".k", hn, prec)).reshape(b, c, hh * ww)
    v = prec.act(_conv(sd, pre + ".v", hn, prec)).reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = torch.softmax(w_, dim=2)
    if prec.mode == "bf16":
        # the HIP kernel rounds the un-normalised exp() to bf16 for the PV MFMA and divides by the
        # fp32 row sum afterwards
        m = torch.bmm(q, k) * (int(c) ** (-0.5))
        e = torch.exp(m - m.max(dim=2, keepdim=True).values)
        w_ = prec.p(e) / e.sum(dim=2, keepdim=True)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    h_ = prec.act(h_)
    h_ = _conv(sd, pre + ".proj_out", h_, prec)
    return prec.act(x + h_)


def forward(sd, cfg, x, t, prec=None, trace=None):
    """Model.forward, unet_small.py:292-332.  x [B,C,H,W] fp32, t [B] float.
    trace: optional list that receives (name, NCHW tensor) after every block (debugging aid)."""
    tr = (lambda n, v: trace.append((n, v.clone()))) if trace is not None else (lambda n, v: None)
    prec = prec or Precision("fp32")
    assert x.shape[2] == x.shape[3] == cfg.resolution
    nres = len(cfg.ch_mult)
    in_ch_mult = (1,) + cfg.ch_mult

    temb = timestep_embedding_sincos(t, cfg.ch)
    temb = F.linear(prec.act(temb), prec.w(sd["temb.dense.0.weight"]), sd["temb.dense.0.bias"])
    temb = swish(temb)
    temb = F.linear(prec.act(temb), prec.w(sd["temb.dense.1.weight"]), sd["temb.dense.1.bias"])
    s_temb = swish(temb)
    tr("s_temb", s_temb)

    hs = [prec.act(_conv(sd, "conv_in", prec.act(x), prec, padding=1))]
    tr("conv_in", hs[0])
    curr_res = cfg.resolution
    for i_level in range(nres):
        block_in = cfg.ch * in_ch_mult[i_level]
        block_out = cfg.ch * cfg.ch_mult[i_level]
        for i_block in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"down.{i_level}.block.{i_block}", hs[-1], s_temb, block_in, block_out, prec)
            tr(f"down.{i_level}.block.{i_block}", h)
            block_in = block_out
            if curr_res in cfg.attn_resolutions:
                h = attn_block(sd, f"down.{i_level}.attn.{i_block}", h, prec)
                tr(f"down.{i_level}.attn.{i_block}", h)
            hs.append(h)
        if i_level != nres - 1:
            # asymmetric zero pad (0,1,0,1) then 3x3 stride 2, unet_small.py:69-76
            hp = F.pad(hs[-1], (0, 1, 0, 1), mode="constant", value=0)
            hs.append(prec.act(_conv(sd, f"down.{i_level}.downsample.conv", hp, prec, stride=2)))
            tr(f"down.{i_level}.downsample", hs[-1])
            curr_res //= 2

    h = hs[-1]
    h = resnet_block(sd, "mid.block_1", h, s_temb, block_in, block_in, prec)
    tr("mid.block_1", h)
    h = attn_block(sd, "mid.attn_1", h, prec)
    tr("mid.attn_1", h)
    h = resnet_block(sd, "mid.block_2", h, s_temb, block_in, block_in, prec)
    tr("mid.block_2", h)

    for i_level in reversed(range(nres)):
        block_out = cfg.ch * cfg.ch_mult[i_level]
        skip_in = cfg.ch * cfg.ch_mult[i_level]
        for i_block in range(cfg.num_res_blocks + 1):
            if i_block == cfg.num_res_blocks:
                skip_in = cfg.ch * in_ch_mult[i_level]
            h = resnet_block(sd, f"up.{i_level}.block.{i_block}", torch.cat([h, hs.pop()], dim=1), s_temb,
                             block_in + skip_in, block_out, prec)
            tr(f"up.{i_level}.block.{i_block}", h)
            block_in = block_out
            if curr_res in cfg.attn_resolutions:
                h = attn_block(sd, f"up.{i_level}.attn.{i_block}", h, prec)
                tr(f"up.{i_level}.attn.{i_block}", h)
        if i_level != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = prec.act(_conv(sd, f"up.{i_level}.upsample.conv", h, prec, padding=1))
            tr(f"up.{i_level}.upsample", h)
            curr_res *= 2

    h = prec.act(swish(_gn(sd, "norm_out", h)))
    return _conv(sd, "conv_out", h, prec, padding=1)
