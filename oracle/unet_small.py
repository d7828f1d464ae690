"""Oracle: DDPM U-Net forward as a functional torch-CPU program (TEST INFRASTRUCTURE).

Restates models/DxMI/unet_small.py (reference root) over a flat state dict `sd`
(name -> tensor, reference key names), so no reference code is needed at run time:
  timestep embedding  unet_small.py:9-27     temb MLP            :296-299
  ResnetBlock         :117-136               AttnBlock           :167-191
  Downsample          :69-76                 Upsample            :50-54
  Model.forward       :292-332
`prec` (oracle.Precision) chooses reference fp32 arithmetic or the bf16 storage model of the HIP
pipeline.  Dropout (nn.Dropout(p) between norm2+swish and conv2, unet_small.py:129): torch's device RNG is
not reproducible across devices, so train-mode parity INJECTS the masks: `dropout_masks` maps a block prefix to
a [B,C,H,W] keep-mask (0/1) and `dropout_p` scales kept values by 1/(1-p).  `dropout_keep_mask` restates the
HIP kernel's counter hash (include/dxmi_hip.h, dxmi_dropout_bf16) in numpy integer arithmetic (bit-exact).
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


def dropout_keep_mask(shape_nchw, p, seed):
    """keep(i) = (mix32(i ^ seed) >> 8) >= p * 2^24 over the NHWC linear index i of the activation (the HIP pipeline's
    layout), returned as a float [B,C,H,W] 0/1 tensor."""
    import numpy as np
    B, C, H, W = shape_nchw
    h = np.arange(B * H * W * C, dtype=np.uint64) ^ np.uint64(seed & 0xFFFFFFFF)
    m = np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7feb352d)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846ca68b)) & m
    h ^= h >> np.uint64(16)
    keep = (h >> np.uint64(8)) >= np.uint64(int(float(p) * 16777216.0))
    return torch.from_numpy(keep.reshape(B, H, W, C).astype(np.float32)).permute(0, 3, 1, 2).contiguous()


def _conv(sd, name, x, prec, stride=1, padding=0):
    return F.conv2d(x, prec.w(sd[name + ".weight"]), sd[name + ".bias"], stride=stride, padding=padding)


def _gn(sd, name, x, eps=1e-6):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps)


def resnet_block(sd, pre, x, s_temb, cin, cout, prec, keep=None, p_drop=0.0):
    """unet_small.py:117-136.  `s_temb` = swish(temb) (shared by all blocks).  x is a stored
    activation (already prec.act-rounded)."""
    h = prec.act(swish(_gn(sd, pre + ".norm1", x)))
    h = _conv(sd, pre + ".conv1", h, prec, padding=1)
    tproj = F.linear(prec.act(s_temb), prec.w(sd[pre + ".temb_proj.weight"]), sd[pre + ".temb_proj.bias"])
    h = prec.act(h + tproj[:, :, None, None])
    h = prec.act(swish(_gn(sd, pre + ".norm2", h)))
    if keep is not None:
        h = prec.act(h * (keep * (1.0 / (1.0 - p_drop))))
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


def forward(sd, cfg, x, t, prec=None, trace=None, dropout_masks=None, dropout_p=0.0):
    """Model.forward, unet_small.py:292-332.  x [B,C,H,W] fp32, t [B] float.
    trace: optional list that receives (name, NCHW tensor) after every block (debugging aid)."""
    tr = (lambda n, v: trace.append((n, v.clone()))) if trace is not None else (lambda n, v: None)
    prec = prec or Precision("fp32")
    dm = dropout_masks or {}
    rb = lambda pre, xx, cin, cout: resnet_block(sd, pre, xx, s_temb, cin, cout, prec, dm.get(pre), dropout_p)
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
            h = rb(f"down.{i_level}.block.{i_block}", hs[-1], block_in, block_out)
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
    h = rb("mid.block_1", h, block_in, block_in)
    tr("mid.block_1", h)
    h = attn_block(sd, "mid.attn_1", h, prec)
    tr("mid.attn_1", h)
    h = rb("mid.block_2", h, block_in, block_in)
    tr("mid.block_2", h)

    for i_level in reversed(range(nres)):
        block_out = cfg.ch * cfg.ch_mult[i_level]
        skip_in = cfg.ch * cfg.ch_mult[i_level]
        for i_block in range(cfg.num_res_blocks + 1):
            if i_block == cfg.num_res_blocks:
                skip_in = cfg.ch * in_ch_mult[i_level]
            h = rb(f"up.{i_level}.block.{i_block}", torch.cat([h, hs.pop()], dim=1), block_in + skip_in, block_out)
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
