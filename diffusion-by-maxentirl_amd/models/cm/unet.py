"""ADM / EDM U-Net (`models.cm.unet.UNetModel`) on the gfx950 kernel library.

Drop-in for the reference class (reference: models/cm/unet.py:523-790): same constructor keywords, same
parameter names and shapes (input_blocks.*, middle_block.*, output_blocks.*, time_embed.*, label_emb,
out.*), so `edm_imagenet64_ema.pt` / `edm_bedroom256_ema.pt`-style state dicts load unchanged; same
forward(x [N,C,H,W] fp32, timesteps [N] float, y [N] long | None) -> [N,out_channels,H,W] fp32.

The torch.nn modules are parameter containers only.  forward() runs a fused HIP program over NHWC
bf16 activations (the reference runs this torso in fp16, use_fp16: True in configs/imagenet64):
  * GroupNorm32(+SiLU) in one or two kernels with fp32 statistics; the FiLM conditioning
    `norm(h) * (1 + scale) + shift` (unet.py:252-256) is folded into the same kernel;
  * every ResBlock's emb_layers (SiLU -> Linear) as ONE linear launch for the whole net;
  * 3x3 / 1x1 MFMA convs with bias, residual and the skip concat (th.cat([h, hs.pop()]), :786) fused;
    ResBlock(up) feeds the conv through its nearest-x2 staging, ResBlock(down) through a 2x2 mean kernel;
  * QKVAttentionLegacy (:401-441; channel layout "(three h d)") as one flash-style MFMA kernel per block.
No CPU path: non-device tensors raise.  With grad enabled forward() goes through models/cm/unet_train.py (HIP backward).
convert_to_fp16/convert_to_fp32 are accepted and change nothing
(storage is bf16, masters fp32).
"""
import math

import torch
import torch.nn as nn

from dxmi_hip import ops
from dxmi_hip._lib import DxmiError
from .nn import conv_nd, linear, normalization, zero_module


class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    pass


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=1)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)


class ResBlock(TimestepBlock):
    """Container of reference :147-226 (indices inside the Sequentials are part of the key names)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint, self.use_scale_shift_norm = use_conv, use_checkpoint, use_scale_shift_norm
        self.up, self.down, self.updown = up, down, up or down
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(
            nn.SiLU(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)


class AttentionBlock(nn.Module):
    """Container of reference :263-308 (legacy attention order; heads = channels // num_head_channels)."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, attention_type="legacy",
                 encoder_channels=None, dims=2, channels_last=False, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            self.num_heads = channels // num_head_channels
        if attention_type != "legacy" or encoder_channels is not None:
            raise NotImplementedError("only the legacy self-attention used by the DxMI configs is implemented")
        self.norm = normalization(channels)
        self.qkv = conv_nd(dims, channels, channels * 3, 1)
        self.proj_out = zero_module(conv_nd(dims, channels, channels, 1))


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(attention_resolutions)
        self.dropout, self.channel_mult, self.conv_resample = dropout, tuple(channel_mult), conv_resample
        self.num_classes, self.use_checkpoint = num_classes, use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.use_scale_shift_norm = use_scale_shift_norm

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)

        def res(cin, cout, **kw):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(c, nh):
            return AttentionBlock(c, use_checkpoint=use_checkpoint, num_heads=nh, num_head_channels=num_head_channels,
                                  use_new_attention_order=use_new_attention_order)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in self.attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, num_heads), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [res(ch + ich, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in self.attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown
                                  else Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, input_ch, out_channels, 3, padding=1)))
        self._packed, self._packed_key = None, None
        self._pack_plan, self._fp32_params = None, False

    def convert_to_fp16(self):
        """Accepted for script compatibility (generate_large.py:129-130): activations are bf16 in the HIP
        program regardless, parameters stay fp32 masters."""

    def convert_to_fp32(self):
        pass

    # ------------------------------------------------------------------ weight fragments
    def _param_key(self):
        return tuple((p.data_ptr(), p._version) for p in ops.fast_parameters(self))

    def _pack(self):
        with ops.pack_batch() as pb:      # every pack below runs in a few multi-tensor launches
            pk = {"te0": ops.pack_conv_weight(self.time_embed[0].weight), "te2": ops.pack_conv_weight(self.time_embed[2].weight)}
            blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
            # the concatenated emb_layers operands live in buffers of their own, refreshed in place by _repack()
            self._emb_w_cat = torch.cat([b.emb_layers[1].weight.detach() for b in blocks], 0).float().contiguous()
            pk["emb_w"] = ops.pack_conv_weight(self._emb_w_cat)
            pk["emb_b"] = torch.cat([b.emb_layers[1].bias.detach() for b in blocks], 0).float().contiguous()
            self._emb_blocks = blocks
            off = 0
            for b in blocks:
                pk[id(b), "eoff"] = off
                off += b.emb_layers[1].out_features
                pk[id(b), "conv1"] = ops.pack_conv_weight(b.in_layers[2].weight)
                pk[id(b), "conv2"] = ops.pack_conv_weight(b.out_layers[3].weight)
                if not isinstance(b.skip_connection, nn.Identity):
                    pk[id(b), "skip"] = ops.pack_conv_weight(b.skip_connection.weight)
            for m in self.modules():
                if isinstance(m, AttentionBlock):
                    pk[id(m), "qkv"] = ops.pack_conv_weight(m.qkv.weight.reshape(3 * m.channels, m.channels, 1, 1))
                    pk[id(m), "proj"] = ops.pack_conv_weight(m.proj_out.weight.reshape(m.channels, m.channels, 1, 1))
                elif isinstance(m, Upsample) and m.use_conv:
                    pk[id(m), "conv"] = ops.pack_conv_weight(m.conv.weight)
                elif isinstance(m, Downsample) and m.use_conv:
                    pk[id(m), "conv"] = ops.pack_conv_weight(m.op.weight)
            conv_in = self.input_blocks[0][0]
            pk["conv_in"] = ops.pack_conv_weight(conv_in.weight, k27=(self.in_channels == 3))
            pk["conv_out"] = ops.pack_conv_weight(self.out[2].weight)
        self._pack_plan = pb.plan
        return pk

    def _repack(self):
        """Parameters updated in place (an optimiser step: same addresses, new versions): the same sources into the same fragment
        buffers, one call (ops.PackPlan)."""
        torch.cat([b.emb_layers[1].weight.detach() for b in self._emb_blocks], 0, out=self._emb_w_cat)
        torch.cat([b.emb_layers[1].bias.detach() for b in self._emb_blocks], 0, out=self._packed["emb_b"])
        self._pack_plan.replay()

    def packed(self):
        key = self._param_key()
        if self._packed is None or key != self._packed_key:
            moved = self._packed is None or getattr(self, "_pack_plan", None) is None or \
                len(key) != len(self._packed_key) or any(a[0] != b[0] for a, b in zip(key, self._packed_key))
            if moved or not self._fp32_params or not ops.PACK_PLAN_REPLAY:
                self._packed = self._pack()
                self._fp32_params = all(p.dtype == torch.float32 and p.is_contiguous() for p in ops.fast_parameters(self))
            else:
                self._repack()
            self._packed_key = key
        return self._packed

    def refresh_packs(self):
        """Bring every packed-weight set of the net that exists up to date with the parameters (no-op when they are)."""
        self.packed()
        if getattr(self, "_packed_t", None) is not None:
            from .unet_train import _pack_t
            _pack_t(self)

    def prepare_capture(self):
        """Before a StepGraph capture (dxmi_hip/graph.py): entries of the transposed set that a backward packs on demand are dropped
        so that the captured step packs them itself; the planned sets are refreshed eagerly (in place: ops.PackPlan)."""
        pkt = getattr(self, "_packed_t", None)
        if pkt is not None:
            for k in [k for k in pkt if k not in self._pack_t_planned]:
                del pkt[k]
        self.refresh_packs()

    # ------------------------------------------------------------------ fused blocks
    # Activations on maps of >= STREAM_GN_MIN_HW pixels travel as (tensor, BlockStats): the conv that produces a tensor writes
    # the GroupNorm statistics of what it stores (per channel pair: the group widths of these nets are 6 ... 32 channels), so
    # GroupNorm32 (+ scale-shift) + SiLU on them is one streaming read + write (ops.groupnorm_apply) instead of the generic
    # statistics + apply pair.  Tensors whose producer writes none (stem, pooled / upsampled x, 1x1 proj_out, 8x8 maps) get
    # them from one extra read (ops.block_stats) when that is cheaper than the generic path, else None = generic path.
    STREAM_GN_MIN_HW = 256

    def _stats_of(self, t):
        return ops.block_stats(t) if t.shape[1] * t.shape[2] >= self.STREAM_GN_MIN_HW and t.shape[3] % 8 == 0 else None

    def _res(self, pk, b, x0, x1, emb_all, s0=None, s1=None):
        gn1, conv1, gn2, conv2 = b.in_layers[0], b.in_layers[2], b.out_layers[0], b.out_layers[3]
        a = ops.groupnorm_silu(x0, gn1.weight, gn1.bias, in1=x1, eps=gn1.eps, silu=True, stats=(s0, s1))
        xs = x0
        if b.up:
            assert x1 is None
            xs = ops.upsample2x(x0)                      # x_upd (:247); h_upd is the conv's nearest-x2 staging
        elif b.down:
            assert x1 is None
            a, xs = ops.pool_act(a, True, ops.ACT_NONE), ops.pool_act(x0, True, ops.ACT_NONE)
        off, eo = pk[id(b), "eoff"], b.emb_layers[1].out_features
        e = emb_all[:, off:off + eo]
        if b.use_scale_shift_norm:
            h, sh = ops.conv2d(a, pk[id(b), "conv1"], bias=conv1.bias, upsample=b.up, want_stats=True)
            a = ops.groupnorm_silu(h, gn2.weight, gn2.bias, eps=gn2.eps, silu=True, scale_shift=e, stats=(sh, None))
        else:
            h, sh = ops.conv2d(a, pk[id(b), "conv1"], bias=conv1.bias, upsample=b.up, addvec=e, want_stats=True)
            a = ops.groupnorm_silu(h, gn2.weight, gn2.bias, eps=gn2.eps, silu=True, stats=(sh, None))
        if (id(b), "skip") in pk:
            sk = b.skip_connection
            xs = ops.conv2d(x0, pk[id(b), "skip"], in1=x1, bias=sk.bias)
        out, so = ops.conv2d(a, pk[id(b), "conv2"], bias=conv2.bias, residual=xs, want_stats=True)
        return out, (so if so is not None else self._stats_of(out))

    def _attn(self, pk, m, x, sx=None):
        N, H, W, C = x.shape
        hn = ops.groupnorm_silu(x, m.norm.weight, m.norm.bias, eps=m.norm.eps, silu=False, stats=(sx, None))
        qkv = ops.conv2d(hn, pk[id(m), "qkv"], bias=m.qkv.bias)
        ch = C // m.num_heads
        a = ops.attention(qkv.view(N, H * W, 3 * C), heads=m.num_heads, scale=1.0 / math.sqrt(ch))
        out = ops.conv2d(a.view(N, H, W, C), pk[id(m), "proj"], bias=m.proj_out.bias, residual=x)
        return out, self._stats_of(out)

    def _seq(self, pk, seq, h, skip, emb_all, sh=None, sskip=None):
        """-> (h, BlockStats | None)"""
        for m in seq:
            if isinstance(m, ResBlock):
                h, sh = self._res(pk, m, h, skip, emb_all, s0=sh, s1=sskip)
                skip = sskip = None
            elif isinstance(m, AttentionBlock):
                h, sh = self._attn(pk, m, h, sh)
            elif isinstance(m, Downsample):
                if m.use_conv:
                    h, sh = ops.conv2d(h, pk[id(m), "conv"], bias=m.op.bias, stride=2, pad=1, want_stats=True)
                else:
                    h, sh = ops.pool_act(h, True, ops.ACT_NONE), None
                sh = sh if sh is not None else self._stats_of(h)
            elif isinstance(m, Upsample):
                if m.use_conv:
                    h, sh = ops.conv2d(h, pk[id(m), "conv"], bias=m.conv.bias, upsample=True, want_stats=True)
                else:
                    h, sh = ops.upsample2x(h), None
                sh = sh if sh is not None else self._stats_of(h)
            else:
                raise DxmiError(f"unexpected layer {type(m).__name__}")
        return h, sh

    # ------------------------------------------------------------------ forward
    def forward(self, x, timesteps, y=None):
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if not x.is_cuda:
            raise DxmiError("models.cm.unet.UNetModel runs only on the HIP device path (no CPU fallback)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in ops.fast_parameters(self))):
            from .unet_train import forward_with_grad  # autograd wrapper around the HIP kernels
            return forward_with_grad(self, x, timesteps, y)
        return self.forward_inference(x, timesteps, y)

    @torch.no_grad()
    def forward_inference(self, x, timesteps, y=None, trace=None):
        tr = (lambda n, v: trace.append((n, v))) if trace is not None else (lambda n, v: None)
        pk = self.packed()
        x = x.contiguous().float()
        emb = ops.timestep_embedding(timesteps, self.model_channels, order=1)
        emb = ops.linear(emb, pk["te0"], self.time_embed[0].bias, post_act=ops.ACT_SILU)
        emb = ops.linear(emb, pk["te2"], self.time_embed[2].bias)
        if self.num_classes is not None:
            assert y.shape == (x.shape[0],)
            emb = emb + self.label_emb.weight[y]
        tr("emb", emb)
        emb_all = ops.linear(emb, pk["emb_w"], pk["emb_b"], pre_act=ops.ACT_SILU)

        conv_in = self.input_blocks[0][0]
        if pk["conv_in"].k27:
            h = ops.conv2d(x, pk["conv_in"], bias=conv_in.bias)
        else:
            h = ops.conv2d(ops.nchw_f32_to_nhwc_bf16(x), pk["conv_in"], bias=conv_in.bias)
        sh = self._stats_of(h)
        hs = [(h, sh)]
        tr("input_blocks.0", h)
        for i in range(1, len(self.input_blocks)):
            h, sh = self._seq(pk, self.input_blocks[i], h, None, emb_all, sh=sh)
            hs.append((h, sh))
            tr(f"input_blocks.{i}", h)
        h, sh = self._seq(pk, self.middle_block, h, None, emb_all, sh=sh)
        tr("middle_block", h)
        for i, blk in enumerate(self.output_blocks):
            skip, sskip = hs.pop()
            h, sh = self._seq(pk, blk, h, skip, emb_all, sh=sh, sskip=sskip)   # the concat is never materialised
            tr(f"output_blocks.{i}", h)
        gn = self.out[0]
        a = ops.groupnorm_silu(h, gn.weight, gn.bias, eps=gn.eps, silu=True, stats=(sh, None))
        return ops.conv2d(a, pk["conv_out"], bias=self.out[2].bias, out_nchw_f32=True)
