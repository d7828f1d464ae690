"""DDPM U-Net (`models.DxMI.unet_small.Model`) on the gfx950 kernel library.

Drop-in for the reference class of the same dotted name (reference: models/DxMI/unet_small.py:194-332):
same constructor keywords, same parameter names/shapes (state dicts load both ways, including the
`log_betas` parameter and `std` buffer that VARSampler hangs on the net), same
forward(x [B,C,H,W] fp32, t [B] float) -> [B,out_ch,H,W] fp32 contract.

The torch.nn layers below are parameter CONTAINERS only (fp32 master weights, reference
initialisation order).  forward() never calls them: it runs a fused HIP program —
  GroupNorm+SiLU (1 kernel) -> 3x3 MFMA conv [+bias +temb +residual] ; fused q|k|v 1x1 conv ->
  MFMA attention -> 1x1 conv + residual ; all 22 temb_proj layers as ONE linear launch —
over NHWC bf16 activations, from bf16 weight fragments that are re-packed only when a
parameter's version counter changes.  No CPU path: non-device tensors raise.
"""
import torch
import torch.nn as nn

from dxmi_hip import ops
from dxmi_hip._lib import DxmiError


def Normalize(in_channels):
    """GroupNorm(32, eps=1e-6, affine) container (reference :35-36)."""
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def _conv(cin, cout, k, stride=1, padding=0):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=padding)


class Upsample(nn.Module):
    """nearest x2 then 3x3 conv (reference :39-54); the upsample is fused into the conv's staging."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = _conv(in_channels, in_channels, 3, 1, 1)


class Downsample(nn.Module):
    """zero pad (0,1,0,1) then 3x3 stride-2 conv (reference :57-76)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = _conv(in_channels, in_channels, 3, 2, 0)


class ResnetBlock(nn.Module):
    """Parameter container of the reference block (:79-136)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = _conv(in_channels, out_channels, 3, 1, 1)
        self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = _conv(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = _conv(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = _conv(in_channels, out_channels, 1, 1, 0)


class AttnBlock(nn.Module):
    """Parameter container of the single-head attention block (:139-191)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = _conv(in_channels, in_channels, 1)
        self.k = _conv(in_channels, in_channels, 1)
        self.v = _conv(in_channels, in_channels, 1)
        self.proj_out = _conv(in_channels, in_channels, 1)


class _Normed:
    """A tensor's GroupNorm(+SiLU) for one specific reader, written by its producer's epilogue (ops.conv2d fuse_gn)."""
    __slots__ = ("y", "norm")

    def __init__(self, y, norm):
        self.y, self.norm = y, norm


class Model(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution):
        super().__init__()
        ch_mult = tuple(ch_mult)  # YAML loaders hand over lists
        self.ch, self.temb_ch = ch, ch * 4
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.out_ch = resolution, in_channels, out_ch
        self.ch_mult, self.attn_resolutions = ch_mult, tuple(attn_resolutions)
        self.dropout_p = dropout
        if not resamp_with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by any DxMI config")

        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = _conv(in_channels, ch, 3, 1, 1)

        curr_res = resolution
        in_ch_mult = (1,) + ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            level = nn.Module()
            level.block, level.attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                level.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out,
                                               temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in self.attn_resolutions:
                    level.attn.append(AttnBlock(block_in))
            if i_level != self.num_resolutions - 1:
                level.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(level)

        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)

        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            level = nn.Module()
            level.block, level.attn = nn.ModuleList(), nn.ModuleList()
            block_out = skip_in = ch * ch_mult[i_level]
            for i_block in range(num_res_blocks + 1):
                if i_block == num_res_blocks:
                    skip_in = ch * in_ch_mult[i_level]
                level.block.append(ResnetBlock(in_channels=block_in + skip_in, out_channels=block_out,
                                               temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in self.attn_resolutions:
                    level.attn.append(AttnBlock(block_in))
            if i_level != 0:
                level.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, level)

        self.norm_out = Normalize(block_in)
        self.conv_out = _conv(block_in, out_ch, 3, 1, 1)
        self._packed = None
        self._packed_key = None
        self._pack_bufs = None
        self.dropout_seed = None          # base seed of the dropout hash (None: torch.initial_seed() at first use)
        self.dropout_seeds_used = []      # seeds of the most recent training forward, in _resblocks() call order
        self._dropout_calls = 0

    def _next_dropout_seed(self):
        """32-bit seed of the next dropout site: hash of (base seed, running call counter).  Deliberately NOT drawn from
        torch's CPU generator, whose stream the trainer's randperm parity depends on."""
        base = torch.initial_seed() if self.dropout_seed is None else self.dropout_seed
        x = (base * 0x9E3779B97F4A7C15 + self._dropout_calls * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 32
        x = (x * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 32
        self._dropout_calls += 1
        seed = x & 0xFFFFFFFF
        self.dropout_seeds_used.append(seed)
        return seed

    # ------------------------------------------------------------------ weight fragments
    def _resblocks(self):
        for lvl in self.down:
            yield from lvl.block
        yield self.mid.block_1
        yield self.mid.block_2
        for lvl in reversed(self.up):
            yield from lvl.block

    def _param_key(self):
        # torch bumps _version on every in-place update (optimizer.step, load_state_dict, .to())
        return tuple((p.data_ptr(), p._version) for p in ops.fast_parameters(self))

    def _pack(self, reuse=None):
        """(Re)build the bf16 MFMA weight fragments from the fp32 master parameters.  reuse: the buffers of the previous pack
        (parameters updated in place): the fragments are rewritten where they are (ops.pack_batch)."""
        with ops.pack_batch(reuse=reuse) as pb:            # every pack below runs in a few multi-tensor launches
            pk = {}
            dev = self.conv_in.weight.device
            f32 = torch.float32
            pk["dense0"] = ops.pack_conv_weight(self.temb.dense[0].weight)
            pk["dense1"] = ops.pack_conv_weight(self.temb.dense[1].weight)
            blocks = list(self._resblocks())
            # every temb_proj of the net in ONE [sum(Cout), temb_ch] operator (reference :123)
            pk["tproj"] = ops.pack_conv_weight(torch.cat([b.temb_proj.weight for b in blocks], 0))
            pk["tproj_bias"] = torch.cat([b.temb_proj.bias.detach() for b in blocks], 0,
                                         out=ops.pack_tensor(sum(b.out_channels for b in blocks), f32, dev))
            off = 0
            for b in blocks:
                pk[id(b), "toff"] = off
                off += b.out_channels
                pk[id(b), "conv1"] = ops.pack_conv_weight(b.conv1.weight)
                pk[id(b), "conv2"] = ops.pack_conv_weight(b.conv2.weight)
                if b.in_channels != b.out_channels:
                    sc = b.conv_shortcut if b.use_conv_shortcut else b.nin_shortcut
                    pk[id(b), "short"] = ops.pack_conv_weight(sc.weight)
            for m in self.modules():
                if isinstance(m, AttnBlock):
                    pk[id(m), "qkv"] = ops.pack_conv_weight(torch.cat([m.q.weight, m.k.weight, m.v.weight], 0))
                    pk[id(m), "qkv_bias"] = torch.cat([m.q.bias.detach(), m.k.bias.detach(), m.v.bias.detach()], 0,
                                                      out=ops.pack_tensor(3 * m.in_channels, f32, dev))
                    pk[id(m), "proj"] = ops.pack_conv_weight(m.proj_out.weight)
                    if m.in_channels == 256:      # the 16x16 blocks: proj_out fused behind the attention (ops.attention_proj)
                        pk[id(m), "proj_attn"] = ops.pack_attn_proj_weight(m.proj_out.weight)
                        # ... and the whole block as one launch (round 5: ops.attn_block, folded Wk^T Wq / Wproj Wv)
                        pk[id(m), "block"] = ops.attn_block_pack(m.q.weight, m.q.bias, m.k.weight, m.v.weight, m.v.bias,
                                                                 m.proj_out.weight, m.proj_out.bias, float(int(m.in_channels) ** (-0.5)))
                elif isinstance(m, (Upsample, Downsample)):
                    pk[id(m), "conv"] = ops.pack_conv_weight(m.conv.weight)
            pk["conv_in"] = ops.pack_conv_weight(self.conv_in.weight, k27=(self.in_channels == 3))
            pk["conv_out"] = ops.pack_conv_weight(self.conv_out.weight)
        assert dev.type == "cuda"
        self._pack_bufs = pb.buffers
        return pk

    @staticmethod
    def _same_storage(key, old_key):
        return old_key is not None and len(key) == len(old_key) and all(a[0] == b[0] for a, b in zip(key, old_key))

    def packed(self):
        key = self._param_key()
        if self._packed is None or key != self._packed_key:
            # parameters updated in place (an optimiser step): the same buffers are rewritten, so every address a captured
            # hipGraph holds stays valid (dxmi_hip/graph.py); moved parameters (.to(), a new tensor): fresh buffers
            reuse = self._pack_bufs if (self._packed is not None and self._same_storage(key, self._packed_key)) else None
            self._packed, self._packed_key = self._pack(reuse=reuse), key
        return self._packed

    def refresh_packs(self):
        """Bring every packed-weight set of the net that exists up to date with the parameters (no-op when they are)."""
        self.packed()
        if getattr(self, "_packed_t", None) is not None:
            from .unet_small_train import _pack_t
            _pack_t(self)

    def prepare_capture(self):
        """Before a StepGraph capture: on-demand entries of the transposed set (packed by a backward when first needed) are
        dropped so that the captured step packs them itself, then everything else is refreshed eagerly."""
        if getattr(self, "_packed_t", None) is not None:
            for k in [k for k in self._packed_t if k not in self._packed_t_planned]:
                del self._packed_t[k]
        self.refresh_packs()

    # ------------------------------------------------------------------ fused blocks
    # Activations on maps of >= STREAM_GN_MIN_HW pixels travel as (tensor, BlockStats): the producing conv's epilogue writes the
    # GroupNorm block statistics of what it stores, so every Normalize() on them is one streaming read + write
    # (ops.groupnorm_apply).  Smaller maps (8x8, 4x4: 8 % of the GroupNorm bytes) keep the one-pass resident kernel.
    FUSE_ATTN_PROJ = True             # proj_out + residual of the 16x16 AttnBlocks behind the attention kernel
    FUSE_ATTN_BLOCK = True            # the 16x16 AttnBlocks as ONE launch (norm, q|k|v, attention, proj_out, residual: ops.attn_block)
    FUSE_GN_SMALL = True              # norm2 of the 4x4 ResnetBlocks from conv1's epilogue (instance override: A-B timing)
    STREAM_GN_MIN_HW = 256            # instance attribute override (tests / A-B timing): 1 << 30 = one-pass GroupNorm everywhere

    def _conv_s(self, x, pw, **kw):
        """conv2d -> (out, BlockStats | None): statistics from the conv's own epilogue where its kernel writes them, else from
        one extra read of the output (stem conv, stride-2 Downsample, 1x1 proj_out)."""
        out, st = ops.conv2d(x, pw, want_stats=True, **kw)
        if st is None and out.shape[1] * out.shape[2] >= self.STREAM_GN_MIN_HW:
            st = ops.block_stats(out)
        return out, st

    def _resblock(self, pk, b, x0, x1, tp, s0=None, s1=None, nxt=None):
        """GN+SiLU -> conv1(+bias+temb) -> GN+SiLU -> conv2(+bias+shortcut); x = [x0 | x1] virtual concat; s0 / s1 their block
        statistics (None: one-pass GroupNorm) or, for s0, a _Normed(x0) some producer's epilogue already wrote for norm1.
        nxt = (GroupNorm module, silu) of the ONLY-normalising next reader of the output: where conv2's kernel can, it writes that
        normalisation too.  Returns (h, BlockStats | _Normed | None)."""
        if isinstance(s0, _Normed):
            assert s0.norm is b.norm1 and x1 is None
            a0 = s0.y
        else:
            a0 = ops.groupnorm_silu(x0, b.norm1.weight, b.norm1.bias, in1=x1, eps=1e-6, silu=True, stats=(s0, s1))
        off = pk[id(b), "toff"]
        stream = x0.shape[1] * x0.shape[2] >= self.STREAM_GN_MIN_HW
        a = None
        if not stream and self.FUSE_GN_SMALL:
            # 4x4 maps: conv1's epilogue normalises its own output (whole images and groups per tile); h has no other reader.
            # Where the kernel for the shape cannot, a is None
            h, a = ops.conv2d(a0, pk[id(b), "conv1"], bias=b.conv1.bias, addvec=tp[:, off:off + b.out_channels],
                              fuse_gn=(b.norm2.weight, b.norm2.bias, 32, 1e-6, True, False))
            sh = None
            if a is None and stream:
                sh = ops.block_stats(h)
        else:
            h, sh = ops.conv2d(a0, pk[id(b), "conv1"], bias=b.conv1.bias, addvec=tp[:, off:off + b.out_channels], want_stats=True)
        if a is None:
            a = ops.groupnorm_silu(h, b.norm2.weight, b.norm2.bias, eps=1e-6, silu=True, stats=(sh, None))
        if b.in_channels != b.out_channels:
            sc_mod = b.conv_shortcut if b.use_conv_shortcut else b.nin_shortcut
            sc = ops.conv2d(x0, pk[id(b), "short"], in1=x1, bias=sc_mod.bias)
        else:
            assert x1 is None
            sc = x0
        if stream:
            return self._conv_s(a, pk[id(b), "conv2"], bias=b.conv2.bias, residual=sc)
        if nxt is not None and self.FUSE_GN_SMALL:
            h, y = ops.conv2d(a, pk[id(b), "conv2"], bias=b.conv2.bias, residual=sc,
                              fuse_gn=(nxt[0].weight, nxt[0].bias, 32, 1e-6, nxt[1], True))
            return h, (None if y is None else _Normed(y, nxt[0]))
        return ops.conv2d(a, pk[id(b), "conv2"], bias=b.conv2.bias, residual=sc), None

    def _attn(self, pk, m, x, sx=None):
        N, H, W, C = x.shape
        if self.FUSE_ATTN_BLOCK and (id(m), "block") in pk and isinstance(sx, ops.BlockStats) and ops.attn_block_supported(H * W, C, 1):
            return ops.attn_block(x, sx, m.norm.weight, m.norm.bias, pk[id(m), "block"], eps=1e-6, want_stats=True)
        if isinstance(sx, _Normed):
            assert sx.norm is m.norm
            hn = sx.y
        else:
            hn = ops.groupnorm_silu(x, m.norm.weight, m.norm.bias, eps=1e-6, silu=False, stats=(sx, None))
        qkv = ops.conv2d(hn, pk[id(m), "qkv"], bias=pk[id(m), "qkv_bias"])
        if self.FUSE_ATTN_PROJ and (id(m), "proj_attn") in pk and ops.attention_proj_supported(H * W, C, 1):
            out, st = ops.attention_proj(qkv.view(N, H * W, 3 * C), pk[id(m), "proj_attn"], m.proj_out.bias, x, heads=1,
                                         scale=float(int(C) ** (-0.5)), want_stats=True)
            return out.view(N, H, W, C), (st if H * W >= self.STREAM_GN_MIN_HW else None)
        a = ops.attention(qkv.view(N, H * W, 3 * C), heads=1, scale=float(int(C) ** (-0.5)))
        if H * W >= self.STREAM_GN_MIN_HW:
            return self._conv_s(a.view(N, H, W, C), pk[id(m), "proj"], bias=m.proj_out.bias, residual=x)
        return ops.conv2d(a.view(N, H, W, C), pk[id(m), "proj"], bias=m.proj_out.bias, residual=x), None

    # ------------------------------------------------------------------ forward
    def forward(self, x, t, temb_rows=None):
        assert x.shape[2] == x.shape[3] == self.resolution
        if not x.is_cuda:
            raise DxmiError("models.DxMI.unet_small.Model runs only on the HIP device path (no CPU fallback)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in ops.fast_parameters(self))):
            from .unet_small_train import forward_with_grad  # autograd wrapper around the HIP kernels
            return forward_with_grad(self, x, t)
        return self.forward_inference(x, t, temb_rows=temb_rows)

    @torch.no_grad()
    def temb_table(self, t):
        """The timestep branch of forward() — embedding, the two dense layers, every block's temb_proj (reference
        unet_small.py:306-309 and :123) — for a vector of timesteps: [len(t), sum of block widths].  The branch depends on t
        only, and every output row depends on its own input row only, so a generation loop whose batch shares one timestep
        per step evaluates it on T rows once instead of on T x B identical rows (VARSampler.sample); the rows are bitwise
        the ones forward() computes itself."""
        pk = self.packed()
        emb = ops.timestep_embedding(t, self.ch, order=0)
        h1 = ops.linear(emb, pk["dense0"], self.temb.dense[0].bias, post_act=ops.ACT_SILU)
        s_temb = ops.linear(h1, pk["dense1"], self.temb.dense[1].bias, post_act=ops.ACT_SILU)  # swish(temb)
        return ops.linear(s_temb, pk["tproj"], pk["tproj_bias"])

    @torch.no_grad()
    def forward_inference(self, x, t, trace=None, temb_rows=None):
        """trace: optional list receiving (name, NHWC bf16 tensor) after every block (debugging aid).
        temb_rows: a row of temb_table() for this call's (batch-uniform) timestep, shape [1, W] or [B, W] with any row stride."""
        tr = (lambda n, v: trace.append((n, v))) if trace is not None else (lambda n, v: None)
        pk = self.packed()
        x = x.contiguous().float()
        if temb_rows is None:
            emb = ops.timestep_embedding(t, self.ch, order=0)
            h1 = ops.linear(emb, pk["dense0"], self.temb.dense[0].bias, post_act=ops.ACT_SILU)
            s_temb = ops.linear(h1, pk["dense1"], self.temb.dense[1].bias, post_act=ops.ACT_SILU)  # swish(temb)
            tp = ops.linear(s_temb, pk["tproj"], pk["tproj_bias"])
            tr("s_temb", s_temb)
        else:
            tp = temb_rows.expand(x.shape[0], -1)      # row stride 0 when one row serves the whole batch

        if pk["conv_in"].k27:
            h, sh = self._conv_s(x, pk["conv_in"], bias=self.conv_in.bias)
        else:
            h, sh = self._conv_s(ops.nchw_f32_to_nhwc_bf16(x), pk["conv_in"], bias=self.conv_in.bias)
        hs = [(h, sh)]
        tr("conv_in", h)
        for i_level, lvl in enumerate(self.down):
            for i_block, blk in enumerate(lvl.block):
                # the output's next normalising reader (the skip connection reads it raw): the level's next block, or
                # mid.block_1 after the last level
                nxt = None
                if len(lvl.attn) == 0:
                    if i_block + 1 < len(lvl.block):
                        nxt = (lvl.block[i_block + 1].norm1, True)
                    elif i_level == self.num_resolutions - 1:
                        nxt = (self.mid.block_1.norm1, True)
                h, sh = self._resblock(pk, blk, hs[-1][0], None, tp, s0=hs[-1][1], nxt=nxt)
                tr(f"down.{i_level}.block.{i_block}", h)
                if len(lvl.attn) > 0:
                    h, sh = self._attn(pk, lvl.attn[i_block], h, sh)
                    tr(f"down.{i_level}.attn.{i_block}", h)
                hs.append((h, sh))
            if i_level != self.num_resolutions - 1:
                ds = lvl.downsample
                hs.append(self._conv_s(hs[-1][0], pk[id(ds), "conv"], bias=ds.conv.bias, stride=2, pad=0, pad_br=1))
                tr(f"down.{i_level}.downsample", hs[-1][0])

        h, sh = hs[-1]
        h, sh = self._resblock(pk, self.mid.block_1, h, None, tp, s0=sh, nxt=(self.mid.attn_1.norm, False))
        tr("mid.block_1", h)
        h, sh = self._attn(pk, self.mid.attn_1, h, sh)
        tr("mid.attn_1", h)
        h, sh = self._resblock(pk, self.mid.block_2, h, None, tp, s0=sh)
        tr("mid.block_2", h)

        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block, blk in enumerate(lvl.block):
                skip, sskip = hs.pop()
                if isinstance(sskip, _Normed):      # normalised for the down path's reader: the concat's norm1 takes the raw tensor
                    sskip = None
                h, sh = self._resblock(pk, blk, h, skip, tp, s0=sh, s1=sskip)  # cat(h, skip) is never materialised
                tr(f"up.{i_level}.block.{i_block}", h)
                if len(lvl.attn) > 0:
                    h, sh = self._attn(pk, lvl.attn[i_block], h, sh)
                    tr(f"up.{i_level}.attn.{i_block}", h)
            if i_level != 0:
                us = lvl.upsample
                h, sh = self._conv_s(h, pk[id(us), "conv"], bias=us.conv.bias, upsample=True)
                tr(f"up.{i_level}.upsample", h)

        a = ops.groupnorm_silu(h, self.norm_out.weight, self.norm_out.bias, eps=1e-6, silu=True, stats=(sh, None))
        return ops.conv2d(a, pk["conv_out"], bias=self.conv_out.bias, out_nchw_f32=True)
