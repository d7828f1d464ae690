"""Autograd path of the ADM / EDM U-Net on the gfx950 kernels (policy step of DxMI on the EDM backbones:
models/DxMI/trainer.py:693-746 back-propagates the sampler loss through OpenAIDiffusion.sample_step into every
U-Net parameter; reference graph = torch autograd over models/cm/unet.py:761-790).

One torch.autograd.Function wraps the network.  forward() is the inference program of models/cm/unet.py (this
package) plus a tape of saved NHWC bf16 activations; backward() walks the tape in reverse:
  conv data gradients   = the forward MFMA kernels on transpose-flipped weight fragments (ResBlock(up) / Upsample:
                          full-resolution gradient then a 2x2 sum; stride-2 Downsample: zero-stuffed gradient);
  conv weight gradients = MFMA pixel-GEMM (dxmi_conv2d_wgrad), bias gradients = column sums;
  GroupNorm32 (+FiLM scale-shift, +SiLU) = dxmi_groupnorm_generic_bwd, which also returns the per-image sums the
                          scale / shift (= emb_layers output) gradients are formed from;
  attention             = batched MFMA GEMMs + softmax backward (dxmi_bgemm_bf16), any head count;
  mean pool / nearest x2 = each other's transposes (dxmi_upsample2x, dxmi_pool_act);
  time_embed, label_emb, emb_layers = tiny dense layers, re-evaluated and differentiated with torch fp32 matmuls.
Dropout must be 0 (every DxMI EDM config sets dropout: 0.0).  Parameter gradients come back in net.parameters() order.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from dxmi_hip import ops


def _pack_t(net):
    key = net._param_key()
    if getattr(net, "_packed_t", None) is not None and net._packed_t_key == key:
        return net._packed_t
    from .unet import AttentionBlock, Downsample, ResBlock, Upsample
    old = getattr(net, "_packed_t", None)
    if ops.PACK_PLAN_REPLAY and old is not None and getattr(net, "_pack_t_plan", None) is not None and net._fp32_params and len(key) == len(net._packed_t_key) \
            and all(a[0] == b[0] for a, b in zip(key, net._packed_t_key)):
        # parameters updated in place: the same sources into the same buffers (ops.PackPlan); the padded conv_out source and the
        # concatenated emb_layers operand are refreshed in place first
        w = net.out[2].weight
        net._wpad_t[: w.shape[0]] = w.detach()
        if "emb_t" in old:
            torch.cat([b.emb_layers[1].weight.detach() for b in net._emb_blocks], 0, out=net._emb_w_cat_t)
            net._pack_t_plan2.replay()
        net._pack_t_plan.replay()
        for k in [k for k in old if k not in net._pack_t_planned]:
            del old[k]                      # packed on demand by the backward (conv_in_t, the per-source skip_t pairs): stale now
        net._packed_t_key = key
        return old
    pk = {}
    with ops.pack_batch() as pb:      # a few multi-tensor launches instead of one per layer
        for m in net.modules():
            if isinstance(m, ResBlock):
                pk[id(m), "conv1"] = ops.pack_conv_weight(m.in_layers[2].weight, transpose_flip=True)
                pk[id(m), "conv2"] = ops.pack_conv_weight(m.out_layers[3].weight, transpose_flip=True)
            elif isinstance(m, AttentionBlock):
                C = m.channels
                pk[id(m), "qkv"] = ops.pack_conv_weight(m.qkv.weight.reshape(3 * C, C, 1, 1), transpose_flip=True)
                pk[id(m), "proj"] = ops.pack_conv_weight(m.proj_out.weight.reshape(C, C, 1, 1), transpose_flip=True)
            elif isinstance(m, Upsample) and m.use_conv:
                pk[id(m), "conv"] = ops.pack_conv_weight(m.conv.weight, transpose_flip=True)
            elif isinstance(m, Downsample) and m.use_conv:
                pk[id(m), "conv"] = ops.pack_conv_weight(m.op.weight, transpose_flip=True)
        w = net.out[2].weight
        wpad = torch.zeros((64,) + tuple(w.shape[1:]), dtype=torch.float32, device=w.device)
        wpad[: w.shape[0]] = w.detach()
        net._wpad_t = wpad
        pk["conv_out"] = ops.pack_conv_weight(wpad, transpose_flip=True)
    net._pack_t_plan, net._pack_t_plan2 = pb.plan, None
    net._pack_t_planned = set(pk.keys()) | {"emb_t", "te2_t"}      # what the two plans refresh
    net._packed_t, net._packed_t_key = pk, key
    return pk


# Training forward: GroupNorm on the producers' block statistics (streaming apply) where the inference path uses them; 0 = the generic
# statistics + apply pair everywhere (rounds 3-5).
STREAM_GN = os.environ.get("DXMI_TRAIN_STREAM_GN", "1") == "1"
# Training forward keeps the attention rows' log-sum-exp for the backward (dxmi_attention_fwd_lse / _bwd_lse); 0 = the backward recomputes it.
SAVE_LSE = os.environ.get("DXMI_ATTN_SAVE_LSE", "1") == "1"


def _up_sum(g):
    """transpose of nearest x2: sum over each 2x2 block (mean * 4, exact in bf16)."""
    out = ops.pool_act(g, True, ops.ACT_NONE)
    out.mul_(4.0)
    return out


def _pool_t(g):
    """transpose of the 2x2 mean pool: every gradient value to its 4 sources, times 1/4."""
    out = ops.upsample2x(g)
    out.mul_(0.25)
    return out


class _EDMUNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, timesteps, y, *params):
        from .unet import AttentionBlock, Downsample, ResBlock, Upsample
        if net.training and net.dropout:
            raise NotImplementedError("UNetModel HIP training path: dropout > 0 is not implemented (DxMI configs use 0.0)")
        pk = net.packed()
        x = x.contiguous().float()
        sinus = ops.timestep_embedding(timesteps, net.model_channels, order=1)
        e0 = ops.linear(sinus, pk["te0"], net.time_embed[0].bias, post_act=ops.ACT_SILU)
        emb = ops.linear(e0, pk["te2"], net.time_embed[2].bias)
        if net.num_classes is not None:
            emb = emb + net.label_emb.weight[y]
        emb_all = ops.linear(emb, pk["emb_w"], pk["emb_b"], pre_act=ops.ACT_SILU)
        tape = []

        def conv_s(*a, **kw):
            """-> (out, BlockStats of out from the conv's epilogue | None)"""
            return ops.conv2d(*a, want_stats=True, **kw) if STREAM_GN else (ops.conv2d(*a, **kw), None)

        def gn_fwd(norm, xin, st=None, st1=None, **kw):
            """GroupNorm(+SiLU) + the statistics partials the backward reads (it then skips its own statistics pass over `xin`;
            autograd saves mean / rstd the same way).  With block statistics from the producer (st; round 6, as in
            forward_inference) the forward is the one-pass streaming apply and the partials are those sums converted
            (dxmi_gn_blockstats_to_generic); otherwise the generic statistics + apply pair, which leaves its partials."""
            sv = []
            out = ops.groupnorm_silu(xin, norm.weight, norm.bias, eps=norm.eps, saved=sv, stats=(st, st1), **kw)
            return out, (sv[0] if sv else None)

        def res(b, x0, x1, st0, st1):
            gn1, conv1, gn2, conv2 = b.in_layers[0], b.in_layers[2], b.out_layers[0], b.out_layers[3]
            a1, s1 = gn_fwd(gn1, x0, st0, st1, in1=x1, silu=True)
            a1p, xs = a1, x0
            if b.up:
                xs = ops.upsample2x(x0)
            elif b.down:
                a1p, xs = ops.pool_act(a1, True, ops.ACT_NONE), ops.pool_act(x0, True, ops.ACT_NONE)
            off, eo = pk[id(b), "eoff"], b.emb_layers[1].out_features
            e = emb_all[:, off:off + eo]
            if b.use_scale_shift_norm:
                h, sh = conv_s(a1p, pk[id(b), "conv1"], bias=conv1.bias, upsample=b.up)
                a2, s2 = gn_fwd(gn2, h, sh, silu=True, scale_shift=e)
            else:
                h, sh = conv_s(a1p, pk[id(b), "conv1"], bias=conv1.bias, upsample=b.up, addvec=e)
                a2, s2 = gn_fwd(gn2, h, sh, silu=True)
            if (id(b), "skip") in pk:
                xs = ops.conv2d(x0, pk[id(b), "skip"], in1=x1, bias=b.skip_connection.bias)
            out, so = conv_s(a2, pk[id(b), "conv2"], bias=conv2.bias, residual=xs)
            tape.append(("res", b, x0, x1, a1p, h, a2, s1, s2))
            return out, (so if so is not None else stats_of(out))

        def attn(m, xa, sx):
            N, H, W, C = xa.shape
            hn, sn = gn_fwd(m.norm, xa, sx, silu=False)
            qkv = ops.conv2d(hn, pk[id(m), "qkv"], bias=m.qkv.bias)
            # the row log-sum-exp rides along (autograd's saved softmax statistics): the backward skips its own sweep for it
            a, lse = ops.attention(qkv.view(N, H * W, 3 * C), heads=m.num_heads, scale=1.0 / math.sqrt(C // m.num_heads), want_lse=True)
            lse = lse if SAVE_LSE else None
            out = ops.conv2d(a.view(N, H, W, C), pk[id(m), "proj"], bias=m.proj_out.bias, residual=xa)
            tape.append(("attn", m, xa, hn, qkv, a, sn, lse))
            return out, stats_of(out)

        def seq(mods, h, skip, sh, sskip):
            for m in mods:
                if isinstance(m, ResBlock):
                    h, sh = res(m, h, skip, sh, sskip)
                    skip = sskip = None
                elif isinstance(m, AttentionBlock):
                    h, sh = attn(m, h, sh)
                elif isinstance(m, Downsample):
                    tape.append(("down", m, h))
                    if m.use_conv:
                        h, sh = conv_s(h, pk[id(m), "conv"], bias=m.op.bias, stride=2, pad=1)
                    else:
                        h, sh = ops.pool_act(h, True, ops.ACT_NONE), None
                    sh = sh if sh is not None else stats_of(h)
                elif isinstance(m, Upsample):
                    tape.append(("up", m, h))
                    if m.use_conv:
                        h, sh = conv_s(h, pk[id(m), "conv"], bias=m.conv.bias, upsample=True)
                    else:
                        h, sh = ops.upsample2x(h), None
                    sh = sh if sh is not None else stats_of(h)
            return h, sh

        stats_of = net._stats_of if STREAM_GN else (lambda t: None)
        conv_in = net.input_blocks[0][0]
        h = ops.conv2d(x, pk["conv_in"], bias=conv_in.bias) if pk["conv_in"].k27 else \
            ops.conv2d(ops.nchw_f32_to_nhwc_bf16(x), pk["conv_in"], bias=conv_in.bias)
        sh = stats_of(h)
        hs = [(h, sh)]
        for i in range(1, len(net.input_blocks)):
            h, sh = seq(net.input_blocks[i], h, None, sh, None)
            tape.append(("push", None, len(hs)))
            hs.append((h, sh))
        h, sh = seq(net.middle_block, h, None, sh, None)
        for blk in net.output_blocks:
            tape.append(("skip", None, len(hs) - 1))
            skip, sskip = hs.pop()
            h, sh = seq(blk, h, skip, sh, sskip)
        gn = net.out[0]
        a_out, ctx.s_out = gn_fwd(gn, h, sh, silu=True)
        out = ops.conv2d(a_out, pk["conv_out"], bias=net.out[2].bias, out_nchw_f32=True)
        ctx.net, ctx.tape, ctx.h_last, ctx.a_out, ctx.x, ctx.sinus, ctx.y = net, tape, h, a_out, x, sinus, y
        ctx.emb_all_shape = emb_all.shape
        ctx.emb_all = emb_all
        return out

    @staticmethod
    def backward(ctx, d_out):
        from .unet import ResBlock
        net, tape = ctx.net, ctx.tape
        pk, pkt = net.packed(), _pack_t(net)
        grads = {}
        N = d_out.shape[0]
        dev = d_out.device
        d_emb_all = torch.zeros(ctx.emb_all_shape, dtype=torch.float32, device=dev)
        emb_all = ctx.emb_all

        def conv_wb(conv, x0, gy, k, x1=None, **kw):
            dw, grads[conv.bias] = ops.conv2d_wgrad(x0, gy, k, in1=x1, with_bias=True, **kw)
            grads[conv.weight] = dw.reshape(conv.weight.shape)

        def gn_bwd(norm, xin, dy, *, in1=None, add0=None, add1=None, silu=True, scale_shift=None, fwd_stats=None):
            dx0, dx1, dg, db, d_ss = ops.groupnorm_generic_bwd(xin, dy, norm.weight, norm.bias, in1=in1, add0=add0, add1=add1,
                                                               eps=norm.eps, silu=silu, scale_shift=scale_shift, fwd_stats=fwd_stats)
            grads[norm.weight], grads[norm.bias] = dg, db
            return dx0, dx1, d_ss

        # ---- head: out = conv(silu(gn(h_last)))
        H, W = d_out.shape[2], d_out.shape[3]
        d_pad = torch.zeros((N, H, W, 64), dtype=torch.bfloat16, device=dev)
        d_pad[..., : d_out.shape[1]] = d_out.permute(0, 2, 3, 1).to(torch.bfloat16)
        with ops.wgrad_branch((ctx.a_out, d_pad)):       # (captured step: a parallel branch, joined at the end of this backward)
            wg = ops._conv2d_wgrad(ctx.a_out, d_pad, 3)
            grads[net.out[2].weight] = wg[: net.out_channels].contiguous()
        grads[net.out[2].bias] = d_out.float().sum((0, 2, 3))
        d_a = ops.conv2d(d_pad, pkt["conv_out"])
        g, _, _ = gn_bwd(net.out[0], ctx.h_last, d_a, fwd_stats=ctx.s_out)

        gskip = {}

        def res_bwd(entry, g):
            _, b, x0, x1, a1p, h, a2, s1, s2 = entry
            gn1, conv1, gn2, conv2 = b.in_layers[0], b.in_layers[2], b.out_layers[0], b.out_layers[3]
            conv_wb(conv2, a2, g, 3)
            d_a2 = ops.conv2d(g, pkt[id(b), "conv2"])
            off, eo = pk[id(b), "eoff"], b.emb_layers[1].out_features
            if b.use_scale_shift_norm:
                d_h, _, d_ss = gn_bwd(gn2, h, d_a2, scale_shift=emb_all[:, off:off + eo], fwd_stats=s2)
                d_emb_all[:, off:off + eo] = d_ss
            else:
                d_h, _, _ = gn_bwd(gn2, h, d_a2, fwd_stats=s2)
                d_emb_all[:, off:off + eo] = ops.colsum_per_image(d_h)
            conv_wb(conv1, a1p, d_h, 3, upsample=b.up)
            d_a1 = ops.conv2d(d_h, pkt[id(b), "conv1"])
            if b.up:
                d_a1 = _up_sum(d_a1)
            elif b.down:
                d_a1 = _pool_t(d_a1)
            if (id(b), "skip") in pk:
                sk = b.skip_connection
                k = sk.weight.shape[-1]
                conv_wb(sk, x0, g, k, x1=x1)
                dxg0, dxg1, _ = gn_bwd(gn1, x0, d_a1, in1=x1, fwd_stats=s1)
                C0 = x0.shape[3]
                w = sk.weight
                if (id(b), "skip_t", C0) not in pkt:     # transposed fragments per concat source, cached with the other packs
                    pkt[id(b), "skip_t", C0] = (ops.pack_conv_weight(w[:, :C0].contiguous(), transpose_flip=True),
                                                ops.pack_conv_weight(w[:, C0:].contiguous(), transpose_flip=True) if x1 is not None else None)
                w0t, w1t = pkt[id(b), "skip_t", C0]
                d_x0 = ops.conv2d(g, w0t, residual=dxg0)
                d_x1 = None
                if x1 is not None:
                    d_x1 = ops.conv2d(g, w1t, residual=dxg1)
                return d_x0, d_x1
            assert x1 is None
            g_id = _up_sum(g) if b.up else (_pool_t(g) if b.down else g)
            d_x0, _, _ = gn_bwd(gn1, x0, d_a1, add0=g_id, fwd_stats=s1)
            return d_x0, None

        def attn_bwd(entry, g):
            _, m, xa, hn, qkv, a, sn, lse = entry
            Nn, Hh, Ww, C = xa.shape
            conv_wb(m.proj_out, a.view(Nn, Hh, Ww, C), g, 1)
            d_a = ops.conv2d(g, pkt[id(m), "proj"])
            d_qkv = ops.attention_bwd(qkv.view(Nn, Hh * Ww, 3 * C), d_a.view(Nn, Hh * Ww, C), m.num_heads,
                                      1.0 / math.sqrt(C // m.num_heads), o=a.view(Nn, Hh * Ww, C), lse=lse).view(Nn, Hh, Ww, 3 * C)
            conv_wb(m.qkv, hn, d_qkv, 1)
            d_hn = ops.conv2d(d_qkv, pkt[id(m), "qkv"])
            d_x, _, _ = gn_bwd(m.norm, xa, d_hn, add0=g, silu=False, fwd_stats=sn)
            return d_x

        i = len(tape) - 1
        while i >= 0:
            e = tape[i]
            kind = e[0]
            if kind == "res":
                d_x0, d_x1 = res_bwd(e, g)
                g = d_x0
                if e[3] is not None:     # first layer of an output block: (h, skip) concat
                    assert tape[i - 1][0] == "skip"
                    gskip[tape[i - 1][2]] = d_x1
                    i -= 1
            elif kind == "attn":
                g = attn_bwd(e, g)
            elif kind == "up":
                _, m, xin = e
                if m.use_conv:
                    conv_wb(m.conv, xin, g, 3, upsample=True)
                    g = _up_sum(ops.conv2d(g, pkt[id(m), "conv"]))
                else:
                    g = _up_sum(g)
            elif kind == "down":
                _, m, xin = e
                if m.use_conv:
                    conv_wb(m.op, xin, g, 3, stride=2, pad=1)
                    g = ops.conv2d(g, pkt[id(m), "conv"], pad=1, pad_br=1, upsample=2)
                else:
                    g = _pool_t(g)
            elif kind == "push":
                idx = e[2]
                if idx in gskip:
                    g = g + gskip.pop(idx)
            elif kind == "skip":
                raise AssertionError("skip entry must directly precede the ResBlock that consumes it")
            i -= 1
        if 0 in gskip:
            g = g + gskip.pop(0)
        conv_in = net.input_blocks[0][0]
        grads[conv_in.bias] = ops.colsum(g)
        grads[conv_in.weight] = ops.stem_conv_wgrad(ctx.x, g)
        dx = None
        if ctx.needs_input_grad[1]:
            if "conv_in_t" not in pkt:
                pkt["conv_in_t"] = ops.pack_conv_weight(conv_in.weight, transpose_flip=True)
            dx = ops.conv2d(g, pkt["conv_in_t"], out_nchw_f32=True)

        # ---- embedding graph (models/cm/unet.py:775-779, :249): time_embed MLP, label embedding, every emb_layers Linear as one
        # operator — dense backward on the HIP kernels (ops.linear_bwd), pre-activations recomputed
        blocks = [m for m in net.modules() if isinstance(m, ResBlock)]
        l0, l2 = net.time_embed[0], net.time_embed[2]
        e0 = ops.linear(ctx.sinus, pk["te0"], l0.bias)
        a0 = F.silu(e0)
        emb = ops.linear(a0, pk["te2"], l2.bias)
        if net.num_classes is not None:
            emb = emb + net.label_emb.weight.detach()[ctx.y]
        s_e = F.silu(emb)
        if "emb_t" not in pkt:
            with ops.pack_batch() as pb2:
                net._emb_w_cat_t = torch.cat([b.emb_layers[1].weight.detach() for b in blocks], 0).float().contiguous()
                pkt["emb_t"] = ops.pack_conv_weight(net._emb_w_cat_t, transpose_flip=True)
                pkt["te2_t"] = ops.pack_conv_weight(l2.weight, transpose_flip=True)
            net._pack_t_plan2 = pb2.plan
        ds, dw_cat, db_cat = ops.linear_bwd(s_e, d_emb_all, pkt["emb_t"])
        off = 0
        for b in blocks:
            eo = b.emb_layers[1].out_features
            grads[b.emb_layers[1].weight] = dw_cat[off:off + eo]
            grads[b.emb_layers[1].bias] = db_cat[off:off + eo]
            off += eo
        d_emb = ops.silu_bwd(emb, ds)
        if net.num_classes is not None:
            grads[net.label_emb.weight] = torch.zeros_like(net.label_emb.weight).index_add_(0, ctx.y, d_emb)
        da0, grads[l2.weight], grads[l2.bias] = ops.linear_bwd(a0, d_emb, pkt["te2_t"])
        de0 = ops.silu_bwd(e0, da0)
        _, grads[l0.weight], grads[l0.bias] = ops.linear_bwd(ctx.sinus, de0, None, need_dx=False)

        ops.wgrad_join()
        out = [None, dx, None, None]
        for prm in net.parameters():
            out.append(grads.get(prm))
        return tuple(out)


def forward_with_grad(net, x, timesteps, y=None):
    return _EDMUNetFn.apply(net, x, timesteps, y, *ops.fast_parameters(net))
