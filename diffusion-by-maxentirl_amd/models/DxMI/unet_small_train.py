"""Autograd path of the DDPM U-Net on the gfx950 kernels (policy step of DxMI:
models/DxMI/trainer.py:361-389 back-propagates the sampler loss through `sample_step` into every
U-Net parameter; reference graph = torch autograd over models/DxMI/unet_small.py:292-332).

One torch.autograd.Function wraps the network.  forward() is the inference program plus saved bf16
NHWC activations (and one dropout seed per ResnetBlock in train mode); backward() walks it in reverse:
  conv data gradients   = the forward MFMA kernels on transpose-flipped weight fragments (stride-2:
                          over the zero-stuffed gradient; upsample: full-res gradient then 2x2 sum),
                          with the skip-connection gradient fused as the epilogue residual;
  conv weight gradients = MFMA pixel-GEMM (dxmi_conv2d_wgrad, also strided / upsampled / 1x1);
  GroupNorm(+SiLU)      = dxmi_groupnorm_silu_bwd (dx split back into the two concat sources);
  attention             = five batched MFMA GEMMs + softmax backward (dxmi_bgemm_bf16);
  dropout               = dxmi_dropout_bf16 with the forward's seed (no stored mask);
  temb MLP + temb_proj  = tiny dense layers: re-evaluated and differentiated with torch fp32 matmuls
                          (<0.05 % of the FLOPs; plain library GEMMs).
Parameter gradients are returned in `net.parameters()` order as fp32 tensors.
"""
import torch
import torch.nn.functional as F

from dxmi_hip import graph as _graph
from dxmi_hip import ops


def _pack_t(net):
    """Transpose-flipped fragments of every conv (cached on the net, refreshed with the parameters)."""
    key = net._param_key()
    if getattr(net, "_packed_t", None) is not None and net._packed_t_key == key:
        return net._packed_t
    from .unet_small import AttnBlock, Downsample, Upsample
    pk = {}
    # parameters updated in place: the previous set's buffers are rewritten (addresses held by a captured hipGraph stay valid)
    reuse = net._pack_t_bufs if (getattr(net, "_packed_t", None) is not None and net._same_storage(key, net._packed_t_key)) else None
    with ops.pack_batch(reuse=reuse) as pb:            # a few multi-tensor launches instead of one per layer
        for b in net._resblocks():
            pk[id(b), "conv1"] = ops.pack_conv_weight(b.conv1.weight, transpose_flip=True)
            pk[id(b), "conv2"] = ops.pack_conv_weight(b.conv2.weight, transpose_flip=True)
            if b.in_channels != b.out_channels:
                sc = b.conv_shortcut if b.use_conv_shortcut else b.nin_shortcut
                pk[id(b), "short"] = sc.weight  # split per concat source at use time
        for m in net.modules():
            if isinstance(m, AttnBlock):
                pk[id(m), "qkv"] = ops.pack_conv_weight(torch.cat([m.q.weight, m.k.weight, m.v.weight], 0), transpose_flip=True)
                pk[id(m), "proj"] = ops.pack_conv_weight(m.proj_out.weight, transpose_flip=True)
            elif isinstance(m, (Upsample, Downsample)):
                pk[id(m), "conv"] = ops.pack_conv_weight(m.conv.weight, transpose_flip=True)
        w = net.conv_out.weight
        wpad = torch.zeros((64,) + tuple(w.shape[1:]), dtype=torch.float32, device=w.device)
        wpad[: w.shape[0]] = w.detach()
        pk["conv_out"] = ops.pack_conv_weight(wpad, transpose_flip=True)
    net._pack_t_bufs = pb.buffers
    net._packed_t_planned = set(pk.keys())       # everything else is packed on demand by a backward and dropped at the next re-pack
    net._packed_t, net._packed_t_key = pk, key
    return pk


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, t, *params):
        pk = net.packed()
        x = x.contiguous().float()
        training, p_drop = net.training, net.dropout_p
        net.dropout_seeds_used = []
        emb = ops.timestep_embedding(t, net.ch, order=0)
        h1 = ops.linear(emb, pk["dense0"], net.temb.dense[0].bias, post_act=ops.ACT_SILU)
        s_temb = ops.linear(h1, pk["dense1"], net.temb.dense[1].bias, post_act=ops.ACT_SILU)
        tp = ops.linear(s_temb, pk["tproj"], pk["tproj_bias"])
        tape = []  # (kind, module, saved...)

        def res(b, x0, x1):
            a1 = ops.groupnorm_silu(x0, b.norm1.weight, b.norm1.bias, in1=x1, eps=1e-6, silu=True)
            off = pk[id(b), "toff"]
            h = ops.conv2d(a1, pk[id(b), "conv1"], bias=b.conv1.bias, addvec=tp[:, off:off + b.out_channels])
            a2 = ops.groupnorm_silu(h, b.norm2.weight, b.norm2.bias, eps=1e-6, silu=True)
            mask = None
            if training and p_drop > 0:
                # nn.Dropout (unet_small.py:129) as a counter-hash kernel: the backward regenerates the mask from the seed
                cap = _graph.current()
                if cap is None:
                    mask = net._next_dropout_seed()
                else:       # captured step: the seed is a host input of the graph, drawn per replay from the same counter hash
                    mask = cap.host_input(torch.int32, 1, lambda: [net._next_dropout_seed()])
                a2 = ops.dropout(a2, p_drop, mask)
            if b.in_channels != b.out_channels:
                sc_mod = b.conv_shortcut if b.use_conv_shortcut else b.nin_shortcut
                sc = ops.conv2d(x0, pk[id(b), "short"], in1=x1, bias=sc_mod.bias)
            else:
                sc = x0
            out = ops.conv2d(a2, pk[id(b), "conv2"], bias=b.conv2.bias, residual=sc)
            tape.append(("res", b, x0, x1, a1, h, a2, mask))
            return out

        def attn(m, xa):
            N, H, W, C = xa.shape
            hn = ops.groupnorm_silu(xa, m.norm.weight, m.norm.bias, eps=1e-6, silu=False)
            qkv = ops.conv2d(hn, pk[id(m), "qkv"], bias=pk[id(m), "qkv_bias"])
            a = ops.attention(qkv.view(N, H * W, 3 * C), heads=1, scale=float(int(C) ** (-0.5)))
            out = ops.conv2d(a.view(N, H, W, C), pk[id(m), "proj"], bias=m.proj_out.bias, residual=xa)
            tape.append(("attn", m, xa, hn, qkv, a))
            return out

        h = ops.conv2d(x, pk["conv_in"], bias=net.conv_in.bias) if pk["conv_in"].k27 else \
            ops.conv2d(ops.nchw_f32_to_nhwc_bf16(x), pk["conv_in"], bias=net.conv_in.bias)
        hs = [h]
        for i_level, lvl in enumerate(net.down):
            for i_block, blk in enumerate(lvl.block):
                h = res(blk, hs[-1], None)
                if len(lvl.attn) > 0:
                    h = attn(lvl.attn[i_block], h)
                tape.append(("push", None, len(hs)))
                hs.append(h)
            if i_level != net.num_resolutions - 1:
                ds = lvl.downsample
                hd = ops.conv2d(hs[-1], pk[id(ds), "conv"], bias=ds.conv.bias, stride=2, pad=0, pad_br=1)
                tape.append(("down", ds, hs[-1]))
                tape.append(("push", None, len(hs)))
                hs.append(hd)
        h = hs[-1]
        tape.append(("mid_start", None, len(hs) - 1))
        h = res(net.mid.block_1, h, None)
        h = attn(net.mid.attn_1, h)
        h = res(net.mid.block_2, h, None)
        for i_level in reversed(range(net.num_resolutions)):
            lvl = net.up[i_level]
            for i_block, blk in enumerate(lvl.block):
                skip_idx = len(hs) - 1
                skip = hs.pop()
                tape.append(("skip", None, skip_idx))
                h = res(blk, h, skip)
                if len(lvl.attn) > 0:
                    h = attn(lvl.attn[i_block], h)
            if i_level != 0:
                us = lvl.upsample
                tape.append(("up", us, h))
                h = ops.conv2d(h, pk[id(us), "conv"], bias=us.conv.bias, upsample=True)
        a_out = ops.groupnorm_silu(h, net.norm_out.weight, net.norm_out.bias, eps=1e-6, silu=True)
        eps = ops.conv2d(a_out, pk["conv_out"], bias=net.conv_out.bias, out_nchw_f32=True)
        ctx.net, ctx.tape, ctx.h_last, ctx.a_out, ctx.x, ctx.emb = net, tape, h, a_out, x, emb
        return eps

    @staticmethod
    def backward(ctx, d_eps):
        net, tape = ctx.net, ctx.tape
        pk, pkt = net.packed(), _pack_t(net)
        grads = {}
        N = d_eps.shape[0]
        dev = d_eps.device
        d_tp = torch.zeros((N, pk["tproj_bias"].numel()), dtype=torch.float32, device=dev)

        def conv_wb(conv, x0, gy, k, x1=None, **kw):
            grads[conv.weight], grads[conv.bias] = ops.conv2d_wgrad(x0, gy, k, in1=x1, with_bias=True, **kw)

        # ---- head: eps = conv_out(silu(gn(h_last)))
        H = d_eps.shape[2]
        d_pad = torch.zeros((N, H, H, 64), dtype=torch.bfloat16, device=dev)
        d_pad[..., : d_eps.shape[1]] = d_eps.permute(0, 2, 3, 1).to(torch.bfloat16)
        with ops.wgrad_branch((ctx.a_out, d_pad)):       # (captured step: a parallel branch, joined at the end of this backward)
            wg = ops._conv2d_wgrad(ctx.a_out, d_pad, 3)
            grads[net.conv_out.weight] = wg[: net.out_ch].contiguous()
        grads[net.conv_out.bias] = d_eps.float().sum((0, 2, 3))
        d_a = ops.conv2d(d_pad, pkt["conv_out"])
        g, _, dg, db = ops.groupnorm_silu_bwd(ctx.h_last, d_a, net.norm_out.weight, net.norm_out.bias, silu=True)
        grads[net.norm_out.weight], grads[net.norm_out.bias] = dg, db

        gskip = {}

        def res_bwd(entry, g, add0=None, add1=None):
            _, b, x0, x1, a1, h, a2, mask = entry
            conv_wb(b.conv2, a2, g, 3)
            d_a2 = ops.conv2d(g, pkt[id(b), "conv2"])
            if mask is not None:
                d_a2 = ops.dropout(d_a2, net.dropout_p, mask, out=d_a2)
            d_h, _, dg2, db2 = ops.groupnorm_silu_bwd(h, d_a2, b.norm2.weight, b.norm2.bias, silu=True)
            grads[b.norm2.weight], grads[b.norm2.bias] = dg2, db2
            off = pk[id(b), "toff"]
            d_tp[:, off:off + b.out_channels] = ops.colsum_per_image(d_h)
            conv_wb(b.conv1, a1, d_h, 3)
            d_a1 = ops.conv2d(d_h, pkt[id(b), "conv1"])
            if b.in_channels != b.out_channels:
                sc_mod = b.conv_shortcut if b.use_conv_shortcut else b.nin_shortcut
                k = sc_mod.weight.shape[-1]
                conv_wb(sc_mod, x0, g, k, x1=x1)
                dxg0, dxg1, dg1, db1 = ops.groupnorm_silu_bwd(x0, d_a1, b.norm1.weight, b.norm1.bias, in1=x1, add0=add0,
                                                              add1=add1, silu=True)
                C0 = x0.shape[3]
                w = pkt[id(b), "short"]
                # transposed fragments of the shortcut, split per concat source; cached with the other packs
                if (id(b), "short_t", C0) not in pkt:
                    pkt[id(b), "short_t", C0] = (ops.pack_conv_weight(w[:, :C0].contiguous(), transpose_flip=True),
                                                 ops.pack_conv_weight(w[:, C0:].contiguous(), transpose_flip=True) if x1 is not None else None)
                w0t, w1t = pkt[id(b), "short_t", C0]
                d_x0 = ops.conv2d(g, w0t, residual=dxg0)
                d_x1 = None
                if x1 is not None:
                    d_x1 = ops.conv2d(g, w1t, residual=dxg1)
            else:
                assert x1 is None
                addin = g if add0 is None else g + add0
                d_x0, d_x1, dg1, db1 = ops.groupnorm_silu_bwd(x0, d_a1, b.norm1.weight, b.norm1.bias, add0=addin, silu=True)
            grads[b.norm1.weight], grads[b.norm1.bias] = dg1, db1
            return d_x0, d_x1

        def attn_bwd(entry, g):
            _, m, xa, hn, qkv, a = entry
            Nn, Hh, Ww, C = xa.shape
            conv_wb(m.proj_out, a.view(Nn, Hh, Ww, C), g, 1)
            d_a = ops.conv2d(g, pkt[id(m), "proj"])
            d_qkv = ops.attention_bwd(qkv.view(Nn, Hh * Ww, 3 * C), d_a.view(Nn, Hh * Ww, C), 1, float(int(C) ** (-0.5)))
            d_qkv = d_qkv.view(Nn, Hh, Ww, 3 * C)
            with ops.wgrad_branch((hn, d_qkv)):
                wq, bq = ops._conv2d_wgrad(hn, d_qkv, 1, with_bias=True)   # the bias gradient from the dY tiles the kernel stages anyway
                for j, conv in enumerate((m.q, m.k, m.v)):
                    grads[conv.weight] = wq[j * C:(j + 1) * C].contiguous()
                    grads[conv.bias] = bq[j * C:(j + 1) * C].contiguous()
            d_hn = ops.conv2d(d_qkv, pkt[id(m), "qkv"])
            d_x, _, dgn, dbn = ops.groupnorm_silu_bwd(xa, d_hn, m.norm.weight, m.norm.bias, add0=g, silu=False)
            grads[m.norm.weight], grads[m.norm.bias] = dgn, dbn
            return d_x

        # ---- reverse walk
        i = len(tape) - 1
        while i >= 0:
            e = tape[i]
            kind = e[0]
            if kind == "up":
                _, us, xin = e
                conv_wb(us.conv, xin, g, 3, upsample=True)
                d_hi = ops.conv2d(g, pkt[id(us), "conv"])
                g = ops.pool_act(d_hi, True, ops.ACT_NONE)
                g.mul_(4.0)  # mean -> sum over the 4 replicated pixels (exact in bf16)
            elif kind == "attn":
                g = attn_bwd(e, g)
            elif kind == "res":
                x1 = e[3]
                if x1 is not None:   # up-path block: (h, skip) concat; the preceding "skip" entry names the hs index
                    d_x0, d_x1 = res_bwd(e, g)
                    assert tape[i - 1][0] == "skip"
                    gskip[tape[i - 1][2]] = d_x1
                    g = d_x0
                    i -= 1
                else:
                    g, _ = res_bwd(e, g)  # down / mid block
            elif kind == "down":
                _, ds, xin = e
                conv_wb(ds.conv, xin, g, 3, stride=2, pad=0)
                g = ops.conv2d(g, pkt[id(ds), "conv"], pad=2, pad_br=0, upsample=2)
            elif kind == "push":
                # h was stored as hs[idx]: the up path may have consumed it as a skip
                idx = e[2]
                if idx in gskip:
                    g = g + gskip.pop(idx)
            elif kind == "mid_start":
                pass
            i -= 1
        # hs[0] = conv_in output: add its skip gradient, then the stem conv
        if 0 in gskip:
            g = g + gskip.pop(0)
        grads[net.conv_in.bias] = ops.colsum(g)
        grads[net.conv_in.weight] = ops.stem_conv_wgrad(ctx.x, g)
        dx = None
        if ctx.needs_input_grad[1]:
            if "conv_in_t" not in pkt:
                pkt["conv_in_t"] = ops.pack_conv_weight(net.conv_in.weight, transpose_flip=True)
            dx = ops.conv2d(g, pkt["conv_in_t"], out_nchw_f32=True)

        # ---- temb MLP + all temb_proj layers (reference unet_small.py:296-299, :123): dense backward on the HIP kernels
        # (ops.linear_bwd: transposed-pack linear for dx, the 1x1 weight-gradient kernel for dW), pre-activations recomputed
        blocks = list(net._resblocks())
        d0, d1 = net.temb.dense
        e0 = ops.linear(ctx.emb, pk["dense0"], d0.bias)                        # pre-activation of dense[0]
        a0 = F.silu(e0)
        e1 = ops.linear(a0, pk["dense1"], d1.bias)
        s_t = F.silu(e1)
        if "tproj_t" not in pkt:
            pkt["tproj_t"] = ops.pack_conv_weight(torch.cat([b.temb_proj.weight for b in blocks], 0), transpose_flip=True)
            pkt["dense1_t"] = ops.pack_conv_weight(d1.weight, transpose_flip=True)
        ds, dw_cat, db_cat = ops.linear_bwd(s_t, d_tp, pkt["tproj_t"])
        off = 0
        for b in blocks:
            grads[b.temb_proj.weight] = dw_cat[off:off + b.out_channels]
            grads[b.temb_proj.bias] = db_cat[off:off + b.out_channels]
            off += b.out_channels
        de1 = ops.silu_bwd(e1, ds)
        da0, grads[d1.weight], grads[d1.bias] = ops.linear_bwd(a0, de1, pkt["dense1_t"])
        de0 = ops.silu_bwd(e0, da0)
        _, grads[d0.weight], grads[d0.bias] = ops.linear_bwd(ctx.emb, de0, None, need_dx=False)

        ops.wgrad_join()
        out = [None, dx, None]
        for prm in net.parameters():
            out.append(grads.get(prm))
        return tuple(out)


def forward_with_grad(net, x, t):
    return _UNetFn.apply(net, x, t, *ops.fast_parameters(net))
