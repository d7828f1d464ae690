"""Autograd path of IGEBMEncoderV2 on the gfx950 kernels (training: models/DxMI/trainer.py:244-326
back-propagates into the value parameters, :369-387 into the input `next_state`).

One torch.autograd.Function wraps the whole network: forward runs the same fused HIP program as
inference and keeps the (bf16 NHWC) activations; backward walks the blocks in reverse with
  * data gradients  = the forward MFMA conv kernel on transpose-flipped weight fragments, with the
    LeakyReLU derivative (mask of the saved activation) and the skip-path gradient fused into its
    epilogue,
  * weight gradients = the MFMA pixel-GEMM (dxmi_conv2d_wgrad), bias gradients = column sums of dY from the same launch,
  * avg-pool / LeakyReLU / head backward = small fused elementwise kernels.
Parameter gradients come back as fp32 tensors in the reference's parameter order, so optimizers,
clip_grad_norm_ and DDP-style flat-gradient all-reduce see ordinary `.grad`s.
"""
import torch

from dxmi_hip import ops

SLOPE = 0.2


class _ValueNetFn(torch.autograd.Function):
    """nfree: the first `nfree` images of the batch take part in the forward only (round 5: the TD target v(next_state) and the
    prediction v(state) of one TD step are ONE forward over [next_state | state] — same parameters, no mode-dependent layer in this
    encoder — and only the second half is back-propagated; every image's value is what a forward on its own would give)."""

    @staticmethod
    def forward(ctx, net, nfree, x, *params):
        pk = net.packed()
        x = x.contiguous().float()
        k27 = pk["conv1"].k27
        x_nhwc = None if k27 else ops.nchw_f32_to_nhwc_bf16(x)
        a0 = ops.conv2d(x if k27 else x_nhwc, pk["conv1"], bias=net.conv1.bias, act=ops.ACT_LEAKY02)
        saved = []  # per block: (inp, h1, out, c2_or_None)
        inp = a0
        for i, b in enumerate(net.blocks):
            h1 = ops.conv2d(inp, pk[i, "conv1"], bias=b.conv1.bias, act=ops.ACT_LEAKY02)
            skip = ops.conv2d(inp, pk[i, "skip"]) if b.skip is not None else inp
            if b.downsample:
                c2 = ops.conv2d(h1, pk[i, "conv2"], bias=b.conv2.bias, residual=skip)
                out = ops.pool_act(c2, True, ops.ACT_LEAKY02)
            else:
                out = ops.conv2d(h1, pk[i, "conv2"], bias=b.conv2.bias, residual=skip, act=ops.ACT_LEAKY02)
            saved.append((inp[nfree:], h1[nfree:], out[nfree:]) if nfree else (inp, h1, out))
            inp = out
        ow, ob = (net.out_scale.weight, net.out_scale.bias) if net.learn_out_scale else (None, None)
        res = ops.value_head(inp, net.linear.weight, net.linear.bias, ow, ob)
        ctx.net, ctx.saved, ctx.a0, ctx.x, ctx.x_nhwc = net, saved, (a0[nfree:] if nfree else a0), (x[nfree:] if nfree else x), x_nhwc
        ctx.nfree = nfree
        ctx.need_dx = x.requires_grad or ctx.needs_input_grad[2]
        assert not (nfree and ctx.need_dx), "a forward-only prefix and an input gradient do not go together"
        # parameter gradients are skipped when no parameter asks for one (the policy step only needs d/dx: its value
        # gradients are discarded by the next zero_grad, reference trainer.py:235, :387)
        ctx.need_dw = any(prm.requires_grad for prm in params)
        return res

    @staticmethod
    def backward(ctx, dres):
        net, saved, a0 = ctx.net, ctx.saved, ctx.a0
        pk_t = net.packed_transposed()
        grads = {}
        feat = saved[-1][2]
        N = feat.shape[0]
        dy = dres.reshape(-1)[ctx.nfree:].float().contiguous()
        # ---- head: y = s @ w + b ; out = y * ow + ob
        w = net.linear.weight.reshape(-1)
        if net.learn_out_scale:
            ow = net.out_scale.weight.reshape(())
            dy_pre = dy * ow
        else:
            dy_pre = dy
        dfeat, s = ops.value_head_bwd(feat, w.contiguous(), dy_pre.contiguous())
        dw = ctx.need_dw
        if dw:
            # the four head-parameter gradients from one launch (round 6: they were ~10 torch elementwise / reduce launches per backward)
            C = w.numel()
            hp = ops.value_head_pgrad(s, w.contiguous(), net.linear.bias, dy, net.out_scale.weight if net.learn_out_scale else None)
            grads[net.linear.weight] = hp[:C].view(1, C)
            grads[net.linear.bias] = hp[C:C + 1]
            if net.learn_out_scale:
                grads[net.out_scale.weight] = hp[C + 1:C + 2].view(1, 1)
                grads[net.out_scale.bias] = hp[C + 2:C + 3]
        # dfeat is the gradient w.r.t. the last block's OUTPUT (post LeakyReLU)
        g_out = dfeat
        for i in reversed(range(len(net.blocks))):
            b = net.blocks[i]
            inp, h1, out = saved[i]
            # gradient w.r.t. conv2 output + skip (before pool / LeakyReLU)
            d_c2 = ops.pool_act_bwd(g_out, out, b.downsample, SLOPE)
            if dw:
                grads[b.conv2.weight], grads[b.conv2.bias] = ops.conv2d_wgrad(h1, d_c2, 3, with_bias=True)
            d_h1 = ops.conv2d(d_c2, pk_t[i, "conv2"], mask_src=h1, mask_slope=SLOPE)  # * LeakyReLU'(h1)
            if dw:
                grads[b.conv1.weight], grads[b.conv1.bias] = ops.conv2d_wgrad(inp, d_h1, 3, with_bias=True)
            if b.skip is not None:
                if dw:
                    grads[b.skip[0].weight] = ops.conv2d_wgrad(inp, d_c2, 1)
                d_skip = ops.conv2d(d_c2, pk_t[i, "skip"])
            else:
                d_skip = d_c2
            # gradient w.r.t. the block input = previous block's output (post LeakyReLU)
            g_out = ops.conv2d(d_h1, pk_t[i, "conv1"], residual=d_skip)
        # ---- stem conv: a0 = LeakyReLU(conv1(x))
        d_a0 = ops.pool_act_bwd(g_out, a0, False, SLOPE)
        assert net.in_chan == 3, "stem backward is written for 3-channel images"
        if dw:
            grads[net.conv1.bias] = ops.colsum(d_a0)
            grads[net.conv1.weight] = ops.stem_conv_wgrad(ctx.x, d_a0)
        dx = None
        if ctx.need_dx:
            # data gradient of the stem: 128 -> 3 channel conv on flipped weights, fp32 NCHW out
            dx = ops.conv2d(d_a0, ops.pack_conv_weight(net.conv1.weight, transpose_flip=True), out_nchw_f32=True)
        ops.wgrad_join()
        out = [None, None, dx]
        for prm in net.parameters():
            out.append(grads.get(prm))
        return tuple(out)


def forward_with_grad(net, x, nfree=0):
    return _ValueNetFn.apply(net, nfree, x, *ops.fast_parameters(net))
