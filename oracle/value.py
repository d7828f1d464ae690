"""Oracle: IGEBMEncoderV2 value / energy network forward (TEST INFRASTRUCTURE).

Restates models/modules.py: ResBlockV2.forward :71-101, IGEBMEncoderV2.forward :142-163 for the
configuration every DxMI config uses (use_spectral_norm False, n_class None, keepdim False,
learn_out_scale True, out_activation linear; configs/cifar10/T10.yaml:20-31) and
models/value.py:8-12 (TimeIndependentValue ignores t).  State-dict keys carry the "net." prefix of
TimeIndependentValue.
"""
import torch
import torch.nn.functional as F

from .precision import Precision

# (in_mult, out_mult, downsample) of the six ResBlockV2, models/modules.py:117-126
BLOCKS = [(1, 1, True), (1, 1, False), (1, 2, True), (2, 2, False), (2, 2, True), (2, 2, False)]


def forward(sd, x, prec=None, prefix="net."):
    prec = prec or Precision("fp32")
    g = lambda k: sd[prefix + k]
    out = F.conv2d(prec.act(x), prec.w(g("conv1.weight")), g("conv1.bias"), padding=1)
    out = prec.act(F.leaky_relu(out, 0.2))
    for i, (_, _, down) in enumerate(BLOCKS):
        inp = out
        h = F.conv2d(inp, prec.w(g(f"blocks.{i}.conv1.weight")), g(f"blocks.{i}.conv1.bias"), padding=1)
        h = prec.act(F.leaky_relu(h, 0.2))
        h = F.conv2d(h, prec.w(g(f"blocks.{i}.conv2.weight")), g(f"blocks.{i}.conv2.bias"), padding=1)
        skip_key = prefix + f"blocks.{i}.skip.0.weight"
        skip = prec.act(F.conv2d(inp, prec.w(sd[skip_key]))) if skip_key in sd else inp
        h = h + skip
        if down:
            h = F.avg_pool2d(prec.act(h), 2)
        out = prec.act(F.leaky_relu(h, 0.2))
    out = F.relu(out)
    out = out.view(out.shape[0], out.shape[1], -1).sum(2)
    out = F.linear(out, g("linear.weight"), g("linear.bias"))
    if prefix + "out_scale.weight" in sd:
        out = F.linear(out, g("out_scale.weight"), g("out_scale.bias"))
    return out
