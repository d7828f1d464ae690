"""TEST INFRASTRUCTURE (oracle): torch-CPU fp32 restatement of the FID InceptionV3 forward — torchvision's published Inception3
architecture with the reference's FID patches (reference pytorch_fid/inception.py:129-163 forward, :193-310 patched blocks;
torchvision.models.inception: BasicConv2d = conv(bias=False) + BatchNorm2d(eps=1e-3) + ReLU, InceptionA / B / C / D / E).

PARITY UNPINNED: torchvision and the FID weight file (pt_inception-2015-12-05-6726825d.pth) are absent from this image, so nothing
the reference itself computes can be reproduced here; this restatement follows the published layer tables and is what the HIP
program (diffusion-by-maxentirl_amd/pytorch_fid/inception.py) is checked against, on formula weights.  Functional over a state
dict with torchvision's names.  Only tests/ may import this module."""
import torch
import torch.nn.functional as F


def _bc(sd, name, x, stride=1, padding=0):
    """BasicConv2d in eval mode."""
    x = F.conv2d(x, sd[name + ".conv.weight"], None, stride=stride, padding=padding)
    x = F.batch_norm(x, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"],
                     training=False, eps=1e-3)
    return F.relu(x)


def _avg(x):
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)        # the FID patch


def inception_a(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b5 = _bc(sd, p + ".branch5x5_2", _bc(sd, p + ".branch5x5_1", x), padding=2)
    b3 = _bc(sd, p + ".branch3x3dbl_3", _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), padding=1), padding=1)
    return torch.cat([b1, b5, b3, _bc(sd, p + ".branch_pool", _avg(x))], 1)


def inception_b(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3", x, stride=2)
    bd = _bc(sd, p + ".branch3x3dbl_3", _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), padding=1), stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def inception_c(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b7 = _bc(sd, p + ".branch7x7_3", _bc(sd, p + ".branch7x7_2", _bc(sd, p + ".branch7x7_1", x), padding=(0, 3)), padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_2", _bc(sd, p + ".branch7x7dbl_1", x), padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_4", _bc(sd, p + ".branch7x7dbl_3", bd, padding=(0, 3)), padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_5", bd, padding=(0, 3))
    return torch.cat([b1, b7, bd, _bc(sd, p + ".branch_pool", _avg(x))], 1)


def inception_d(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3_2", _bc(sd, p + ".branch3x3_1", x), stride=2)
    b7 = _bc(sd, p + ".branch7x7x3_3", _bc(sd, p + ".branch7x7x3_2", _bc(sd, p + ".branch7x7x3_1", x), padding=(0, 3)), padding=(3, 0))
    b7 = _bc(sd, p + ".branch7x7x3_4", b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def inception_e(sd, p, x, max_pool):
    b1 = _bc(sd, p + ".branch1x1", x)
    t = _bc(sd, p + ".branch3x3_1", x)
    b3 = torch.cat([_bc(sd, p + ".branch3x3_2a", t, padding=(0, 1)), _bc(sd, p + ".branch3x3_2b", t, padding=(1, 0))], 1)
    t = _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), padding=1)
    bd = torch.cat([_bc(sd, p + ".branch3x3dbl_3a", t, padding=(0, 1)), _bc(sd, p + ".branch3x3dbl_3b", t, padding=(1, 0))], 1)
    pooled = F.max_pool2d(x, kernel_size=3, stride=1, padding=1) if max_pool else _avg(x)      # Mixed_7c's max pool: inception.py:303-308
    return torch.cat([b1, b3, bd, _bc(sd, p + ".branch_pool", pooled)], 1)


def forward(sd, inp, resize_input=True, normalize_input=True, last_block=3):
    """-> list of the four blocks' outputs (reference InceptionV3.forward with output_blocks = 0..last_block)."""
    x = inp
    if resize_input:
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    if normalize_input:
        x = 2 * x - 1
    outs = []
    x = _bc(sd, "Conv2d_2b_3x3", _bc(sd, "Conv2d_2a_3x3", _bc(sd, "Conv2d_1a_3x3", x, stride=2)), padding=1)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    outs.append(x)
    if last_block >= 1:
        x = F.max_pool2d(_bc(sd, "Conv2d_4a_3x3", _bc(sd, "Conv2d_3b_1x1", x)), kernel_size=3, stride=2)
        outs.append(x)
    if last_block >= 2:
        for n in ("Mixed_5b", "Mixed_5c", "Mixed_5d"):
            x = inception_a(sd, n, x)
        x = inception_b(sd, "Mixed_6a", x)
        for n in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
            x = inception_c(sd, n, x)
        outs.append(x)
    if last_block >= 3:
        x = inception_d(sd, "Mixed_7a", x)
        x = inception_e(sd, "Mixed_7b", x, False)
        x = inception_e(sd, "Mixed_7c", x, True)
        outs.append(F.adaptive_avg_pool2d(x, (1, 1)))
    return outs
