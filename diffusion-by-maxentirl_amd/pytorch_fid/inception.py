"""`pytorch_fid.inception.InceptionV3` on the gfx950 kernels — the feature extractor of the FID evaluation (SURVEY 8 f4;
reference pytorch_fid/inception.py:16-163: blocks, resize / normalise, `forward -> list of feature maps`; the FID patches of
:193-310 — average pools that do not count padding in Mixed_5b-5d / 6b-6e / 7b, a MAX pool in Mixed_7c — over torchvision's
Inception3, whose published architecture is restated here because torchvision is not in the target image).

The torch.nn layers are parameter CONTAINERS with torchvision's names (`Mixed_5b.branch1x1.conv.weight`, `...bn.running_mean`),
so the FID weight file (pt_inception-2015-12-05-6726825d.pth: inception.py:13) loads with `load_fid_weights()`; the module's own
state dict has the reference class's keys (`blocks.0.0.conv.weight` ...).  forward() runs the HIP program: bilinear resize + 2x-1
into NHWC bf16, every BasicConv2d as one generic implicit-GEMM MFMA launch with its BatchNorm folded into the weights and ReLU in
the epilogue, the blocks' concatenations written in place (channel windows of one tensor), the FID pools, the global average pool.
PARITY IS UNPINNED: neither torchvision nor the weight file exists in this image, so the tests check the program against a
torch-CPU restatement of the same architecture on formula weights (oracle/inception.py), not against the reference's outputs.
No CPU path: non-device tensors raise.
"""
import os

import torch
import torch.nn as nn

from dxmi_hip import ops
from dxmi_hip._lib import DxmiError


class BasicConv2d(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d(eps=0.001) + ReLU container (torchvision.models.inception.BasicConv2d)."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        pair = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        self.k, self.s, self.p = pair(kernel_size), pair(stride), pair(padding)


def _named(**mods):
    m = nn.Module()
    for k, v in mods.items():
        setattr(m, k, v)
    return m


def InceptionA(cin, pool_features):
    return _named(kind="A", branch1x1=BasicConv2d(cin, 64, 1), branch5x5_1=BasicConv2d(cin, 48, 1), branch5x5_2=BasicConv2d(48, 64, 5, padding=2),
                  branch3x3dbl_1=BasicConv2d(cin, 64, 1), branch3x3dbl_2=BasicConv2d(64, 96, 3, padding=1),
                  branch3x3dbl_3=BasicConv2d(96, 96, 3, padding=1), branch_pool=BasicConv2d(cin, pool_features, 1))


def InceptionB(cin):
    return _named(kind="B", branch3x3=BasicConv2d(cin, 384, 3, stride=2), branch3x3dbl_1=BasicConv2d(cin, 64, 1),
                  branch3x3dbl_2=BasicConv2d(64, 96, 3, padding=1), branch3x3dbl_3=BasicConv2d(96, 96, 3, stride=2))


def InceptionC(cin, c7):
    return _named(kind="C", branch1x1=BasicConv2d(cin, 192, 1), branch7x7_1=BasicConv2d(cin, c7, 1),
                  branch7x7_2=BasicConv2d(c7, c7, (1, 7), padding=(0, 3)), branch7x7_3=BasicConv2d(c7, 192, (7, 1), padding=(3, 0)),
                  branch7x7dbl_1=BasicConv2d(cin, c7, 1), branch7x7dbl_2=BasicConv2d(c7, c7, (7, 1), padding=(3, 0)),
                  branch7x7dbl_3=BasicConv2d(c7, c7, (1, 7), padding=(0, 3)), branch7x7dbl_4=BasicConv2d(c7, c7, (7, 1), padding=(3, 0)),
                  branch7x7dbl_5=BasicConv2d(c7, 192, (1, 7), padding=(0, 3)), branch_pool=BasicConv2d(cin, 192, 1))


def InceptionD(cin):
    return _named(kind="D", branch3x3_1=BasicConv2d(cin, 192, 1), branch3x3_2=BasicConv2d(192, 320, 3, stride=2),
                  branch7x7x3_1=BasicConv2d(cin, 192, 1), branch7x7x3_2=BasicConv2d(192, 192, (1, 7), padding=(0, 3)),
                  branch7x7x3_3=BasicConv2d(192, 192, (7, 1), padding=(3, 0)), branch7x7x3_4=BasicConv2d(192, 192, 3, stride=2))


def InceptionE(cin, max_pool):
    return _named(kind="E2" if max_pool else "E1", branch1x1=BasicConv2d(cin, 320, 1), branch3x3_1=BasicConv2d(cin, 384, 1),
                  branch3x3_2a=BasicConv2d(384, 384, (1, 3), padding=(0, 1)), branch3x3_2b=BasicConv2d(384, 384, (3, 1), padding=(1, 0)),
                  branch3x3dbl_1=BasicConv2d(cin, 448, 1), branch3x3dbl_2=BasicConv2d(448, 384, 3, padding=1),
                  branch3x3dbl_3a=BasicConv2d(384, 384, (1, 3), padding=(0, 1)), branch3x3dbl_3b=BasicConv2d(384, 384, (3, 1), padding=(1, 0)),
                  branch_pool=BasicConv2d(cin, 192, 1))


# torchvision's attribute names of the layers, in block order (reference inception.py:83-124)
_LAYOUT = [["Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3"], ["Conv2d_3b_1x1", "Conv2d_4a_3x3"],
           ["Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"], ["Mixed_7a", "Mixed_7b", "Mixed_7c"]]


class InceptionV3(nn.Module):
    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks=(DEFAULT_BLOCK_INDEX,), resize_input=True, normalize_input=True, requires_grad=False,
                 use_fid_inception=True, weights=None):
        """Constructor of the reference class (inception.py:31-36) + `weights`: path of the FID weight file (or a state dict with
        torchvision's names); None leaves the torch default initialisation — the reference downloads the file, which this image
        cannot."""
        super().__init__()
        if not use_fid_inception:
            raise NotImplementedError("use_fid_inception=False (torchvision's own weights and pools) is not built: the DxMI scripts score FID")
        if requires_grad:
            raise NotImplementedError("InceptionV3 on the HIP path is an inference-only extractor")
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, "Last possible output block index is 3"
        layers = {"Conv2d_1a_3x3": BasicConv2d(3, 32, 3, stride=2), "Conv2d_2a_3x3": BasicConv2d(32, 32, 3),
                  "Conv2d_2b_3x3": BasicConv2d(32, 64, 3, padding=1), "Conv2d_3b_1x1": BasicConv2d(64, 80, 1),
                  "Conv2d_4a_3x3": BasicConv2d(80, 192, 3), "Mixed_5b": InceptionA(192, 32), "Mixed_5c": InceptionA(256, 64),
                  "Mixed_5d": InceptionA(288, 64), "Mixed_6a": InceptionB(288), "Mixed_6b": InceptionC(768, 128),
                  "Mixed_6c": InceptionC(768, 160), "Mixed_6d": InceptionC(768, 160), "Mixed_6e": InceptionC(768, 192),
                  "Mixed_7a": InceptionD(768), "Mixed_7b": InceptionE(1280, False), "Mixed_7c": InceptionE(2048, True)}
        self.blocks = nn.ModuleList()
        self._by_name = {}
        tails = [nn.MaxPool2d(kernel_size=3, stride=2), nn.MaxPool2d(kernel_size=3, stride=2), None, nn.AdaptiveAvgPool2d(output_size=(1, 1))]
        for b in range(self.last_needed_block + 1):
            mods = [layers[n] for n in _LAYOUT[b]]
            for n in _LAYOUT[b]:
                self._by_name[n] = layers[n]
            self.blocks.append(nn.Sequential(*(mods + ([tails[b]] if tails[b] is not None else []))))
        for prm in self.parameters():
            prm.requires_grad = False
        self._packed, self._packed_key = None, None
        if weights is not None:
            self.load_fid_weights(torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else weights)

    def load_fid_weights(self, state_dict):
        """Load a torchvision-named Inception3 state dict (`Conv2d_1a_3x3.conv.weight`, `Mixed_5b.branch1x1.bn.running_var` ...: the
        FID weight file); the classifier (`fc.*`) and layers beyond `last_needed_block` are ignored, a missing needed tensor raises."""
        for name, mod in self._by_name.items():
            sub = {k[len(name) + 1:]: v for k, v in state_dict.items() if k.startswith(name + ".")}
            mod.load_state_dict(sub, strict=True)
        self._packed = None

    # ------------------------------------------------------------------ packed weights
    def _convs(self):
        for name, mod in self._by_name.items():
            if isinstance(mod, BasicConv2d):
                yield name, mod
            else:
                for cn, c in mod.named_children():
                    if isinstance(c, BasicConv2d):
                        yield f"{name}.{cn}", c

    def packed(self):
        key = tuple((t.data_ptr(), t._version) for _, c in self._convs() for t in (c.conv.weight, c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var))
        if self._packed is None or key != self._packed_key:
            self._packed = {id(c): ops.gconv_pack(c.conv.weight, (c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var), c.bn.eps)
                            for _, c in self._convs()}
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------ program
    @staticmethod
    def _run(pk, c, x, out=None, coff=0):
        return ops.gconv(x, pk[id(c)], stride=c.s, pad=c.p, relu=True, out=out, coff=coff)

    def _mixed(self, pk, m, x):
        run = lambda c, t, out=None, coff=0: self._run(pk, c, t, out, coff)
        N, H, W, _ = x.shape
        new = lambda h, w, c: torch.empty((N, h, w, c), dtype=torch.bfloat16, device=x.device)
        k = m.kind
        if k == "A":
            pf = m.branch_pool.conv.out_channels
            out = new(H, W, 224 + pf)
            run(m.branch1x1, x, out, 0)
            run(m.branch5x5_2, run(m.branch5x5_1, x), out, 64)
            run(m.branch3x3dbl_3, run(m.branch3x3dbl_2, run(m.branch3x3dbl_1, x)), out, 128)
            run(m.branch_pool, ops.pool3x3(x, 1, 1, avg_exclude_pad=True), out, 224)
        elif k == "B":
            OH = (H - 3) // 2 + 1
            out = new(OH, OH, 480 + x.shape[3])
            run(m.branch3x3, x, out, 0)
            run(m.branch3x3dbl_3, run(m.branch3x3dbl_2, run(m.branch3x3dbl_1, x)), out, 384)
            ops.pool3x3(x, 2, 0, out=out, coff=480)
        elif k == "C":
            out = new(H, W, 768)
            run(m.branch1x1, x, out, 0)
            run(m.branch7x7_3, run(m.branch7x7_2, run(m.branch7x7_1, x)), out, 192)
            t = run(m.branch7x7dbl_2, run(m.branch7x7dbl_1, x))
            run(m.branch7x7dbl_5, run(m.branch7x7dbl_4, run(m.branch7x7dbl_3, t)), out, 384)
            run(m.branch_pool, ops.pool3x3(x, 1, 1, avg_exclude_pad=True), out, 576)
        elif k == "D":
            OH = (H - 3) // 2 + 1
            out = new(OH, OH, 512 + x.shape[3])
            run(m.branch3x3_2, run(m.branch3x3_1, x), out, 0)
            run(m.branch7x7x3_4, run(m.branch7x7x3_3, run(m.branch7x7x3_2, run(m.branch7x7x3_1, x))), out, 320)
            ops.pool3x3(x, 2, 0, out=out, coff=512)
        else:           # E1 / E2
            out = new(H, W, 2048)
            run(m.branch1x1, x, out, 0)
            t = run(m.branch3x3_1, x)
            run(m.branch3x3_2a, t, out, 320)
            run(m.branch3x3_2b, t, out, 704)
            t = run(m.branch3x3dbl_2, run(m.branch3x3dbl_1, x))
            run(m.branch3x3dbl_3a, t, out, 1088)
            run(m.branch3x3dbl_3b, t, out, 1472)
            # Mixed_7b: average over the in-bounds pixels; Mixed_7c: MAX pool (the FID model's own quirk, inception.py:303-308)
            pooled = ops.pool3x3(x, 1, 1, avg_exclude_pad=(k == "E1"))
            run(m.branch_pool, pooled, out, 1856)
        return out

    @torch.no_grad()
    def forward(self, inp):
        """inp [B, 3, H, W] in (0, 1) on the device -> list of fp32 NCHW feature maps of the selected blocks (reference :129-163)."""
        if not inp.is_cuda:
            raise DxmiError("pytorch_fid.inception.InceptionV3 runs only on the HIP device path (no CPU fallback)")
        pk = self.packed()
        x = inp.contiguous().float()
        N, _, H, W = x.shape
        OH, OW = (299, 299) if self.resize_input else (H, W)
        h = ops.resize_bilinear_nhwc16(x, OH, OW, normalize=self.normalize_input)
        outp = []
        for idx in range(self.last_needed_block + 1):
            for name in _LAYOUT[idx]:
                m = self._by_name[name]
                h = self._run(pk, m, h) if isinstance(m, BasicConv2d) else self._mixed(pk, m, h)
            if idx in (0, 1):
                h = ops.pool3x3(h, 2, 0)
            if idx == 3:
                feat = ops.global_avgpool(h)
                if idx in self.output_blocks:
                    outp.append(feat.view(N, -1, 1, 1))
            elif idx in self.output_blocks:
                outp.append(ops.nhwc_bf16_to_nchw_f32(h))
        return outp


class FIDInceptionV3(InceptionV3):
    """The extractor `--fid_extractor pytorch_fid.inception:FIDInceptionV3` resolves to (the scripts instantiate a class): InceptionV3
    pool3 features with the FID weights named by DXMI_FID_WEIGHTS (pt_inception-2015-12-05-6726825d.pth).  Without the file this
    raises — scoring FID on an untrained extractor is never what the caller wants."""

    def __init__(self, dims=2048):
        path = os.environ.get("DXMI_FID_WEIGHTS")
        if not path or not os.path.exists(path):
            raise DxmiError("DXMI_FID_WEIGHTS must name the FID Inception weight file (pt_inception-2015-12-05-6726825d.pth); "
                            "it cannot be downloaded into this image")
        super().__init__([InceptionV3.BLOCK_INDEX_BY_DIM[dims]], weights=path)
