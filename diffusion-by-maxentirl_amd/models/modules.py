"""IGEBM value/energy network (`models.modules.IGEBMEncoderV2`) on the gfx950 kernel library.

Drop-in for the reference classes (reference: models/modules.py:28-186): same constructor
arguments, parameter names and shapes.  The torch.nn layers are fp32 parameter containers; the
forward is a fused HIP program over NHWC bf16 activations:
  image conv (K=27 im2col MFMA, +bias, LeakyReLU fused) ->
  6 x [conv3x3+bias+LeakyReLU | conv3x3+bias+skip(+LeakyReLU) | (avgpool2+LeakyReLU)] ->
  ReLU + spatial sum + Linear(256,1) + out_scale in one head kernel.
Only the configuration used by the DxMI configs is implemented on the device
(use_spectral_norm False, n_class None, keepdim False, out_activation linear):
anything else raises at construction.
"""
import torch
import torch.nn as nn

from dxmi_hip import ops
from dxmi_hip._lib import DxmiError


def process_single_t(x, t):
    """Scalar / length-1 t -> int64 vector of len(x) (reference :183-186)."""
    if isinstance(t, int) or len(t.shape) == 0 or len(t) == 1:
        t = torch.ones([x.shape[0]], dtype=torch.long, device=x.device) * t
    return t


def get_activation(s_act):
    """reference :8-26."""
    table = {"relu": lambda: nn.ReLU(inplace=True), "sigmoid": nn.Sigmoid, "softplus": nn.Softplus,
             "tanh": nn.Tanh, "leakyrelu": lambda: nn.LeakyReLU(0.2, inplace=True),
             "softmax": lambda: nn.Softmax(dim=1), "swish": lambda: nn.SiLU(inplace=True)}
    if s_act == "linear":
        return None
    if s_act not in table:
        raise ValueError(f"Unexpected activation: {s_act}")
    return table[s_act]()


class ResBlockV2(nn.Module):
    """Parameter container (reference :28-101): conv1, conv2, optional 1x1 `skip.0` (no bias)."""

    def __init__(self, in_channel, out_channel, n_class=None, downsample=False, use_spectral_norm=True):
        super().__init__()
        if n_class is not None or use_spectral_norm:
            raise NotImplementedError("class-conditional / spectral-norm ResBlockV2 is not used by the DxMI configs")
        self.conv1 = nn.Conv2d(in_channel, out_channel, 3, padding=1, bias=True)
        self.conv2 = nn.Conv2d(out_channel, out_channel, 3, padding=1, bias=True)
        self.class_embed = None
        self.skip = None
        if in_channel != out_channel or downsample:
            self.skip = nn.Sequential(nn.Conv2d(in_channel, out_channel, 1, bias=False))
        self.downsample = downsample


class IGEBMEncoderV2(nn.Module):
    def __init__(self, in_chan=3, out_chan=1, n_class=None, use_spectral_norm=False, keepdim=True,
                 out_activation="linear", avg_pool_dim=1, learn_out_scale=False, nh=128):
        super().__init__()
        if use_spectral_norm or n_class is not None or keepdim or out_activation != "linear" or out_chan != 1:
            raise NotImplementedError("IGEBMEncoderV2: only the DxMI configuration (no spectral norm, n_class None, "
                                      "keepdim False, linear output, out_chan 1) is built on the HIP path")
        self.keepdim, self.use_spectral_norm, self.avg_pool_dim = keepdim, use_spectral_norm, avg_pool_dim
        self.in_chan, self.nh = in_chan, nh
        self.conv1 = nn.Conv2d(in_chan, nh, 3, padding=1)
        self.blocks = nn.ModuleList([
            ResBlockV2(nh, nh, n_class, downsample=True, use_spectral_norm=use_spectral_norm),
            ResBlockV2(nh, nh, n_class, use_spectral_norm=use_spectral_norm),
            ResBlockV2(nh, nh * 2, n_class, downsample=True, use_spectral_norm=use_spectral_norm),
            ResBlockV2(nh * 2, nh * 2, n_class, use_spectral_norm=use_spectral_norm),
            ResBlockV2(nh * 2, nh * 2, n_class, downsample=True, use_spectral_norm=use_spectral_norm),
            ResBlockV2(nh * 2, nh * 2, n_class, use_spectral_norm=use_spectral_norm),
        ])
        self.linear = nn.Linear(nh * 2, out_chan)
        self.out_activation = get_activation(out_activation)
        self.pre_activation = None
        self.learn_out_scale = learn_out_scale
        if learn_out_scale:
            self.out_scale = nn.Linear(1, 1, bias=True)
        self._packed, self._packed_key, self._pack_bufs = None, None, None
        self._packed_t, self._packed_t_key, self._pack_t_bufs = None, None, None

    # ---- bf16 weight fragments, rebuilt when a parameter's version changes
    def packed(self):
        key = tuple((p.data_ptr(), p._version) for p in ops.fast_parameters(self))
        if self._packed is None or key != self._packed_key:
            # parameters updated in place: the same buffers are rewritten (addresses held by a captured hipGraph stay valid)
            reuse = self._pack_bufs if (self._packed is not None and self._same_storage(key, self._packed_key)) else None
            with ops.pack_batch(reuse=reuse) as pb:             # one multi-tensor launch for the net's weights
                pk = {"conv1": ops.pack_conv_weight(self.conv1.weight, k27=(self.in_chan == 3))}
                for i, b in enumerate(self.blocks):
                    pk[i, "conv1"] = ops.pack_conv_weight(b.conv1.weight)
                    pk[i, "conv2"] = ops.pack_conv_weight(b.conv2.weight)
                    if b.skip is not None:
                        pk[i, "skip"] = ops.pack_conv_weight(b.skip[0].weight)
            self._packed, self._packed_key, self._pack_bufs = pk, key, pb.buffers
        return self._packed

    @staticmethod
    def _same_storage(key, old_key):
        return old_key is not None and len(key) == len(old_key) and all(a[0] == b[0] for a, b in zip(key, old_key))

    def refresh_packs(self):
        """Bring the packed-weight sets that exist up to date with the parameters (no-op when they are)."""
        self.packed()
        if self._packed_t is not None:
            self.packed_transposed()

    def prepare_capture(self):
        """Before a StepGraph capture (dxmi_hip/graph.py)."""
        self.refresh_packs()

    def packed_transposed(self):
        """Transpose-flipped fragments: the data-gradient operators of the block convs."""
        key = tuple((p.data_ptr(), p._version) for p in ops.fast_parameters(self))
        if self._packed_t is None or key != self._packed_t_key:
            pk = {}
            reuse = self._pack_t_bufs if (self._packed_t is not None and self._same_storage(key, self._packed_t_key)) else None
            with ops.pack_batch(reuse=reuse) as pb:
                for i, b in enumerate(self.blocks):
                    pk[i, "conv1"] = ops.pack_conv_weight(b.conv1.weight, transpose_flip=True)
                    pk[i, "conv2"] = ops.pack_conv_weight(b.conv2.weight, transpose_flip=True)
                    if b.skip is not None:
                        pk[i, "skip"] = ops.pack_conv_weight(b.skip[0].weight, transpose_flip=True)
            self._packed_t, self._packed_t_key, self._pack_t_bufs = pk, key, pb.buffers
        return self._packed_t

    def train(self, mode=True):
        """The trainers switch the value net between eval() and train() around every TD target (reference trainer.py:288-295), 24
        recursive walks of ~50 modules per train step; no module of this encoder reads the flag (no dropout, no batch statistics,
        spectral norm unsupported), so the flags are set through a flat list."""
        mods = self.__dict__.get("_dxmi_all_modules")
        if mods is None:
            mods = list(self.modules())
            self.__dict__["_dxmi_all_modules"] = mods
        for m in mods:
            m.__dict__["training"] = mode
        return self

    def forward(self, input, y=None):
        if not input.is_cuda:
            raise DxmiError("models.modules.IGEBMEncoderV2 runs only on the HIP device path (no CPU fallback)")
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in ops.fast_parameters(self))):
            from .value_train import forward_with_grad  # autograd wrapper around the HIP kernels
            return forward_with_grad(self, input)
        return self.forward_inference(input)

    def forward_pair(self, x_free, x_grad):
        """(net(x_free) without a graph, net(x_grad) with one) from ONE forward over the concatenated batch: the TD steps of the
        trainers evaluate the target v(next_state) and the prediction v(state) with the same parameters (reference
        models/DxMI/trainer.py:288-300 switches eval() / train() in between, which no layer of this encoder reads), and the small
        maps of this net leave a 256-image launch one tile per CU with every fixed cost exposed.  Each image's value is bitwise
        what its own forward gives where the library's kernel choice does not depend on the batch (default knobs)."""
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in ops.fast_parameters(self))) or x_free.requires_grad or x_grad.requires_grad:
            with torch.no_grad():
                t = self.forward(x_free)
            return t, self.forward(x_grad)
        from .value_train import forward_with_grad
        n0 = x_free.shape[0]
        res = forward_with_grad(self, torch.cat((x_free.detach(), x_grad), 0), nfree=n0)
        return res[:n0].detach(), res[n0:]

    def forward_pair_packed(self, x_cat, n_free):
        """forward_pair on an already concatenated batch: values of all rows, autograd graph through rows n_free.. only."""
        from .value_train import forward_with_grad
        return forward_with_grad(self, x_cat, nfree=n_free)

    @torch.no_grad()
    def forward_inference(self, input):
        pk = self.packed()
        x = input.contiguous().float()
        if pk["conv1"].k27:
            out = ops.conv2d(x, pk["conv1"], bias=self.conv1.bias, act=ops.ACT_LEAKY02)
        else:
            out = ops.conv2d(ops.nchw_f32_to_nhwc_bf16(x), pk["conv1"], bias=self.conv1.bias, act=ops.ACT_LEAKY02)
        for i, b in enumerate(self.blocks):
            h = ops.conv2d(out, pk[i, "conv1"], bias=b.conv1.bias, act=ops.ACT_LEAKY02)
            skip = ops.conv2d(out, pk[i, "skip"]) if b.skip is not None else out
            if b.downsample:
                h = ops.conv2d(h, pk[i, "conv2"], bias=b.conv2.bias, residual=skip)
                out = ops.pool_act(h, True, ops.ACT_LEAKY02)
            else:
                out = ops.conv2d(h, pk[i, "conv2"], bias=b.conv2.bias, residual=skip, act=ops.ACT_LEAKY02)
        ow, ob = (self.out_scale.weight, self.out_scale.bias) if self.learn_out_scale else (None, None)
        res = ops.value_head(out, self.linear.weight, self.linear.bias, ow, ob)
        self.pre_activation = res
        return res

    def load_pretrained(self, ckpt):
        """Load `conv1.*` and `blocks.*` from a checkpoint whose keys carry a 4-char prefix
        ("net.") (reference :165-180)."""
        conv1, blocks = {}, {}
        for k, v in ckpt["state_dict"].items():
            k_ = k[4:]
            if k_.startswith("conv1"):
                conv1[k_[len("conv1."):]] = v
            elif k_.startswith("blocks"):
                blocks[k_[len("blocks."):]] = v
        self.conv1.load_state_dict(conv1)
        self.blocks.load_state_dict(blocks)
