"""Operand / storage rounding model shared by the oracle networks (test infrastructure)."""
import torch


class Precision:
    """mode "fp32": identity everywhere (reference arithmetic, models/* are fp32 end to end).
    mode "bf16": mirrors the HIP pipeline's storage points —
        act(x): an activation tensor written to HBM (NHWC bf16 between kernels)
        w(x)  : a conv / linear weight operand (packed to bf16 once)
        p(x)  : the attention probabilities fed to the PV MFMA
    Accumulation stays fp32 in both modes."""

    def __init__(self, mode="fp32"):
        assert mode in ("fp32", "bf16")
        self.mode = mode

    def _r(self, x):
        return x.to(torch.bfloat16).to(torch.float32) if self.mode == "bf16" else x

    act = _r
    w = _r
    p = _r
