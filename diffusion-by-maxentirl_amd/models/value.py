"""Value plugin (`models.value.TimeIndependentValue`), reference: models/value.py:3-15."""
import torch.nn as nn


class TimeIndependentValue(nn.Module):
    """V(x, t, y=None): ignores t, forwards to the wrapped energy network (HIP path)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x, t, y=None):
        return self.net(x, y) if y is not None else self.net(x)

    def load_pretrained(self, ckpt):
        self.net.load_pretrained(ckpt)
