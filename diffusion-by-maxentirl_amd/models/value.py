"""Value plugin (`models.value.TimeIndependentValue`), reference: models/value.py:3-15."""
import torch.nn as nn


class TimeIndependentValue(nn.Module):
    """V(x, t, y=None): ignores t, forwards to the wrapped energy network (HIP path)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x, t, y=None):
        return self.net(x, y) if y is not None else self.net(x)

    def forward_pair(self, x_free, t_free, x_grad, t_grad, y=None):
        """(V(x_free, t_free) without a graph, V(x_grad, t_grad) with one) in one forward of the wrapped network (t is ignored):
        the TD target and the TD prediction of one step (models.modules.IGEBMEncoderV2.forward_pair)."""
        if y is not None or not hasattr(self.net, "forward_pair"):
            import torch
            was_training = self.net.training
            self.net.eval()                    # the reference evaluates the TD target under eval() (trainer.py:288-300)
            try:
                with torch.no_grad():
                    tgt = self.forward(x_free, t_free, y=y)
            finally:
                self.net.train(was_training)
            return tgt, self.forward(x_grad, t_grad, y=y)
        return self.net.forward_pair(x_free, x_grad)

    def forward_pair_packed(self, x_cat, n_free):
        """forward_pair on an already concatenated batch [x_free | x_grad] (the trainers' fused TD step writes the two halves in
        place): -> V of all 2B rows, with a graph through the second half only.  None when the wrapped network has no such form."""
        fn = getattr(self.net, "forward_pair_packed", None)
        return None if fn is None else fn(x_cat, n_free)

    def load_pretrained(self, ckpt):
        self.net.load_pretrained(ckpt)
