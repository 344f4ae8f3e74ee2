"""Containers of the BBB encoders (reference: networks/bbb/misc.py:23-54)."""
from torch import nn


class ModuleWrapper(nn.Module):
    """Runs its children in order and returns (output, summed KL of every BBB layer underneath)."""

    def set_flag(self, flag_name, value):
        setattr(self, flag_name, value)
        for m in self.children():
            if hasattr(m, "set_flag"):
                m.set_flag(flag_name, value)

    def forward(self, x):
        for module in self.children():
            x = module(x)
        kl = 0.0
        for module in self.modules():
            if hasattr(module, "kl_loss"):
                kl = kl + module.kl_loss()
                release = getattr(module, "release_kl_graph", None)
                if release is not None:
                    release()
        return x, kl


class FlattenLayer(ModuleWrapper):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features

    def forward(self, x):
        return x.reshape(-1, self.num_features)
