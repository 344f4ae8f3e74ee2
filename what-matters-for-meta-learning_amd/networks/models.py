"""Shared building blocks (reference: networks/models.py:27-60 EncoderFC, 195-203 AttnLinear).

These classes are parameter containers with the reference's construction order (so a
seeded model draws identical initial weights) and state_dict keys; their own forward()
runs the mlhot HIP linear kernels, while the fused model path reads the parameters directly.
"""
import torch
from torch import nn

from mlhot.ops import LinearFunction


class EncoderFC(nn.Module):
    """(Linear+ReLU) x len(n_hidden_units_r) then Linear(-> dim_r); keys `layers.{0,2,..}`."""

    def __init__(self, input_dim, n_hidden_units_r, dim_r):
        super().__init__()
        self.input_dim, self.n_hidden_units_r, self.dim_r = input_dim, list(n_hidden_units_r), dim_r
        widths = [input_dim] + self.n_hidden_units_r
        mods = []
        for a, b in zip(widths[:-1], widths[1:]):
            mods += [nn.Linear(a, b), nn.ReLU(inplace=True)]
        mods.append(nn.Linear(widths[-1], dim_r))
        self.layers = nn.Sequential(*mods)

    def linears(self):
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def forward(self, x):
        lins = self.linears()
        for lin in lins[:-1]:
            x = LinearFunction.apply(x, lin.weight, lin.bias, "relu")
        return LinearFunction.apply(x, lins[-1].weight, lins[-1].bias, "none")


class AttnLinear(nn.Module):
    """nn.Linear whose weight is re-drawn N(0, in^-1)  (models.py:195-203)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=True)
        torch.nn.init.normal_(self.linear.weight, std=in_channels ** -0.5)

    def forward(self, x):
        return LinearFunction.apply(x, self.linear.weight, self.linear.bias, "none")
