"""Bayes-by-backprop convolution (reference: networks/bbb/BBBConv.py:38-108).

W = W_mu + eps * log1p(exp(W_rho)) is re-sampled on EVERY forward (also in eval: `sample=True` is the
reference's default), with eps drawn on the torch CPU generator in the reference's order (weight, then
bias) so seeded runs reproduce; the sample, the KL term and their backward are the mlhot_bbb_sample
kernels, the convolution is mlhot_conv2d.  `fuse_relu` lets the caller fold the ReLU that follows.
"""
import torch
from torch.nn import Parameter

from mlhot.ops import BBBSampleFunction, Conv2dFunction
from . import eps
from .misc import ModuleWrapper

PRIORS = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-3, 0.1)}


class BBBConv2d(ModuleWrapper):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True, priors=None,
                 device="cpu"):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("dilation is never used by the reference models")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.stride, self.padding, self.use_bias = stride, padding, bias
        pri = dict(PRIORS, **(priors or {}))
        if (pri["prior_mu"], pri["prior_sigma"]) != (0, 0.1):
            raise NotImplementedError("the KL kernel is written for the reference's N(0, 0.1^2) prior")
        self.posterior_mu_initial, self.posterior_rho_initial = pri["posterior_mu_initial"], pri["posterior_rho_initial"]
        self.W_mu = Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.W_rho = Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        if bias:
            self.bias_mu = Parameter(torch.empty(out_channels))
            self.bias_rho = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias_mu", None)
            self.register_parameter("bias_rho", None)
        self.fuse_relu = False
        self._kl = None
        self.presampled = None
        self.reset_parameters()

    def reset_parameters(self):
        self.W_mu.data.normal_(*self.posterior_mu_initial)
        self.W_rho.data.normal_(*self.posterior_rho_initial)
        if self.use_bias:
            self.bias_mu.data.normal_(*self.posterior_mu_initial)
            self.bias_rho.data.normal_(*self.posterior_rho_initial)

    def sample(self):
        """(weight, bias, kl) with the reference's draw order (weight eps, then bias eps)."""
        dev = self.W_mu.device
        w_eps = eps.draw(self.W_mu.size(), dev)
        weight, kl = BBBSampleFunction.apply(self.W_mu, self.W_rho, w_eps)
        bias = None
        if self.use_bias:
            b_eps = eps.draw(self.bias_mu.size(), dev)
            bias, kl_b = BBBSampleFunction.apply(self.bias_mu, self.bias_rho, b_eps)
            kl = kl + kl_b
        self._kl = kl
        return weight, bias, kl

    def forward(self, x, sample=True):
        if self.presampled is not None:                    # sample_all() drew this forward's weights already
            (weight, bias), self.presampled = self.presampled, None
        else:
            weight, bias, _ = self.sample()
        return Conv2dFunction.apply(x, weight, bias, self.stride, self.padding, self.fuse_relu)

    def kl_loss(self):
        return self._kl

    def release_kl_graph(self):
        """Keep the value of the last KL but drop its autograd graph.  A layer that held on to the graph would keep the
        gradient-accumulation nodes of its parameters alive from one forward to the next, pinned to the stream of the
        FIRST forward - which breaks hipGraph capture of a later step (the engine then syncs with that old stream)."""
        if self._kl is not None:
            self._kl = self._kl.detach()
