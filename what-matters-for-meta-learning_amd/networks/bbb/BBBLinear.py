"""Bayes-by-backprop linear layer (reference: networks/bbb/BBBLinear.py:33-100).

Same contract as BBBConv2d: W = W_mu + eps * log1p(exp(W_rho)) re-sampled on every forward with eps from
the torch CPU generator (weight, then bias), KL(prior || posterior) as the reference literally computes it;
the sample / KL and their backward are the mlhot_bbb_sample kernels, the product is mlhot_linear.
"""
import torch
from torch.nn import Parameter

from mlhot.ops import BBBSampleFunction, LinearFunction
from .BBBConv import PRIORS
from . import eps
from .misc import ModuleWrapper


class BBBLinear(ModuleWrapper):
    def __init__(self, in_features, out_features, bias=True, priors=None, device="cpu"):
        super().__init__()
        self.in_features, self.out_features, self.use_bias = in_features, out_features, bias
        pri = dict(PRIORS, **(priors or {}))
        if (pri["prior_mu"], pri["prior_sigma"]) != (0, 0.1):
            raise NotImplementedError("the KL kernel is written for the reference's N(0, 0.1^2) prior")
        self.posterior_mu_initial, self.posterior_rho_initial = pri["posterior_mu_initial"], pri["posterior_rho_initial"]
        self.W_mu = Parameter(torch.empty(out_features, in_features))
        self.W_rho = Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias_mu = Parameter(torch.empty(out_features))
            self.bias_rho = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias_mu", None)
            self.register_parameter("bias_rho", None)
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
        """(weight, bias, kl) with the reference's draw order."""
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
        return LinearFunction.apply(x, weight, bias, "none")

    def kl_loss(self):
        return self._kl

    def release_kl_graph(self):
        """Keep the value of the last KL but drop its autograd graph.  A layer that held on to the graph would keep the
        gradient-accumulation nodes of its parameters alive from one forward to the next, pinned to the stream of the
        FIRST forward - which breaks hipGraph capture of a later step (the engine then syncs with that old stream)."""
        if self._kl is not None:
            self._kl = self._kl.detach()
