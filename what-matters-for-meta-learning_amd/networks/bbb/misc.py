"""Containers of the BBB encoders (reference: networks/bbb/misc.py:23-54)."""
from torch import nn

from . import eps


def sample_twice(layers):
    """Two independent weight / bias samples of `layers` (a Bayes-by-backprop encoder that runs twice per step: context images,
    then target images) in ONE launch pair.  Draw order = the reference's: every layer of the first pass (weight, then bias),
    then every layer of the second pass.  Returns (tensors of sample 1, tensors of sample 2, KL) with the tensors in layer
    order [w, b, w, b, ...]; the KL does not depend on eps, so it is the KL either pass would report (bbb/misc.py:40-44)."""
    from mlhot.ops import BBBSampleMultiFunction
    if any(not layer.use_bias for layer in layers) or 2 * len(layers) > 32:
        raise ValueError("sample_twice: expects <= 16 layers, all with a bias")
    mu_rho, eps_list = [], []
    for _ in range(2):
        for layer in layers:
            dev = layer.W_mu.device
            eps_list.append(eps.draw(layer.W_mu.size(), dev))
            eps_list.append(eps.draw(layer.bias_mu.size(), dev))
    for layer in layers:
        mu_rho += [layer.W_mu, layer.W_rho, layer.bias_mu, layer.bias_rho]
        layer._kl = None
    *ws, kl = BBBSampleMultiFunction.apply(eps_list, *mu_rho)
    k = 2 * len(layers)
    return list(ws[:k]), list(ws[k:]), kl


def sample_all(layers):
    """Weight and bias samples of `layers` (BBBConv2d / BBBLinear, in the order their forwards will run) in ONE launch pair
    instead of two per tensor.  The eps draws happen here, layer by layer, weight then bias - the order and the generator state the
    reference's per-layer forwards would see (bbb/BBBConv.py:88-95) - and every layer is handed its sample for its next forward.
    Returns the summed KL of all layers (what ModuleWrapper.forward adds up, bbb/misc.py:40-44)."""
    from mlhot.ops import BBBSampleMultiFunction
    total, todo = 0.0, list(layers)
    while todo:
        chunk, todo = todo[:16], todo[16:]                 # <= 32 tensors per call
        eps_list, mu_rho, slots = [], [], []
        for layer in chunk:
            dev = layer.W_mu.device
            eps_list.append(eps.draw(layer.W_mu.size(), dev))
            mu_rho += [layer.W_mu, layer.W_rho]
            slots.append((layer, "w"))
            if layer.use_bias:
                eps_list.append(eps.draw(layer.bias_mu.size(), dev))
                mu_rho += [layer.bias_mu, layer.bias_rho]
                slots.append((layer, "b"))
        *ws, kl = BBBSampleMultiFunction.apply(eps_list, *mu_rho)
        got = {}
        for (layer, kind), w in zip(slots, ws):
            got.setdefault(id(layer), [layer, None, None])[1 if kind == "w" else 2] = w
        for layer, w, b in got.values():
            layer.presampled = (w, b)
            layer._kl = None                               # the KL of a batched sample is the caller's total
        total = total + kl
    return total


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
