"""`networks.VanillaMAML` of the reference is OUTSIDE the accelerated hot path (the task-batched CNP/ANP
forward+backward): MAML inner-loop adaptation.  The module exists so a config naming it fails loudly and clearly."""
from torch import nn


class VanillaMAML(nn.Module):
    def __init__(self, config=None, *args, **kwargs):
        raise NotImplementedError("method 'VanillaMAML' (MAML inner-loop adaptation) is not part of the MI355X hot-path build; "
                                  "in scope: CNP*/ANP* (vanilla, ResNet, MR and Distractor variants) - see INTEGRATION.md")
