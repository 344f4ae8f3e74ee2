"""`networks.FCLCNPShapeNet1D` of the reference is OUTSIDE the accelerated hot path (the task-batched CNP/ANP
forward+backward): functional contrastive learning (NT-Xent loss on a 4-tuple forward).  The module exists so a config naming it fails loudly and clearly."""
from torch import nn


class FCLCNPShapeNet1D(nn.Module):
    def __init__(self, config=None, *args, **kwargs):
        raise NotImplementedError("method 'FCLCNPShapeNet1D' (functional contrastive learning) is not part of the MI355X hot-path build; "
                                  "in scope: CNP*/ANP* (vanilla, ResNet, MR and Distractor variants) - see INTEGRATION.md")
