"""`networks.MMAMLShapeNet1D` of the reference is OUTSIDE the accelerated hot path (the task-batched CNP/ANP
forward+backward): MMAML inner-loop adaptation (its task embedding, networks.conv_embedding_model.ConvEmbeddingModel, IS implemented).  The module exists so a config naming it fails loudly and clearly."""
from torch import nn


class MMAMLShapeNet1D(nn.Module):
    def __init__(self, config=None, *args, **kwargs):
        raise NotImplementedError("method 'MMAMLShapeNet1D' (MMAML inner-loop adaptation) is not part of the MI355X hot-path build; "
                                  "in scope: CNP*/ANP* (vanilla, ResNet, MR and Distractor variants) - see INTEGRATION.md")
