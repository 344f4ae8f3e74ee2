"""Plugin `networks.CondNeuralProcess` (reference: networks/CondNeuralProcess.py) - see networks/_resnet_np.py."""
from networks._resnet_np import ResNetNP


class CondNeuralProcess(ResNetNP):
    ATTENTION = False
