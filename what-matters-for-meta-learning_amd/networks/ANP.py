"""Plugin `networks.ANP` (reference: networks/ANP.py) - see networks/_resnet_np.py."""
from networks._resnet_np import ResNetNP


class ANP(ResNetNP):
    ATTENTION = True
