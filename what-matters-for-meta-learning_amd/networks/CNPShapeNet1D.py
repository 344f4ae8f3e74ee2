"""Plugin `networks.CNPShapeNet1D` (reference: networks/CNPShapeNet1D.py) - see networks/_vanilla.py."""
from networks._vanilla import VanillaNP


class CNPShapeNet1D(VanillaNP):
    ATTENTION = False
    OUT_TANH = True
