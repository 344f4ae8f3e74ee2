"""Plugin `networks.ANPShapeNet1D` (reference: networks/ANPShapeNet1D.py) - see networks/_vanilla.py."""
from networks._vanilla import VanillaNP


class ANPShapeNet1D(VanillaNP):
    ATTENTION = True
    OUT_TANH = True
