"""Plugin `networks.ANPVanillaPascal1D` (reference: networks/ANPVanillaPascal1D.py) - see networks/_vanilla.py."""
from networks._vanilla import VanillaNP


class ANPVanillaPascal1D(VanillaNP):
    ATTENTION = True
    OUT_TANH = False
