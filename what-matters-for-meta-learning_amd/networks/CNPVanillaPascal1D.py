"""Plugin `networks.CNPVanillaPascal1D` (reference: networks/CNPVanillaPascal1D.py) - see networks/_vanilla.py."""
from networks._vanilla import VanillaNP


class CNPVanillaPascal1D(VanillaNP):
    ATTENTION = False
    OUT_TANH = False
