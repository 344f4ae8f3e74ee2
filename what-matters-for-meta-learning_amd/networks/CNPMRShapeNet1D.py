"""Plugin `networks.CNPMRShapeNet1D` (reference: networks/CNPMRShapeNet1D.py): vanilla-encoder CNP with the
Bayes-by-backprop image encoder (meta-regularisation); see networks/_vanilla_mr.py."""
from networks._vanilla_mr import BBBEncoder, VanillaMR  # noqa: F401


class CNPMRShapeNet1D(VanillaMR):
    ATTENTION = False
    OUT_TANH = True
    REDRAW_DECODER0 = True
