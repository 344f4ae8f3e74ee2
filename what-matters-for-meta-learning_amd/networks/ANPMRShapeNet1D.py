"""Plugin `networks.ANPMRShapeNet1D` (reference: networks/ANPMRShapeNet1D.py): vanilla-encoder ANP with the
Bayes-by-backprop image encoder (meta-regularisation); see networks/_vanilla_mr.py."""
from networks._vanilla_mr import BBBEncoder, VanillaMR  # noqa: F401


class ANPMRShapeNet1D(VanillaMR):
    ATTENTION = True
    OUT_TANH = True
    REDRAW_DECODER0 = False
