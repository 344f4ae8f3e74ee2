"""Plugin `networks.ANPMR` (reference: networks/ANPMR.py): vanilla-encoder ANP with the
Bayes-by-backprop image encoder (meta-regularisation); see networks/_vanilla_mr.py."""
from networks._vanilla_mr import BBBEncoder, VanillaMR  # noqa: F401


class ANPMR(VanillaMR):
    ATTENTION = True
    OUT_TANH = False
    REDRAW_DECODER0 = False
