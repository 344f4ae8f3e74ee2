"""Plugin `networks.CNPMR` (reference: networks/CNPMR.py): vanilla-encoder CNP with the
Bayes-by-backprop image encoder (meta-regularisation); see networks/_vanilla_mr.py."""
from networks._vanilla_mr import BBBEncoder, VanillaMR  # noqa: F401


class CNPMR(VanillaMR):
    ATTENTION = False
    OUT_TANH = False
    REDRAW_DECODER0 = False
