"""Plugin `networks.FCLANP` (reference: networks/FCLANP.py): the ResNet-encoder ANP with functional contrastive learning - the
forward takes the target labels as 4th argument and returns the NT-Xent term over the per-target attention outputs
(trainer/losses.py:91-99) as 4th value.  Same parameters, construction order and kernels as `networks.ANP`; see
networks/_resnet_np.py."""
from networks._resnet_np import ResNetNP


class FCLANP(ResNetNP):
    ATTENTION = True
    CONTRASTIVE = True
