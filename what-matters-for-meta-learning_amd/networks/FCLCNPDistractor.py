"""Plugin `networks.FCLCNPDistractor` (reference: networks/FCLCNPDistractor.py): CNPDistractor with functional contrastive
learning - the target set goes through the image / task encoders and the aggregator as well, and the NT-Xent term between the
context-set and the target-set task embeddings (trainer/losses.py:83-88) comes back as 4th value; see networks/_resnet_np.py."""
from networks._resnet_np import ResNetNP


class FCLCNPDistractor(ResNetNP):
    ATTENTION = False
    TRANSFORM_Y = True
    CONTRASTIVE = True
