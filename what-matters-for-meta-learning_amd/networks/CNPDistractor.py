"""Plugin `networks.CNPDistractor` (reference: networks/CNPDistractor.py): the ResNet-encoder CNP of the Distractor task,
whose context labels pass through `transform_y` = Linear(label_dim -> dim_w) before the task encoder - see networks/_resnet_np.py."""
from networks._resnet_np import ResNetNP


class CNPDistractor(ResNetNP):
    ATTENTION = False
    TRANSFORM_Y = True
