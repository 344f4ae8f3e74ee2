"""Plugin `networks.ANPDistractor` (reference: networks/ANPDistractor.py): the ResNet-encoder ANP of the Distractor task,
whose context labels pass through `transform_y` = Linear(label_dim -> dim_w) before the task encoder - see networks/_resnet_np.py."""
from networks._resnet_np import ResNetNP


class ANPDistractor(ResNetNP):
    ATTENTION = True
    TRANSFORM_Y = True
