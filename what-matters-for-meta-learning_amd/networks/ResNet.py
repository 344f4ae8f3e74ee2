"""BN-free ResNet trunk of the ShapeNet3D / Distractor encoders (reference: networks/ResNet.py:37-215).

Parameter containers with the reference's construction and (re-)initialisation order - every conv is
an nn.Conv2d created in the same sequence and then re-drawn with kaiming_normal_(fan_out, relu) in
`modules()` order - so seeded weights and state_dict keys (`layer{1..4}.0.conv{1,2}`,
`layer{1..4}.0.downsample.0`, the unused `fc`) match.  forward() runs the mlhot conv kernels.
"""
import torch.nn as nn

from mlhot.ops import AddReluFunction, Conv2dFunction


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=True)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=True)


def run_conv(conv, x, relu=False, weight=None, bias=None):
    """conv: an nn.Conv2d container (square kernel / stride / padding); weight / bias override its own."""
    w = conv.weight if weight is None else weight
    b = conv.bias if bias is None else bias
    return Conv2dFunction.apply(x, w, b, conv.stride[0], conv.padding[0], relu)


class BasicBlock(nn.Module):
    """relu(conv3x3_s(x)) -> conv3x3_1 -> + skip(x) -> relu   (ResNet.py:58-72; no batch norm)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, taps=None):
        out = run_conv(self.conv1, x, relu=True)
        if taps is not None:
            taps.append(out)
        out = run_conv(self.conv2, out)
        identity = run_conv(self.downsample[0], x) if self.downsample is not None else x
        y = AddReluFunction.apply(out, identity)
        if taps is not None:
            taps.append(y)
        return y


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, pretrained=False, progress=False, skip_kernel=1):
        super().__init__()
        if pretrained:
            raise NotImplementedError("pretrained weights are never requested by the reference models (models.py:90)")
        self.inplanes = 64
        self.skip_kernel = skip_kernel
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, 64, layers[0], stride=2)
        self.layer2 = self._make_layer(block, 64, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 64, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 64, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.adaptmax = nn.AdaptiveMaxPool2d((2, 2))
        self.fc = nn.Linear(512 * block.expansion, num_classes)   # unused by the models, but part of the state_dict
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.inplanes = 64

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def trunk(self, x, taps=None):
        """taps: optional list that receives every post-ReLU activation (diagnostics / routing tests)."""
        if taps is not None:
            taps.append(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk(x, taps)
        return x
