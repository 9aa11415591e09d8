"""ResNet-50 (v1.5: stride on the 3x3) skeleton: layer sizes only, `pretrained` ignored."""
import torch.nn as nn


class _Block(nn.Module):
    def __init__(self, cin, width, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, 4 * width, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * width)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * width, 1, stride, bias=False), nn.BatchNorm2d(4 * width))


def _stage(cin, width, n, stride):
    return nn.Sequential(_Block(cin, width, stride, True), *[_Block(4 * width, width, 1, False) for _ in range(n - 1)])


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = _stage(64, 64, 3, 1)
        self.layer2 = _stage(256, 128, 4, 2)
        self.layer3 = _stage(512, 256, 6, 2)
        self.layer4 = _stage(1024, 512, 3, 2)


def resnet50(pretrained=False, **kw):
    return _ResNet50()
