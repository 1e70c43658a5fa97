"""Backbone + dual feature pyramid: parameter containers with the reference's module tree.

Module / parameter names, shapes and registration order equal ``network/fpn.py`` of the reference
(Bottleneck :9-34, FPN :37-126, FPN50/FPN101 :128-134), so ``state_dict()`` keys match the
reference (and torchvision ResNet keys for the bottom-up part: ``model.fpn.load_state_dict(
resnet_zoo, strict=False)`` keeps working, training/multipose_keypoint_train.py:73-75).

These classes hold parameters only.  The arithmetic runs in ``engine.Engine`` on HIP kernels;
``FPN.forward`` is provided for API parity and routes through the owning ``poseNet`` engine.
"""
import torch
import torch.nn as nn


def _conv(cin, cout, k, stride=1, padding=0, bias=True):
    # created on the meta device: poseNet materialises every parameter inside its flat arena
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=padding, bias=bias, device="meta")


def _bn(c):
    return nn.BatchNorm2d(c, device="meta")


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1):
        super(Bottleneck, self).__init__()
        self.conv1 = _conv(in_planes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = _conv(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = _conv(planes, self.expansion * planes, 1, bias=False)
        self.bn3 = _bn(self.expansion * planes)
        self.downsample = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.downsample = nn.Sequential(
                _conv(in_planes, self.expansion * planes, 1, stride=stride, bias=False),
                _bn(self.expansion * planes),
            )


class FPN(nn.Module):
    def __init__(self, block, num_blocks):
        super(FPN, self).__init__()
        self.in_planes = 64
        self.conv1 = _conv(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = _bn(64)
        # bottom-up
        self.layer1 = self._make_layer(block, 64, num_blocks[0], stride=1)
        self.layer2 = self._make_layer(block, 128, num_blocks[1], stride=2)
        self.layer3 = self._make_layer(block, 256, num_blocks[2], stride=2)
        self.layer4 = self._make_layer(block, 512, num_blocks[3], stride=2)
        # detection (RetinaNet) pyramid
        self.conv6 = _conv(2048, 256, 3, stride=2, padding=1)
        self.conv7 = _conv(256, 256, 3, stride=2, padding=1)
        self.latlayer1 = _conv(2048, 256, 1)
        self.latlayer2 = _conv(1024, 256, 1)
        self.latlayer3 = _conv(512, 256, 1)
        self.toplayer0 = _conv(256, 256, 3, padding=1)
        self.toplayer1 = _conv(256, 256, 3, padding=1)
        self.toplayer2 = _conv(256, 256, 3, padding=1)
        # keypoint pyramid
        self.toplayer = _conv(2048, 256, 1)
        self.flatlayer1 = _conv(1024, 256, 1)
        self.flatlayer2 = _conv(512, 256, 1)
        self.flatlayer3 = _conv(256, 256, 1)
        self.smooth1 = _conv(256, 256, 3, padding=1)
        self.smooth2 = _conv(256, 256, 3, padding=1)
        self.smooth3 = _conv(256, 256, 3, padding=1)
        self._owner = None

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def forward(self, x):
        """[[fp2,fp3,fp4,fp5],[p3,p4,p5,p6,p7]] as f32 NCHW-shaped tensors (fpn.py:126); inference only."""
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("FPN is a parameter container; it computes only as part of poseNet")
        return owner._fpn_features(x)


def FPN50():
    return FPN(Bottleneck, [3, 4, 6, 3])


def FPN101():
    return FPN(Bottleneck, [3, 4, 23, 3])
