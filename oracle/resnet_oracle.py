"""CPU restatement of the reference's offline feature extractor (test infrastructure -- see oracle/__init__).

The reference builds ``torchvision.models.resnet101(pretrained=True)``, replaces ``layer4`` / ``avgpool`` / ``fc`` by
``nn.Identity`` and runs it in eval mode under ``no_grad`` (reference scripts/preprocess/extract_features.py:98-105,
124-131).  The network itself lives in a third-party dependency that is absent from /root/reference and from this
image: torchvision, pinned at 0.5.0 (reference requirements.txt:7).  This file restates that version's published
definition -- ``ResNet._forward_impl`` (conv1, bn1, relu, maxpool, layer1..3) and ``Bottleneck.forward`` (1x1 -> 3x3
with the block's stride -> 1x1 x4, identity or 1x1-stride downsample, add, ReLU; "ResNet v1.5": the stride sits on the
3x3 convolution) with ``layers = [3, 4, 23, 3]`` -- as plain torch CPU fp32 ops over a ``state_dict`` with torchvision's
key names.

PARITY UNPINNED: there are no torchvision sources, no pretrained weights and no golden vectors in the reference to pin
this restatement against (the reference's tests hold nothing for the extractor); tests compare the HIP path with it on
random weights.  What anchors it is the call site (the three Identity replacements, eval mode, 224x224 inputs giving
(1024, 14, 14) features: extract_features.py:3, 60-61, 113-115) and the shapes of torchvision's checkpoint keys.
"""
from typing import Dict

import torch
import torch.nn.functional as F

LAYERS = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 23, 2))  # (name, planes, blocks, first stride)
EPS = 1e-5  # nn.BatchNorm2d default

# reference extract_features.py:70-73 (the third std really is 0.224 there)
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.224)


def preprocess(images_uint8: torch.Tensor) -> torch.Tensor:
    """ToTensor + Normalize of the reference's transform on (N, 3, H, W) uint8 images (Resize is the caller's)."""
    x = images_uint8.float() / 255.0
    mean = torch.tensor(MEAN).view(1, 3, 1, 1)
    std = torch.tensor(STD).view(1, 3, 1, 1)
    return (x - mean) / std


def _bn(sd: Dict[str, torch.Tensor], name: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training=False, eps=EPS)


def _bottleneck(sd, prefix: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    out = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"])))
    out = F.relu(_bn(sd, prefix + ".bn2", F.conv2d(out, sd[prefix + ".conv2.weight"], stride=stride, padding=1)))
    out = _bn(sd, prefix + ".bn3", F.conv2d(out, sd[prefix + ".conv3.weight"]))
    identity = x
    if prefix + ".downsample.0.weight" in sd:
        identity = _bn(sd, prefix + ".downsample.1", F.conv2d(x, sd[prefix + ".downsample.0.weight"], stride=stride))
    return F.relu(out + identity)


def resnet101_stage3(sd: Dict[str, torch.Tensor], images: torch.Tensor) -> torch.Tensor:
    """(N, 3, 224, 224) normalised images -> (N, 1024, 14, 14) features."""
    with torch.no_grad():
        x = F.relu(_bn(sd, "bn1", F.conv2d(images, sd["conv1.weight"], stride=2, padding=3)))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for name, _, blocks, stride in LAYERS:
            for b in range(blocks):
                x = _bottleneck(sd, "%s.%d" % (name, b), x, stride if b == 0 else 1)
        return x
