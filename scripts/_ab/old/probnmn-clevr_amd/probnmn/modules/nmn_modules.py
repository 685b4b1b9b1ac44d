"""The seven neural modules of the NMN -- class surface of the reference's
``probnmn.modules.nmn_modules`` (reference: probnmn/modules/nmn_modules.py:11-251), computed by
the gfx950 kernels in ``libprobnmn_hip.so``.

Constructors, parameter names (``conv1`` .. ``conv6``, ``conv``, ``projection``), parameter
shapes and initialisation calls are the reference's, made in the same order, so a given
``torch.manual_seed`` yields the same initial weights and ``state_dict`` keys are interchangeable.
``forward`` takes the reference's NCHW tensors; it needs ``dim == 128``, 14x14 maps and a ROCm
device, and raises otherwise -- there is no eager fallback.

Inside :class:`~probnmn.models.nmn.NeuralModuleNetwork` these classes are parameter holders only:
the network batches all module calls of a step through the grouped kernels
(``probnmn.runtime``).  Standalone calls go through ``probnmn.runtime.ops`` (same kernels, one
launch per primitive, full autograd).
"""
import torch
from torch import nn


def _ops():
    from probnmn.runtime import ops  # deferred: importing this module must not need the GPU library

    return ops


class AndModule(nn.Module):
    """Set intersection of two attentions: elementwise minimum (reference :25-27)."""

    def forward(self, attn1, attn2):
        return _ops().minmax(attn1, attn2, is_max=False)


class OrModule(nn.Module):
    """Set union of two attentions: elementwise maximum (reference :43-45)."""

    def forward(self, attn1, attn2):
        return _ops().minmax(attn1, attn2, is_max=True)


class AttentionModule(nn.Module):
    """(features, attention) -> attention: mask, two 3x3 convs + ReLU, 1x1 conv to one channel,
    sigmoid (reference :72-87)."""

    def __init__(self, dim: int):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(dim, 1, kernel_size=1, padding=0)
        nn.init.kaiming_normal_(self.conv1.weight)
        nn.init.kaiming_normal_(self.conv2.weight)
        nn.init.kaiming_normal_(self.conv3.weight)
        self.dim = dim

    def forward(self, feats, attn):
        ops = _ops()
        x = ops.conv3x3_relu(feats, self.conv1.weight, self.conv1.bias, mask=attn)
        x = ops.conv3x3_relu(x, self.conv2.weight, self.conv2.bias)
        return ops.dot1_sigmoid(x, self.conv3.weight, self.conv3.bias)


class QueryModule(nn.Module):
    """(features, attention) -> encoding: mask, two 3x3 convs + ReLU (reference :111-123)."""

    def __init__(self, dim: int):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        nn.init.kaiming_normal_(self.conv1.weight)
        nn.init.kaiming_normal_(self.conv2.weight)
        self.dim = dim

    def forward(self, feats, attn):
        ops = _ops()
        x = ops.conv3x3_relu(feats, self.conv1.weight, self.conv1.bias, mask=attn)
        return ops.conv3x3_relu(x, self.conv2.weight, self.conv2.bias)


class RelateModule(nn.Module):
    """(features, attention) -> attention through dilated 3x3 convs (dilation 1,2,4,8,1), then the
    one-channel head (reference :144-168)."""

    DILATIONS = (1, 2, 4, 8, 1)

    def __init__(self, dim: int):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1, dilation=1)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=2, dilation=2)
        self.conv3 = nn.Conv2d(dim, dim, kernel_size=3, padding=4, dilation=4)
        self.conv4 = nn.Conv2d(dim, dim, kernel_size=3, padding=8, dilation=8)
        self.conv5 = nn.Conv2d(dim, dim, kernel_size=3, padding=1, dilation=1)
        self.conv6 = nn.Conv2d(dim, 1, kernel_size=1, padding=0)
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6):
            nn.init.kaiming_normal_(conv.weight)
        self.dim = dim

    def forward(self, feats, attn):
        ops = _ops()
        x = feats
        for i, d in enumerate(self.DILATIONS, start=1):
            conv = getattr(self, "conv%d" % i)
            x = ops.conv3x3_relu(x, conv.weight, conv.bias, mask=attn if i == 1 else None, dilation=d)
        return ops.dot1_sigmoid(x, self.conv6.weight, self.conv6.bias)


class SameModule(nn.Module):
    """(features, attention) -> attention: correlate every location with the feature vector at
    the arg-max of the attention (first maximum; row = idx // size as under the reference's
    torch 1.4), append the attention, 1x1 conv, sigmoid (reference :194-208)."""

    def __init__(self, dim: int):
        super().__init__()
        self.conv = nn.Conv2d(dim + 1, 1, kernel_size=1)
        nn.init.kaiming_normal_(self.conv.weight)
        self.dim = dim

    def forward(self, feats, attn):
        return _ops().same(feats, attn, self.conv.weight, self.conv.bias)


class ComparisonModule(nn.Module):
    """(encoding, encoding) -> encoding: 1x1 projection of the concatenation, two 3x3 convs
    (reference :231-244; ``projection`` keeps the default Conv2d init)."""

    def __init__(self, dim: int):
        super().__init__()
        self.projection = nn.Conv2d(2 * dim, dim, kernel_size=1, padding=0)
        self.conv1 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(dim, dim, kernel_size=3, padding=1)
        nn.init.kaiming_normal_(self.conv1.weight)
        nn.init.kaiming_normal_(self.conv2.weight)

    def forward(self, in1, in2):
        ops = _ops()
        x = ops.projection_relu(in1, in2, self.projection.weight, self.projection.bias)
        x = ops.conv3x3_relu(x, self.conv1.weight, self.conv1.bias)
        return ops.conv3x3_relu(x, self.conv2.weight, self.conv2.bias)


class Flatten(nn.Module):
    """Keep the batch dimension, flatten the rest (reference :247-251)."""

    def forward(self, x):
        return x.reshape(x.size(0), -1)
