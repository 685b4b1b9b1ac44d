"""ResNet-101 stage-3 feature extractor on the MI355X: images -> the (1024, 14, 14) features the NMN consumes.

The reference does this offline with torchvision (``scripts/preprocess/extract_features.py:98-105``:
``resnet101(pretrained=True)`` with ``layer4`` / ``avgpool`` / ``fc`` replaced by ``nn.Identity``, eval mode; ``:124-131`` the
forward under ``no_grad``; ``:60-73`` the 224x224 resize and the normalisation) and writes an H5 file that the training
readers load back.  :class:`ResNet101Stage3` is that network with torchvision's parameter names -- a torchvision
``resnet101`` checkpoint loads with ``load_state_dict(sd, strict=False)`` (the ``layer4.*`` / ``fc.*`` keys are the ones the
reference throws away) -- running on ``libprobnmn_hip.so``: 94 launches of ``pnmn_conv2d_nhwc`` (implicit GEMM on the fp32
matrix cores; eval-mode batch norm, residual add and ReLU in the epilogue) and one max pool, NHWC throughout.  The
result is a ``channels_last`` tensor, which is the layout ``NeuralModuleNetwork`` and the feature stores
(``probnmn.data.feature_store``) take without a layout pass -- on this part the features of all 70 000 CLEVR training
images fit in HBM (56 GB of 288), so extraction can feed a ``DeviceFeatureStore`` directly instead of a file.

No CPU fallback: parameters and images must be on a ROCm device.
"""
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from probnmn import _hip

#: torchvision.models.resnet101: Bottleneck blocks per stage (the reference drops stage 4)
LAYERS = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 23, 2))
IMAGE_SIZE = 224            # reference extract_features.py:60-61
MEAN = (0.485, 0.456, 0.406)  # reference extract_features.py:72
STD = (0.229, 0.224, 0.224)   # (sic: the reference's third std is 0.224)


class _Bottleneck(nn.Module):
    """Parameter container with torchvision's names; the arithmetic is ResNet101Stage3.forward's."""

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.stride = stride
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, 4 * planes, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(4 * planes))
        else:
            self.downsample = None


class _Conv:
    """One folded convolution: packed weight [Cout][Kpad], scale / shift on the device."""

    __slots__ = ("w", "scale", "shift", "cin", "cout", "k", "stride", "pad")


class ResNet101Stage3(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for name, planes, blocks, stride in LAYERS:
            layer = []
            for b in range(blocks):
                layer.append(_Bottleneck(inplanes, planes, stride if b == 0 else 1, downsample=(b == 0)))
                inplanes = 4 * planes
            setattr(self, name, nn.Sequential(*layer))
        self.eval()
        self._packed: Optional[Dict[str, _Conv]] = None
        self._packed_key = None

    # ---- weights ------------------------------------------------------------------------------------------------
    def _fold(self, conv: nn.Conv2d, bn: nn.BatchNorm2d) -> _Conv:
        w = conv.weight.detach()
        cout, cin, kh, kw = w.shape
        cin_p = (cin + 3) // 4 * 4  # (the image: 3 -> 4 channels)
        wk = torch.zeros(cout, kh * kw, cin_p, dtype=torch.float32, device=w.device)
        wk[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        floats = _hip.lib().pnmn_conv2d_weight_floats(cout, cin_p, kh, kw)
        if floats <= 0:
            raise _hip.HipLibraryError("pnmn_conv2d_weight_floats(%d, %d, %d, %d) = %d" % (cout, cin_p, kh, kw, floats))
        packed = torch.zeros(cout, floats // cout, dtype=torch.float32, device=w.device)
        packed[:, : kh * kw * cin_p] = wk.reshape(cout, -1)
        c = _Conv()
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        c.w, c.scale, c.shift = packed, scale.float().contiguous(), (bn.bias.detach() - bn.running_mean * scale).float().contiguous()
        c.cin, c.cout, c.k, c.stride, c.pad = cin_p, cout, kh, conv.stride[0], conv.padding[0]
        return c

    def _pack(self) -> Dict[str, _Conv]:
        # (re-folded when any parameter or buffer was written: load_state_dict, .to(), an optimizer)
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if self._packed is None or key != self._packed_key:
            packed = {"conv1": self._fold(self.conv1, self.bn1)}
            for name, _, blocks, _ in LAYERS:
                for b, blk in enumerate(getattr(self, name)):
                    p = "%s.%d" % (name, b)
                    packed[p + ".conv1"] = self._fold(blk.conv1, blk.bn1)
                    packed[p + ".conv2"] = self._fold(blk.conv2, blk.bn2)
                    packed[p + ".conv3"] = self._fold(blk.conv3, blk.bn3)
                    if blk.downsample is not None:
                        packed[p + ".downsample"] = self._fold(blk.downsample[0], blk.downsample[1])
            self._packed, self._packed_key = packed, key
        return self._packed

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("the extractor runs in eval mode only (reference extract_features.py:105)")
        return super().train(False)

    # ---- forward ------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv(c: _Conv, x: torch.Tensor, relu: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        n, h, w, cin = x.shape
        assert cin == c.cin
        ho = (h + 2 * c.pad - c.k) // c.stride + 1
        wo = (w + 2 * c.pad - c.k) // c.stride + 1
        y = torch.empty(n, ho, wo, c.cout, dtype=torch.float32, device=x.device)
        d = np.zeros(1, _hip.CONV2D_DESC)
        d[0] = (x.data_ptr(), c.w.data_ptr(), c.scale.data_ptr(), c.shift.data_ptr(),
                residual.data_ptr() if residual is not None else 0, y.data_ptr(),
                n, h, w, cin, ho, wo, c.cout, c.k, c.k, c.stride, c.pad, int(relu))
        _hip.check(_hip.lib().pnmn_conv2d_nhwc(d.ctypes.data, _hip.stream_ptr(x.device)), "conv2d_nhwc")
        return y

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """``images``: (N, 3, H, W) float, normalised as the reference's transform does (:func:`preprocess`); H and W
        multiples of 32.  Returns (N, 1024, H / 16, W / 16) fp32 in ``channels_last`` memory format."""
        dev = self.conv1.weight.device
        if dev.type != "cuda" or images.device != dev:
            raise _hip.HipLibraryError("ResNet101Stage3 runs on a ROCm device only (parameters on %s, images on %s): "
                                       "there is no CPU fallback" % (dev, images.device))
        if images.dim() != 4 or images.size(1) != 3 or images.size(2) % 32 or images.size(3) % 32:
            raise ValueError("expected images (N, 3, H, W) with H and W multiples of 32, got %s" % (tuple(images.shape),))
        packed = self._pack()
        n, _, h, w = images.shape
        x = torch.zeros(n, h, w, 4, dtype=torch.float32, device=dev)
        x[..., :3] = images.float().permute(0, 2, 3, 1)
        x = self._conv(packed["conv1"], x, relu=True)
        pooled = torch.empty(n, x.size(1) // 2, x.size(2) // 2, 64, dtype=torch.float32, device=dev)
        _hip.check(_hip.lib().pnmn_maxpool3x3s2_nhwc(x.data_ptr(), pooled.data_ptr(), n, x.size(1), x.size(2), 64,
                                                     _hip.stream_ptr(dev)), "maxpool3x3s2")
        x = pooled
        for name, _, blocks, _ in LAYERS:
            for b in range(blocks):
                p = "%s.%d" % (name, b)
                identity = x
                if p + ".downsample" in packed:
                    identity = self._conv(packed[p + ".downsample"], x, relu=False)
                out = self._conv(packed[p + ".conv1"], x, relu=True)
                out = self._conv(packed[p + ".conv2"], out, relu=True)
                x = self._conv(packed[p + ".conv3"], out, relu=True, residual=identity)
        return x.permute(0, 3, 1, 2)  # (N, 1024, h/16, w/16) over NHWC storage

    def flops_per_image(self, h: int = IMAGE_SIZE, w: int = IMAGE_SIZE) -> float:
        """Algorithmic FLOPs of one image (2 x MACs of the 94 convolutions; the image counted with its 3 channels)."""
        total, size, inplanes = 2.0 * (h // 2) * (w // 2) * 64 * 49 * 3, (h // 4, w // 4), 64
        for _, planes, blocks, stride in LAYERS:
            for b in range(blocks):
                s = stride if b == 0 else 1
                out_size = (size[0] // s, size[1] // s)
                total += 2.0 * size[0] * size[1] * inplanes * planes
                total += 2.0 * out_size[0] * out_size[1] * planes * planes * 9
                total += 2.0 * out_size[0] * out_size[1] * planes * 4 * planes
                if b == 0:
                    total += 2.0 * out_size[0] * out_size[1] * inplanes * 4 * planes
                size, inplanes = out_size, 4 * planes
        return total


def preprocess(images_uint8: torch.Tensor) -> torch.Tensor:
    """``ToTensor`` + ``Normalize`` of the reference's transform (extract_features.py:70-73) on (N, 3, H, W) uint8 images,
    on whatever device they are; ``Resize((224, 224))`` (PIL, bilinear) stays with the caller that decodes the PNGs."""
    x = images_uint8.float() / 255.0
    mean = torch.tensor(MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def extract_features(model: ResNet101Stage3, image_batches: Iterable[torch.Tensor], out: torch.Tensor) -> int:
    """The reference's extraction loop (extract_features.py:124-131) into ``out`` -- (n_images, 1024, 14, 14), any device, any
    memory format: a ``channels_last`` tensor in HBM is what ``DeviceFeatureStore`` / the NMN read in place, a host tensor
    is the H5 file's content.  ``image_batches`` yields normalised (b, 3, 224, 224) batches in image order.  Returns the
    number of images written."""
    dev = model.conv1.weight.device
    counter = 0
    for batch in image_batches:
        feats = model(batch.to(dev, non_blocking=True))
        out[counter: counter + feats.size(0)].copy_(feats)
        counter += feats.size(0)
    return counter
