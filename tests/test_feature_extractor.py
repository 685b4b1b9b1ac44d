"""The ResNet-101 stage-3 feature extractor (probnmn.data.feature_extractor, csrc/resnet.hip) against its oracle
(oracle/resnet_oracle.py: torchvision 0.5.0's definition restated; parity unpinned -- torchvision is absent) and against
torch's own convolution on the device.  Reference: scripts/preprocess/extract_features.py:98-105, 124-131."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from probnmn import _hip
from probnmn.data.feature_extractor import LAYERS, ResNet101Stage3, extract_features, preprocess


def _randomise(model, seed):
    """He-initialised convolutions and batch-norm statistics of a trained network's order of magnitude (all four tensors
    of every batch norm matter: an extractor that ignored running_var would pass with the defaults)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in model.state_dict().items():
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            elif name.endswith("running_mean") or name.endswith(".bias"):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif t.dim() == 4:
                fan_in = t.size(1) * t.size(2) * t.size(3)
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            else:  # batch-norm weight; small on a block's last norm keeps 30 residual blocks from blowing up
                t.copy_((0.3 if "bn3" in name else 1.0) * (0.75 + 0.5 * torch.rand(t.shape, generator=g)))


def test_parameter_names_and_count_are_torchvisions():
    """A torchvision resnet101 checkpoint must load: same key names and shapes.  Known answers from torchvision's
    published model: 44 549 160 parameters in all, 14 964 736 of them in layer4 and 2 049 000 in fc -- the parts the
    reference replaces by Identity (extract_features.py:101-103)."""
    m = ResNet101Stage3()
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 44549160 - 14964736 - 2049000
    assert sd["conv1.weight"].shape == (64, 3, 7, 7) and sd["bn1.running_var"].shape == (64,)
    assert sd["layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1) and "layer1.1.downsample.0.weight" not in sd
    assert sd["layer2.0.conv2.weight"].shape == (128, 128, 3, 3) and sd["layer3.22.conv3.weight"].shape == (1024, 256, 1, 1)
    assert sd["layer3.0.downsample.1.num_batches_tracked"].shape == ()
    assert [len(getattr(m, name)) for name, _, _, _ in LAYERS] == [3, 4, 23]
    assert not any(k.startswith(("layer4", "fc")) for k in sd)
    assert m.flops_per_image() == pytest.approx(13.98e9, rel=1e-3)
    with pytest.raises(NotImplementedError):
        m.train()


def test_oracle_shapes_and_preprocess():
    from oracle import resnet_oracle

    m = ResNet101Stage3()
    _randomise(m, 0)
    x = torch.randint(0, 256, (2, 3, 64, 96), dtype=torch.uint8)
    a, b = preprocess(x), resnet_oracle.preprocess(x)
    assert torch.equal(a, b)
    assert float(a[0, 2, 0, 0]) == pytest.approx((float(x[0, 2, 0, 0]) / 255.0 - 0.406) / 0.224, abs=1e-6)
    y = resnet_oracle.resnet101_stage3(m.state_dict(), a)
    assert y.shape == (2, 1024, 4, 6) and float(y.min()) >= 0.0 and torch.isfinite(y).all()


def test_cpu_fails_loudly():
    m = ResNet101Stage3()
    with pytest.raises(_hip.HipLibraryError):
        m(torch.zeros(1, 3, 224, 224))


CASES = [  # (N, H, W, Cin, Cout, k, stride, pad, relu, residual)
    (2, 64, 64, 4, 64, 7, 2, 3, True, False),      # the stem's 7x7 / 2 on the padded image
    (3, 14, 14, 256, 256, 3, 1, 1, True, False),   # M = 588: a partial last tile
    (2, 28, 28, 128, 128, 3, 2, 1, True, False),   # a stage's first 3x3 carries the stride
    (2, 28, 28, 256, 512, 1, 2, 0, False, False),  # downsample
    (2, 14, 14, 256, 1024, 1, 1, 0, True, True),   # a block's last 1x1: + identity, ReLU
    (1, 56, 56, 64, 64, 1, 1, 0, True, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_conv2d_nhwc_matches_torch(case):
    n, h, w, cin, cout, k, stride, pad, relu, residual = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale, shift = 0.5 + torch.rand(cout, generator=g), 0.1 * torch.randn(cout, generator=g)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if residual else None
    ref = F.conv2d(x.double(), wt.double(), stride=stride, padding=pad) * scale.view(1, -1, 1, 1).double() + shift.view(1, -1, 1, 1).double()
    if residual:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    lib = _hip.lib()
    floats = lib.pnmn_conv2d_weight_floats(cout, cin, k, k)
    packed = torch.zeros(cout, floats // cout)
    packed[:, : k * k * cin] = wt.permute(0, 2, 3, 1).reshape(cout, -1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    pd, sd, hd = packed.to(dev), scale.to(dev), shift.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if residual else None
    y = torch.full((n, ho, wo, cout), float("nan"), device=dev)
    d = np.zeros(1, _hip.CONV2D_DESC)
    d[0] = (xd.data_ptr(), pd.data_ptr(), sd.data_ptr(), hd.data_ptr(), rd.data_ptr() if residual else 0, y.data_ptr(),
            n, h, w, cin, ho, wo, cout, k, k, stride, pad, int(relu))
    _hip.check(lib.pnmn_conv2d_nhwc(d.ctypes.data, _hip.stream_ptr(dev)), "conv2d")
    torch.cuda.synchronize()
    torch.testing.assert_close(y.permute(0, 3, 1, 2).cpu().double(), ref, rtol=2e-5, atol=2e-5)
    # wrong output size / channel counts are refused, not launched
    d[0]["Ho"] += 1
    assert lib.pnmn_conv2d_nhwc(d.ctypes.data, _hip.stream_ptr(dev)) == _hip.ESHAPE
    d[0]["Ho"] -= 1
    d[0]["Cout"] = cout + 32
    assert lib.pnmn_conv2d_nhwc(d.ctypes.data, _hip.stream_ptr(dev)) == _hip.ESHAPE


@pytest.mark.gpu
def test_maxpool3x3s2_matches_torch():
    dev = torch.device("cuda:0")
    x = torch.randn(3, 64, 112, 112, generator=torch.Generator().manual_seed(4))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    y = torch.empty(3, 56, 56, 64, device=dev)
    _hip.check(_hip.lib().pnmn_maxpool3x3s2_nhwc(xd.data_ptr(), y.data_ptr(), 3, 112, 112, 64, _hip.stream_ptr(dev)), "maxpool")
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 3, 2, 1))


@pytest.mark.gpu
def test_extractor_matches_oracle_and_feeds_the_network():
    """Two 224x224 images through all 94 convolutions against the CPU oracle (fp32 both sides, different summation orders:
    2e-4 of the largest feature, measured ~2e-6), in the layout the NMN takes in place."""
    from oracle import resnet_oracle

    dev = torch.device("cuda:0")
    m = ResNet101Stage3()
    _randomise(m, 1)
    cpu_sd = {k: v.clone() for k, v in m.state_dict().items()}
    images = preprocess(torch.randint(0, 256, (2, 3, 224, 224), dtype=torch.uint8, generator=torch.Generator().manual_seed(2)))
    ref = resnet_oracle.resnet101_stage3(cpu_sd, images)
    m.to(dev)
    got = m(images.to(dev))
    assert got.shape == (2, 1024, 14, 14) and got.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    assert scale > 0.1 and float((ref > 0).float().mean()) > 0.1  # (a dead network would compare equal trivially)
    assert float((got.cpu() - ref).abs().max()) <= 2e-4 * scale
    # a changed statistic is seen (the folded weights are rebuilt), and the loop writes what forward returns
    with torch.no_grad():
        m.layer3[22].bn3.running_var.mul_(4.0)
    again = m(images.to(dev))
    assert float((again - got).abs().max()) > 1e-3 * scale
    out = torch.empty(2, 1024, 14, 14, device=dev).contiguous(memory_format=torch.channels_last)
    assert extract_features(m, [images[:1], images[1:]], out) == 2
    assert torch.equal(out, again)
