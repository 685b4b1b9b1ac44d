"""Debug aid: ones input, weights select channel c0 of every tap -> output = number of taps inside the image."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch
from probnmn import _hip
dev = torch.device("cuda:0")
C, H, W = 128, 14, 14
HW = H * W
n = 129
for c0 in (0, 5, 17, 40):
    x = torch.zeros(n, HW, C, device=dev)
    x[:, :, c0] = torch.arange(1, HW + 1, device=dev, dtype=torch.float32)[None, :]   # pixel p carries p+1 in channel c0
    w = torch.zeros(C, 9, C, device=dev); w[:, :, c0] = 1.0
    b = torch.zeros(C, device=dev)
    out = torch.full((n, HW, C), float("nan"), device=dev)
    rec = np.zeros(n, _hip.CONV_ITEM)
    for i in range(n):
        rec[i]["in"], rec[i]["weight"], rec[i]["bias"], rec[i]["out"] = x[i].data_ptr(), w.data_ptr(), b.data_ptr(), out[i].data_ptr()
        rec[i]["dilation"] = 1
    items = _hip.to_device(rec, dev)
    _hip.check(_hip.lib().pnmn_conv_nhwc(items.data_ptr(), n, H, W, 1, 9, C, C, 1, 0, _hip.stream_ptr(dev)), "conv")
    torch.cuda.synchronize()
    good = out[0, :, 0].reshape(H, W)
    bad = out[128, :, 0].reshape(H, W)
    d = (bad - good)
    print("c0=%d: item 128 - item 0 (channel 0), nonzero entries:" % c0)
    nz = d.nonzero()
    for (y, xx) in nz[:12].tolist():
        print("   pixel (%d,%d): diff %.1f (good %.1f)" % (y, xx, float(d[y, xx]), float(good[y, xx])))
    print("   total nonzero", int((d != 0).sum()), " equal across out channels:", bool((out[128] == out[128][:, :1]).all()))
