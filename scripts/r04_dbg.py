"""Debug aid (round 4): streamed conv kernel vs torch conv2d on the device; prints WHERE results differ.
usage: python scripts/r04_dbg.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch, torch.nn.functional as F
from probnmn import _hip
dev = torch.device("cuda:0")
C, H, W = 128, 14, 14
HW = H * W

def case(n, split, dils, masked, nw=4):
    _hip.lib().pnmn_conv_force_split(split)
    g = torch.Generator().manual_seed(n)
    x = torch.relu(torch.randn(n, C, H, W, generator=g)).to(dev)
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g)).to(dev)
    w = (torch.randn(nw, C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev)
    b = (torch.randn(nw, C, generator=g) * 0.1).to(dev)
    xd = x.permute(0, 2, 3, 1).reshape(n, HW, C).contiguous()
    md = m.reshape(n, HW).contiguous()
    wd = [w[k].permute(0, 2, 3, 1).reshape(C, 9, C).contiguous() for k in range(nw)]
    out = torch.full((n, HW, C), float("nan"), device=dev)
    rec = np.zeros(n, _hip.CONV_ITEM)
    for i in range(n):
        rec[i]["in"] = xd[i].data_ptr()
        rec[i]["mask"] = md[i].data_ptr() if masked else 0
        rec[i]["weight"], rec[i]["bias"], rec[i]["out"] = wd[i % nw].data_ptr(), b[i % nw].data_ptr(), out[i].data_ptr()
        rec[i]["dilation"] = dils[i % len(dils)]
    items = _hip.to_device(rec, dev)
    _hip.check(_hip.lib().pnmn_conv_nhwc(items.data_ptr(), n, H, W, 1, 9, C, C, 1, 1, _hip.stream_ptr(dev)), "conv")
    torch.cuda.synchronize()
    got = out.reshape(n, H, W, C).permute(0, 3, 1, 2)
    xin = x * m if masked else x
    bad_total = 0
    for i in range(n):
        d = dils[i % len(dils)]
        ref = F.relu(F.conv2d(xin[i:i + 1], w[i % nw], b[i % nw], padding=d, dilation=d))[0]
        bad = ~torch.isclose(got[i], ref, rtol=2e-4, atol=2e-4)
        if bad.any():
            bad_total += 1
            if bad_total <= 4:
                idx = bad.nonzero()
                ch = sorted(set(idx[:, 0].tolist())); px = sorted(set((idx[:, 1] * W + idx[:, 2]).tolist()))
                print("   item %d dil %d: %d bad; channels %s..%s (%d distinct) pixels %s" % (i, d, int(bad.sum()), ch[:3], ch[-3:], len(ch), px[:24]))
    print("n=%d split=%d dils=%s masked=%d -> %d bad items" % (n, split, dils, masked, bad_total), flush=True)

for n, split, dils, masked in [(8, 1, [1], 0), (8, 1, [1], 1), (8, 1, [2], 0), (8, 1, [8], 0), (8, 2, [1, 2], 1), (8, 4, [1, 2, 4, 8], 1),
                               (129, 0, [1], 0), (129, 0, [1, 2], 1), (300, 0, [1, 2], 1), (600, 1, [1, 2, 4, 8], 1)]:
    case(n, split, dils, masked)
_hip.lib().pnmn_conv_force_split(0)
