"""TFLOP/s of one grouped module-conv launch (128 -> 128, 3x3, masked, ReLU; 14x14) by number of items and
K-split, each forced through PNMN_CONV_KSPLIT in its own process ("auto": the library's launch planner).
usage: python scripts/conv_modes.py <auto|1|2|4|8>"""
import os, sys
mode = sys.argv[1]
if mode != "auto":
    os.environ["PNMN_CONV_KSPLIT"] = mode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch
from probnmn import _hip

dev = torch.device("cuda:0")
C, HW = 128, 196
torch.manual_seed(0)
NMAX = 1024
x = torch.randn(NMAX, HW, C, device=dev)
m = torch.rand(NMAX, HW, device=dev)
ws = [torch.randn(C, 9, C, device=dev) * 0.03 for _ in range(15)]
b = torch.zeros(C, device=dev)
y = torch.empty(NMAX, HW, C, device=dev)
out = []
for n in (16, 32, 64, 96, 128, 192, 256, 320, 384, 512, 768, 1024):
    rec = np.zeros(n, _hip.CONV_ITEM)
    e = np.arange(n, dtype=np.int64)
    rec["in"], rec["mask"], rec["out"] = x.data_ptr() + e * HW * C * 4, m.data_ptr() + e * HW * 4, y.data_ptr() + e * HW * C * 4
    NW = int(os.environ.get("NW", "15"))
    if os.environ.get("SORTED"):
        rec["weight"] = np.asarray([ws[i * NW // n].data_ptr() for i in range(n)], dtype=np.uint64)
    else:
        rec["weight"] = np.asarray([ws[i % NW].data_ptr() for i in range(n)], dtype=np.uint64)
    rec["bias"], rec["dilation"] = b.data_ptr(), 1
    items = _hip.to_device(rec, dev)
    st = _hip.stream_ptr(dev)
    run = lambda: _hip.check(_hip.lib().pnmn_conv_nhwc(items.data_ptr(), n, 14, 14, 1, 9, C, C, 1, 1, st), "conv")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out.append("%d:%.1f" % (n, 2.0 * n * HW * C * 9 * C / ms / 1e9))
print("%-6s" % mode, "  ".join(out), flush=True)
