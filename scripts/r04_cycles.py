"""Cycle accounting of the streamed conv kernel (PNMN_CONV_DBGPTR): per contraction wave, cycles at barriers, in epilogues,
in units and in the kernel.   usage: python scripts/r04_cycles.py [n_items] [masked]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch
dev = torch.device("cuda:0")
dbg = torch.zeros(256 * 5 * 8, dtype=torch.int64, device=dev)
os.environ["PNMN_CONV_DBGPTR"] = str(dbg.data_ptr())
os.environ.setdefault("PNMN_LIB", os.path.join(ROOT, "probnmn-clevr_amd", "lib", "libprobnmn_cycles.so"))  # (make -C probnmn-clevr_amd/csrc cycles)
from probnmn import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
masked = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 14
gated = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dil = int(sys.argv[5]) if len(sys.argv) > 5 else 1
C, HW = 128, S * S
x = torch.randn(n, HW, C, device=dev); m = torch.rand(n, HW, device=dev)
ws = [torch.randn(C, 9, C, device=dev) * 0.03 for _ in range(15)]
b = torch.zeros(C, device=dev); y = torch.empty(n, HW, C, device=dev)
rec = np.zeros(n, _hip.CONV_ITEM); e = np.arange(n, dtype=np.int64)
rec["in"], rec["out"] = x.data_ptr() + e * HW * C * 4, y.data_ptr() + e * HW * C * 4
if masked: rec["mask"] = m.data_ptr() + e * HW * 4
gt = torch.randn(n, HW, C, device=dev)
if gated: rec["gate"] = gt.data_ptr() + e * HW * C * 4
rec["weight"] = np.asarray([ws[i * 15 // n].data_ptr() for i in range(n)], dtype=np.uint64)
rec["bias"], rec["dilation"] = b.data_ptr(), dil
items = _hip.to_device(rec, dev); st = _hip.stream_ptr(dev)
run = lambda: _hip.check(_hip.lib().pnmn_conv_nhwc(items.data_ptr(), n, S, S, 1, 9, C, C, 1, 1, st), "conv")
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = dbg.cpu().numpy().astype(np.float64)
d = raw[: 256 * 4 * 8].reshape(256, 4, 8)
ld = raw[256 * 4 * 8:].reshape(256, 8)
units = d[:, :, 3]
print("n=%d masked=%d %dx%d gated=%d dil=%d: %.1f us, %.1f TFLOP/s" % (n, masked, S, S, gated, dil, ms * 1e3, 2.0 * n * HW * C * 9 * C / ms / 1e9))
print("per wave: kernel cycles mean %.0f max %.0f  (%.0f cycles/us if the longest wave spans the launch)" % (d[:, :, 4].mean(), d[:, :, 4].max(), d[:, :, 4].max() / (ms * 1e3)))
print("units per workgroup: mean %.2f max %.0f" % (units[:, 0].mean(), units.max()))
tot = d[:, :, 4].sum()
print("share of wave time: barriers %.1f %%  epilogue+next-unit decode %.1f %%  inside units %.1f %%" % (100 * d[:, :, 0].sum() / tot, 100 * d[:, :, 1].sum() / tot, 100 * d[:, :, 2].sum() / tot))
per_unit = d[:, :, 2].sum() / units.sum()
print("cycles per unit %.0f (3744 MFMAs x 32 = 119808 ideal): barrier %.0f, epilogue %.0f" % (per_unit, d[:, :, 0].sum() / units.sum(), d[:, :, 1].sum() / units.sum()))
for w in range(4):
    print("  wave %d: unit %.0f barrier %.0f" % (w, d[:, w, 2].sum() / units[:, w].sum(), d[:, w, 0].sum() / units[:, w].sum()))
print("loader per unit: issue %.0f  wait-landed %.0f  prologue+table %.0f  barriers %.0f" % tuple(ld[:, k].sum() / units[:, 0].sum() for k in range(4)))
