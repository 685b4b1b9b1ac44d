"""WRITE_SIZE of 1x1 convolutions with different output strides / block counts (run under rocprofv3 --pmc WRITE_SIZE):
why does the classifier conv (128 -> 512, pixel stride 2 KB) write twice its output?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch
from probnmn import _hip
dev = torch.device("cuda:0")
lib = _hip.lib()
n, HW, C = 1024, 196, 128
x = torch.randn(n, HW, C, device=dev)
def run(cout, stride, tag):
    w = torch.randn(cout, 1, C, device=dev) * 0.05
    out = torch.zeros(n, HW, stride, device=dev)
    recs = np.zeros(n, _hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["weight"], recs[i]["out"] = x[i].data_ptr(), w.data_ptr(), out[i].data_ptr()
    d = _hip.to_device(recs, dev)
    for _ in range(2):
        _hip.check(lib.pnmn_conv_nhwc(d.data_ptr(), n, 14, 14, 1, 1, C, stride, cout // C, 1, _hip.stream_ptr(dev)), tag)
    torch.cuda.synchronize()
    print(tag, "algorithmic output MB: %.1f" % (n * HW * cout * 4 / 1e6))
run(512, 512, "A cout 512 stride 512")
run(128, 128, "B cout 128 stride 128")
run(128, 512, "C cout 128 stride 512")
run(256, 256, "D cout 256 stride 256")
