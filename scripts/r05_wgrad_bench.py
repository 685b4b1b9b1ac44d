"""3x3 weight-gradient launches on an idle chip: the shapes of a 1024-question joint step (module convs: ~6 000 items
over ~15 weights in jobs of 8; stem conv2: 512 items, one weight; stem conv1: 512 items, 1024 input channels) through
pnmn_conv_wgrad, checked against torch on a small case first.  Prints ms and TFLOP/s per shape.

    python scripts/r05_wgrad_bench.py [--cycles]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np
import torch
import torch.nn.functional as F

from probnmn import _hip as hip

H = W = 14
HW = H * W
C = 128
dev = torch.device("cuda:0")


def ptr(t, off=0):
    return t.data_ptr() + 4 * off


def check(dilation, n=5, masked=True, cin=C):
    g = torch.Generator().manual_seed(7 + dilation)
    x = torch.relu(torch.randn(n, cin, H, W, generator=g))
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g)) if masked else torch.ones(n, 1, H, W)
    w = (torch.randn(C, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    y = F.relu(F.conv2d(x * m, w, b, padding=dilation, dilation=dilation))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(t.size(0), -1, t.size(1)).contiguous().to(dev)
    xd, md, dyd, yd = nhwc(x), m.reshape(n, HW).to(dev), nhwc(dy), nhwc(y.detach())
    items = np.zeros(n, hip.WGRAD_ITEM)
    for i in range(n):
        items[i]["x"], items[i]["dy"], items[i]["gate"] = ptr(xd[i]), ptr(dyd[i]), ptr(yd[i])
        if masked:
            items[i]["xmask"] = ptr(md[i])
        items[i]["dilation"] = dilation
    dw = torch.zeros(C, 9, cin, device=dev)
    db = torch.zeros(C, device=dev)
    jobs = np.zeros(2, hip.WGRAD_JOB)
    jobs["dw"], jobs["dbias"] = ptr(dw), ptr(db)
    jobs[0]["item_begin"], jobs[0]["item_end"] = 0, 3
    jobs[1]["item_begin"], jobs[1]["item_end"] = 3, n
    ib, jb = hip.to_device(items, dev), hip.to_device(jobs, dev)
    hip.check(hip.lib().pnmn_conv_wgrad(ib.data_ptr(), jb.data_ptr(), 2, H, W, 9, cin // C, 1, cin, C, hip.stream_ptr(dev)), "wgrad")
    torch.cuda.synchronize()
    ref = w.grad.permute(0, 2, 3, 1).reshape(C, 9, cin)
    e1 = (dw.cpu() - ref).abs().max().item() / ref.abs().max().item()
    e2 = (db.cpu() - b.grad).abs().max().item() / b.grad.abs().max().item()
    print("check dilation %d cin %4d masked %d: dW rel err %.2e, dbias rel err %.2e" % (dilation, cin, masked, e1, e2), flush=True)
    assert e1 < 1e-4 and e2 < 1e-4


def bench(name, n_items, n_weights, chunk, cin=C, gated=True, masked=False, dil=(1,), reps=5):
    x = torch.randn(n_items, HW, cin, device=dev)
    dy = torch.randn(n_items, HW, C, device=dev)
    gate = torch.randn(n_items, HW, C, device=dev)
    mask = torch.rand(n_items, HW, device=dev)
    dw = torch.zeros(n_weights, C, 9, cin, device=dev)
    db = torch.zeros(n_weights, C, device=dev)
    items = np.zeros(n_items, hip.WGRAD_ITEM)
    per = (n_items + n_weights - 1) // n_weights
    jl = []
    for i in range(n_items):
        items[i]["x"], items[i]["dy"] = ptr(x[i]), ptr(dy[i])
        if gated:
            items[i]["gate"] = ptr(gate[i])
        if masked:
            items[i]["xmask"] = ptr(mask[i])
        items[i]["dilation"] = dil[(i // per) % len(dil)]
    for wi in range(n_weights):
        lo, hi = wi * per, min((wi + 1) * per, n_items)
        for s in range(lo, hi, chunk):
            jl.append((ptr(dw[wi]), ptr(db[wi]), s, min(s + chunk, hi)))
    jobs = np.zeros(len(jl), hip.WGRAD_JOB)
    for k, (a, b, s, e) in enumerate(jl):
        jobs[k]["dw"], jobs[k]["dbias"], jobs[k]["item_begin"], jobs[k]["item_end"] = a, b, s, e
    ib, jb = hip.to_device(items, dev), hip.to_device(jobs, dev)
    sp = hip.stream_ptr(dev)
    fn = lambda: hip.check(hip.lib().pnmn_conv_wgrad(ib.data_ptr(), jb.data_ptr(), len(jl), H, W, 9, cin // C, 1, cin, C, sp), "wgrad")
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    flops = 2.0 * n_items * HW * C * 9 * cin
    print("%-34s %5d items %3d weights chunk %2d (%4d jobs): %.3f ms  %.1f TFLOP/s" % (name, n_items, n_weights, chunk, len(jl), best, flops / best / 1e9), flush=True)


if __name__ == "__main__":
    hip.lib()
    for d in (1, 2, 4, 8):
        check(d)
    check(1, n=7, masked=False)
    check(1, n=4, cin=256)
    bench("module wgrad (1024 q)", 6180, 15, 8, masked=True, dil=(1, 1, 1, 2, 4, 8, 1))
    bench("module wgrad, no mask", 6180, 15, 8)
    bench("module wgrad, dilation 8", 6180, 15, 8, dil=(8,))
    bench("module wgrad chunk 16", 6180, 15, 16)
    bench("module wgrad chunk 4", 6180, 15, 4)
    bench("module wgrad (128 q)", 780, 15, 3)
    bench("stem conv2 wgrad (512)", 512, 1, 8)
    bench("stem conv2 wgrad (512) chunk 4", 512, 1, 4)
    bench("stem conv2 wgrad (256)", 256, 1, 4)
    bench("stem conv1 wgrad (512)", 512, 1, 8, cin=1024)
    bench("stem conv1 wgrad (512) chunk 16", 512, 1, 16, cin=1024)
