"""PCIe-inclusive ingest rate of the feature store (questions/s) and its overlap with a training step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch
from probnmn.data.feature_store import PinnedFeatureStore

dev = torch.device("cuda:0")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 14
N = 4096 if size == 14 else 1024
rng = np.random.Generator(np.random.Philox(0))
feats = np.empty((N, 1024, size, size), np.float32)
for lo in range(0, N, 256):
    feats[lo:lo + 256] = rng.standard_normal((min(256, N - lo), 1024, size, size), dtype=np.float32)
t0 = time.perf_counter(); store = PinnedFeatureStore(feats); t1 = time.perf_counter()
print("store: %d x 1024 x %d x %d fp32 = %.2f GB pinned in %.1f s" % (N, size, size, feats.nbytes / 1e9, t1 - t0))
for B in (128, 1024):
    idx = torch.from_numpy(rng.integers(0, N, B)).to(dev)
    out = store.gather(idx, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        store.gather(idx, dev, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = B * 1024 * size * size * 4
    print("gather %4d questions (%dx%d): %.3f ms = %.1f GB/s over PCIe = %.0f questions/s" % (B, size, size, ms, nbytes / ms / 1e6, B / ms * 1e3))
