"""Diagnostic: N identical examples in one batch must give identical activations / logits."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np, torch
from fixtures import encode_programs
from oracle import nmn_oracle
from probnmn.models.nmn import NeuralModuleNetwork
from probnmn.vocabulary import Vocabulary

size = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
vocab = Vocabulary.clevr()
torch.manual_seed(21)
net = NeuralModuleNetwork(vocab, image_feature_size=(1024, size, size))
cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
dev = torch.device("cuda:0")
net.to(dev).train()
case = "count filter_shape[sphere] scene"
programs = encode_programs([case] * n, vocab.get_token_to_index_vocabulary("programs"))
itos = vocab.get_index_to_token_vocabulary("programs")
g = torch.Generator().manual_seed(3)
f1 = torch.relu(torch.randn(1, 1024, size, size, generator=g))
features = f1.repeat(n, 1, 1, 1)
answers = torch.full((n,), 3, dtype=torch.long)
out = net(features.to(dev), programs.to(dev), answers.to(dev))
out["loss"].mean().backward()
eng = net.engine
HW = size * size
for name, ch in (("xin", 1024), ("stem1", 128), ("feat", 128), ("final", 128), ("cls", 1024), ("gcls", 1024), ("gfinal", 128), ("gfeat", 128), ("gstem1", 128)):
    t = eng._ws[name][: n * HW * ch].view(n, HW, ch)
    d = (t - t[0:1]).abs().amax(dim=(1, 2))
    print("%-8s per-example max |x_i - x_0|:" % name, d.tolist(), " scale", float(t[0].abs().max()))
    if float(d.max()) > 0:
        i = int(d.argmax())
        dd = (t[i] - t[0]).abs()
        bad = (dd > 0).nonzero()
        print("   example", i, "n_bad", bad.shape[0], "pixels", sorted(set(bad[:, 0].tolist()))[:40])
# oracle on one example
sd = {k: v.clone() for k, v in cpu_sd.items()}
ref = nmn_oracle.nmn_forward(sd, itos, f1, programs[:1], answers[:1])
print("logits gpu", out["loss"].tolist(), "oracle", ref["loss"].tolist())
