"""Which held-out examples does the device get wrong after a 100-iteration module-training trajectory that the oracle and its
controls get right?  (tests/test_parity_hygiene_gpu.py::test_long_module_training_trajectory found 120-123 / 128 against
126-128.)  Prints per wrong example: program, answer, the device's and the oracle's prediction and logit margin; and the Adam
step counts of the device's modules against the number of iterations in which each module appeared."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import test_trajectory_gpu as tj
from oracle import nmn_oracle
from oracle.train_oracle import OracleModuleTrainer
from probnmn.models.nmn import NeuralModuleNetwork
from probnmn.trainers.module_training import ModuleTrainingStep
from probnmn.vocabulary import Vocabulary

torch.set_num_threads(16)
dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
itos = vocab.get_index_to_token_vocabulary("programs")
torch.manual_seed(0)
net = NeuralModuleNetwork(vocab)
cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net.to(dev)
lr, iters, B = 3e-4, int(sys.argv[1]) if len(sys.argv) > 1 else 100, 32
trainer = ModuleTrainingStep(net, lr=lr)
ref = OracleModuleTrainer(cpu_sd, itos, lr=lr)
seen = {}
for it in range(iters):
    batch = tj.learnable_batch(vocab, B, seed=5000 + it)
    for row in batch["program"].tolist():
        for t in set(row):
            seen.setdefault(itos[t], set()).add(it)
    g = float(trainer.step(tj.to_dev(batch, dev))["loss"])
    w = float(ref.step(batch)["loss"])
    if it % 10 == 0:
        print("%3d %.4f %.4f" % (it, g, w), flush=True)
held = tj.learnable_batch(vocab, 128, seed=99)
net.eval()
with torch.no_grad():
    d = tj.to_dev(held, dev)
    out = net(d["image"], d["program"], d["answer"])
    pred = out["predictions"].cpu()
    ro = nmn_oracle.nmn_forward(ref.params, itos, held["image"], held["program"], held["answer"])
    # the ORACLE's forward on the DEVICE's weights: is it the weights or the forward pass?
    dev_sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    cross = nmn_oracle.nmn_forward(dev_sd, itos, held["image"], held["program"], held["answer"])
print("device correct %d, oracle %d, oracle forward on the device's weights %d" % (int((pred == held["answer"]).sum()), int((ro["predictions"] == held["answer"]).sum()),
                                                                                   int((cross["predictions"] == held["answer"]).sum())))
print("device forward vs oracle forward on the device's weights: predictions equal %d / 128, loss diff %.2e"
      % (int((pred == cross["predictions"]).sum()), float((out["loss"].cpu() - cross["loss"]).abs().max())))
wrong = (pred != held["answer"]).nonzero().flatten().tolist()
for i in wrong:
    toks = [itos[t] for t in held["program"][i].tolist() if t != 0]
    print("example %3d answer %2d device %2d oracle %2d  program %s" % (i, int(held["answer"][i]), int(pred[i]), int(ro["predictions"][i]), " ".join(toks)))
# Adam step counts of the device's trunk parameters against how often their module appeared
opt = trainer.optimizer
arena = opt.arenas[0]
steps = opt._arena_steps[0]
print("module: iterations in which it appeared / Adam steps of its first parameter (the reference's Adam: first appearance .. end)")
done = set()
for i, name in enumerate(arena.names):
    mod = name.split(".")[0]
    if mod in done or mod in ("stem", "classifier"):
        continue
    done.add(mod)
    its = sorted(seen.get(mod, []))
    want = iters - its[0] if its else 0
    flag = "" if int(steps[i]) == want else "   <-- differs"
    print("  %-28s appeared %3d times, first at %3s: Adam steps %3d, reference %3d%s" % (mod, len(its), its[0] if its else "-", int(steps[i]), want, flag))
