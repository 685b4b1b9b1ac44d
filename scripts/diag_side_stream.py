"""Loop the 1024-row side-stream vs single-stream comparison of tests/test_full_size_gpu.py until the gradients
disagree, and print where: per layer errors, the classifier's weights before the step, the first FC layer's
pre-activations and which ReLU gates differ.  (Round 3: the 1-in-8 disagreement at the third step is ONE gate --
row 297, unit 72, pre-activation +1.3e-8 / -7.9e-9 -- after Adam moved the weights apart by ~1e-9.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import probnmn.models.nmn as nmn_mod
from probnmn.data.synthetic import synthetic_batch
from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import JointTrainingStep
from probnmn.vocabulary import Vocabulary

DEV = torch.device("cuda:0")
vocab = Vocabulary.clevr()
batch = synthetic_batch(vocab, 1024, seed=5)
sup = batch["supervision"]
batch = {k: v.to(DEV) for k, v in batch.items()}
batch["supervision"] = sup
STEPS = int(os.environ.get("DIAG_STEPS", "3"))


PRE = []
_orig_fc = nmn_mod._first_fc


def _recording_fc(layer, x):
    y = _orig_fc(layer, x)
    PRE.append((y.detach().clone(), x.detach().clone()))
    return y


nmn_mod._first_fc = _recording_fc


def run(use_side):
    PRE.clear()
    torch.manual_seed(0)
    nmn = NeuralModuleNetwork(vocab).to(DEV)
    pg, qr = ProgramGenerator(vocab).to(DEV), QuestionReconstructor(vocab).to(DEV)
    prior = ProgramPrior(vocab, hidden_size=256).to(DEV)
    step = JointTrainingStep(pg, qr, prior, nmn, objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=1e-6)
    step.nmn_stream, step.nmn_stream_max_rows = use_side, 1 << 30
    step.shared_conv_cus = 256
    torch.manual_seed(1)
    outs = []
    for _ in range(STEPS):
        w_before = {n: p.detach().clone() for n, p in nmn.named_parameters() if n.startswith("classifier")}
        out = step.step(batch)
        torch.cuda.synchronize()
        outs.append((out["programs"].clone(), float(out["objective"]), float(out["loss"]["nmn"]),
                     {n: p.grad.detach().clone() for n, p in nmn.named_parameters() if p.grad is not None},
                     w_before, PRE[-1]))
    step.close()
    return outs


for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    a, b = run(True), run(False)
    bad = False
    for k, ((pa, oa, na, ga, wa, (ha, xa)), (pb, ob, nb, gb, wb, (hb, xb))) in enumerate(zip(a, b)):
        same_prog = bool(torch.equal(pa, pb))
        worst = []
        for n in ga:
            scale = float(gb[n].abs().max()) + 1e-12
            err = ((ga[n] - gb[n]).abs().reshape(-1) / scale)
            worst.append((float(err.median()), float(err.max()), n))
        worst.sort(reverse=True)
        flag = worst[0][0] > 1e-4
        print("trial %d step %d: programs equal %s, objective %.6f / %.6f, nmn loss %.6f / %.6f, worst median err %.2e (%s)%s" % (
            trial, k, same_prog, oa, ob, na, nb, worst[0][0], worst[0][2], "   <-- MISMATCH" if flag else ""), flush=True)
        if flag:
            bad = True
            for n in ("classifier.6.weight", "classifier.6.bias", "classifier.4.weight", "classifier.4.bias", "classifier.0.weight", "stem.2.weight", "stem.0.weight"):
                scale = float(gb[n].abs().max()) + 1e-12
                err = ((ga[n] - gb[n]).abs().reshape(-1) / scale)
                l2 = float((ga[n] - gb[n]).double().norm() / gb[n].double().norm())
                print("    %-22s median %.2e max %.2e l2 %.2e" % (n, float(err.median()), float(err.max()), l2))
            for n in wa:
                print("    weights before this step, %-20s max |diff| %.3e" % (n, float((wa[n] - wb[n]).abs().max())))
            print("    pooled max |diff| %.3e, pre-activation max |diff| %.3e" % (float((xa - xb).abs().max()), float((ha - hb).abs().max())))
            flips = ((ha > 0) != (hb > 0)).nonzero()
            print("    gate flips (row, unit):", flips.tolist()[:20], "of", flips.size(0))
            for r, u in flips.tolist()[:20]:
                print("      row %d unit %d: pre-activation %.3e / %.3e" % (r, u, float(ha[r, u]), float(hb[r, u])))
            db = (ga["classifier.4.bias"] - gb["classifier.4.bias"]).abs()
            units = (db > 1e-3 * float(gb["classifier.4.bias"].abs().max())).nonzero().flatten().tolist()
            print("    units whose bias gradient differs:", units[:20], "of", len(units))
            if not same_prog:
                print("    rows with different programs:", (pa != pb).any(1).nonzero().flatten().tolist()[:10])
            break
    if bad:
        break
