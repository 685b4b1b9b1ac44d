"""Where the time of one validation batch goes (evaluate_answer_accuracy, 256 questions)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch, bench
from probnmn.models import NeuralModuleNetwork, ProgramGenerator
from probnmn.vocabulary import Vocabulary
dev = torch.device("cuda:0"); vocab = Vocabulary.clevr(); torch.manual_seed(0)
nmn = NeuralModuleNetwork(vocab).to(dev).eval(); pg = ProgramGenerator(vocab).to(dev).eval()
bs = [bench.device_batch(vocab, 256, 5000 + i, dev) for i in range(6)]
def sync(): torch.cuda.synchronize()
with torch.no_grad():
    for b in bs[:3]:
        o = pg(b["question"], b["program"], decoding_strategy="greedy"); nmn(b["image"], o["predictions"], b["answer"])
    sync()
    for name, fn in (("pg greedy teacher-forced", lambda b: pg(b["question"], b["program"], decoding_strategy="greedy")),
                     ("nmn forward (host programs)", lambda b: nmn(b["image"], b["program"].cpu(), b["answer"])),
                     ("nmn forward (device programs)", lambda b: nmn(b["image"], b["program"], b["answer"]))):
        t0 = time.perf_counter()
        for b in bs: fn(b)
        sync(); print("%-32s %.2f ms / batch" % (name, (time.perf_counter() - t0) / len(bs) * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for b in bs:
        o = pg(b["question"], b["program"], decoding_strategy="greedy"); nmn(b["image"], o["predictions"], b["answer"])
    sync(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
