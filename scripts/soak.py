"""Soak: many joint-training steps at several batch sizes / map sizes with fresh synthetic batches, checking that
every loss stays finite and no kernel traps or hangs (run under `timeout`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import JointTrainingStep
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for size, B, kw in ((14, 1024, {}), (14, 128, {}), (14, 37, {}), (28, 96, {"deep": True, "program_length": 40})):
    torch.manual_seed(size + B)
    nmn = NeuralModuleNetwork(vocab, image_feature_size=(1024, size, size)).to(dev)
    pg = ProgramGenerator(vocab, max_decoding_steps=40 if size == 28 else 26).to(dev)
    qr, prior = QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
    batches = [bench.device_batch(vocab, B, 100 + i, dev, image_feature_size=(1024, size, size), **kw) for i in range(3)]
    bench.fit_program_generator(pg, vocab, batches[0], dev, 400, 0.9)
    step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
    t0 = time.time()
    for i in range(steps):
        out = step.step(batches[i % 3])
        if i % 25 == 24 or i == steps - 1:
            obj = float(out["objective"])
            assert obj == obj and abs(obj) < 1e9, (size, B, i, obj)
    torch.cuda.synchronize()
    print("maps %dx%d, %4d questions: %d steps ok, %.1f ms/step, objective %.4f" % (size, size, B, steps, (time.time() - t0) / steps * 1e3, float(out["objective"])), flush=True)
    del step, nmn, pg, qr, prior, batches
    torch.cuda.empty_cache()
