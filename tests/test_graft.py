"""probnmn_graft.install(): only ``probnmn.models`` / ``probnmn.modules`` become this build's; the rest of
the REFERENCE's package (trainers, data, utils, config) keeps working -- run against the real reference in
the build container (the GPU box has no /root/reference: skipped there), in a fresh interpreter.  The
reference's own CheckpointManager then saves and restores the grafted NMN together with ClampAdam and the
lr scheduler (probnmn/utils/checkpointing.py:68-157, _trainer.py:120-130)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = textwrap.dedent('''
    import os, sys, types
    ROOT, REF = sys.argv[1], sys.argv[2]
    # stand-ins for third-party imports of the reference that this image lacks (none takes part in what is tested)
    class _Logger:
        def info(self, *a, **k): pass
    sys.modules["loguru"] = types.SimpleNamespace(logger=_Logger())
    sys.path.insert(0, os.path.join(ROOT, "probnmn-clevr_amd"))
    import probnmn_graft
    probnmn = probnmn_graft.install(REF)
    import torch
    under = lambda m, d: os.path.abspath(m.__file__).startswith(os.path.abspath(d) + os.sep)
    PRODUCT = os.path.join(ROOT, "probnmn-clevr_amd")
    assert under(probnmn, REF)
    import probnmn.models, probnmn.modules.elbo, probnmn.modules.nmn_modules, probnmn.modules.seq2seq_base
    for m in (probnmn.models, probnmn.models.nmn, probnmn.modules.elbo, probnmn.modules.nmn_modules, probnmn.modules.seq2seq_base):
        assert under(m, PRODUCT), m.__file__
    import probnmn.utils.checkpointing as ckpt          # the reference's, untouched
    assert under(ckpt, REF) and under(sys.modules["probnmn.utils"], REF)
    import probnmn.optim, probnmn.vocabulary, probnmn.running_metrics  # product-only names resolve through the extended path
    assert under(probnmn.optim, PRODUCT)
    import probnmn_amd_steps
    assert probnmn_amd_steps.JointTrainingStep.__name__ == "JointTrainingStep"
    for name in ("trainers", "data", "config", "evaluators"):  # still the reference's files on the package path
        spec = __import__("importlib").util.find_spec("probnmn." + name)
        assert os.path.abspath(spec.origin).startswith(os.path.abspath(REF) + os.sep), (name, spec.origin)

    # the reference's CheckpointManager drives the grafted model + ClampAdam + scheduler
    from probnmn.models import NeuralModuleNetwork
    from probnmn.optim import ClampAdam
    from probnmn.vocabulary import Vocabulary
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    nmn = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=32)
    opt = ClampAdam(nmn.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=0, threshold=1e-3)
    sched.step(0.5); sched.step(0.5)
    out = sys.argv[3]
    manager = ckpt.CheckpointManager(serialization_dir=out, keep_recent=100, optimizer=opt, scheduler=sched, nmn=nmn)
    manager.step(7, metric=0.25)
    saved = torch.load(os.path.join(out, "checkpoint_7.pth"))
    assert set(saved) == {"nmn", "optimizer", "scheduler", "iteration"} and saved["iteration"] == 7
    torch.manual_seed(1)
    nmn2 = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=32)
    opt2 = ClampAdam(nmn2.parameters(), lr=1.0)
    sched2 = torch.optim.lr_scheduler.ReduceLROnPlateau(opt2, mode="max", factor=0.5, patience=0, threshold=1e-3)
    it = ckpt.CheckpointManager(serialization_dir=out, optimizer=opt2, scheduler=sched2, nmn=nmn2).load(os.path.join(out, "checkpoint_7.pth"))
    assert it == 7 and opt2.lr == 5e-4 and sched2.best == sched.best
    for (k, a), (_, b) in zip(nmn.state_dict().items(), nmn2.state_dict().items()):
        assert torch.equal(a, b), k
    print("GRAFT-OK")
''')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "probnmn")), reason="needs the reference (build container only)")
def test_graft_replaces_models_and_modules_only(tmp_path):
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, REF, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GRAFT-OK" in r.stdout, r.stdout + r.stderr
