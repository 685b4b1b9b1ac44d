"""CPU oracle for the probnmn-clevr hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain-PyTorch-CPU (fp32) restatement of the reference's
algorithm, each function citing the reference ``file:line`` it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and
only as the checker / the timed CPU baseline -- never from the product package
(``probnmn-clevr_amd/``), which fails loudly when its HIP library is missing instead of
falling back to anything here.

Pinning status
--------------
* NMN modules, NMN forward (program interpreter, validity, classifier, loss) and the
  REINFORCE / ELBO arithmetic are pinned against the real reference, imported in the build
  container by ``oracle/make_golden.py`` (stubs only for the absent ``allennlp`` / ``yacs``
  imports); the resulting vectors live in ``tests/golden/``.
* The seq2seq models (ProgramGenerator / QuestionReconstructor / ProgramPrior) subclass
  ``allennlp==0.9.0`` which is not vendored in the reference tree and not installable here:
  **parity unpinned** for that arithmetic.  It is restated from the reference call sites and
  the published AllenNLP 0.9.0 semantics and pinned only against torch
  primitives (``nn.LSTM``, ``nn.LSTMCell``, ``F.softmax``) and hand-derived known answers.
"""
