"""Graft the MI355X models onto the REFERENCE's ``probnmn`` package.

    # scripts/train.py (or evaluate.py / inference.py), before the first `import probnmn...`
    import sys; sys.path.insert(0, "/path/to/this/repo/probnmn-clevr_amd")
    import probnmn_graft; probnmn_graft.install()

After ``install()``:
  * ``probnmn`` is still the reference's package -- ``probnmn.trainers``, ``probnmn.evaluators``,
    ``probnmn.data``, ``probnmn.utils``, ``probnmn.config`` are untouched;
  * ``probnmn.models`` and ``probnmn.modules`` (and everything under them) are this build's: the classes
    ``scripts/train.py:10-22,125-126`` reaches through the trainers -- ``NeuralModuleNetwork``,
    ``ProgramGenerator``, ``QuestionReconstructor``, ``ProgramPrior``, ``QuestionCodingElbo``,
    ``JointTrainingElbo``, the seven modules -- run on the gfx950 kernels;
  * the build's own support modules, whose names the reference does not use (``probnmn._hip``,
    ``probnmn.runtime``, ``probnmn.optim``, ``probnmn.parallel``, ``probnmn.vocabulary``,
    ``probnmn.running_metrics``), resolve through the extended package path;
  * the fused iterations are importable as ``probnmn_amd_steps`` (``ModuleTrainingStep``,
    ``QuestionCodingStep``, ``JointTrainingStep``) without touching ``probnmn.trainers``.

Nothing of the reference is copied or modified; the graft only decides which file backs which module name.
(Without the reference on ``sys.path`` -- this repository's tests, bench.py -- put ``probnmn-clevr_amd/`` on
``sys.path`` and ``import probnmn``: the build's package is then complete on its own.)
"""
import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT = os.path.join(_HERE, "probnmn")
GRAFTED = ("models", "modules")


def _load_package(name: str, directory: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(directory, "__init__.py"),
                                                  submodule_search_locations=[directory])
    module = importlib.util.module_from_spec(spec)
    sys.modules[name] = module
    spec.loader.exec_module(module)
    return module


def install(reference_root: str = None):
    """``reference_root``: directory that contains the reference's ``probnmn/`` (optional when it is
    already importable).  Must run before anything imports ``probnmn.models`` / ``probnmn.modules``."""
    for sub in GRAFTED:
        if "probnmn." + sub in sys.modules:
            raise RuntimeError("probnmn.%s is already imported: call probnmn_graft.install() first" % sub)
    if reference_root is not None and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    if _HERE in sys.path:  # `import probnmn` must find the reference, not the build's own complete package
        sys.path.remove(_HERE)
        sys.path.append(_HERE)
    probnmn = importlib.import_module("probnmn")
    root = os.path.dirname(os.path.abspath(probnmn.__file__))
    if os.path.samefile(root, PRODUCT):
        raise RuntimeError("`import probnmn` resolved to the MI355X build itself; put the reference first on "
                           "sys.path (or pass reference_root=...)")
    if PRODUCT not in list(probnmn.__path__):
        probnmn.__path__.append(PRODUCT)  # product-only module names resolve here; shared names stay the reference's
    # probnmn.modules first: probnmn.models imports from it
    for sub in ("modules", "models"):
        module = _load_package("probnmn." + sub, os.path.join(PRODUCT, sub))
        setattr(probnmn, sub, module)
    steps = _load_package("probnmn_amd_steps", os.path.join(PRODUCT, "trainers"))
    from probnmn_amd_steps.joint_training import JointTrainingStep, QuestionCodingStep
    from probnmn_amd_steps.module_training import ModuleTrainingStep

    steps.JointTrainingStep, steps.QuestionCodingStep, steps.ModuleTrainingStep = (
        JointTrainingStep, QuestionCodingStep, ModuleTrainingStep)
    return probnmn
