"""The drop-in claim as a test: the product's ``probnmn.models`` / ``probnmn.modules`` classes against the
reference's class surface recorded in tests/golden/surface.json by oracle/make_surface.py (imported where
the reference imports in the build container, parsed where it needs allennlp 0.9.0) -- constructor and
``forward`` signatures with their defaults, ``from_config`` and the config keys it reads, return-dict and
metric keys, package exports, and the NMN's ``state_dict`` keys and shapes (SURVEY 8b, App. D;
/root/reference/scripts/train.py:10-22,125-126, probnmn/models/__init__.py:1-7).

Rule for signatures: the reference's parameters, in order, with the same defaults, must be a prefix of
the product's; the product may append keyword parameters WITH defaults (e.g. ``max_decoding_steps`` on
the generator, which the reference hard-codes -- BASELINE config 5 needs it)."""
import inspect
import json
import os

import pytest
import torch

import probnmn.models as models
from probnmn.modules import elbo, nmn_modules, seq2seq_base
from probnmn.vocabulary import Vocabulary

from fixtures import namespaces


@pytest.fixture(scope="module")
def surface(golden_dir):
    with open(os.path.join(golden_dir, "surface.json")) as f:
        return json.load(f)


def _product_class(name):
    for mod in (models, nmn_modules, elbo, seq2seq_base):
        if hasattr(mod, name):
            return getattr(mod, name)
    raise AssertionError("class %s is missing from the product" % name)


def _norm(v):
    return list(v) if isinstance(v, tuple) else v


def _check_signature(cls_name, method, want, fn):
    got = [(n, p) for n, p in inspect.signature(fn).parameters.items() if n not in ("self", "cls")]
    assert len(got) >= len(want), (cls_name, method, [n for n, _ in got])
    for (name, p), w in zip(got, want):
        assert name == w["name"], (cls_name, method, name, w["name"])
        has = p.default is not inspect._empty
        assert has == w["has_default"], (cls_name, method, name)
        if has:
            assert _norm(p.default) == w["default"], (cls_name, method, name, p.default, w["default"])
    for name, p in got[len(want):]:  # product-only parameters must be optional
        assert p.default is not inspect._empty or p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL), (cls_name, method, name)


def test_every_reference_class_exists_with_the_same_signatures(surface):
    for cls_name, entry in surface["classes"].items():
        cls = _product_class(cls_name)
        for method, m in entry["methods"].items():
            if method == "_forward_loop":
                continue  # private; recorded for its return-dict keys only
            assert hasattr(cls, method), (cls_name, method)
            fn = getattr(cls, method)
            fn = fn.__func__ if inspect.ismethod(fn) else fn
            _check_signature(cls_name, method, m["signature"], fn)


def test_package_exports(surface):
    assert sorted(models.__all__) == sorted(surface["models_all"])
    for name in surface["models_all"]:
        assert inspect.isclass(getattr(models, name))
    for name in ("Seq2SeqBase",):
        assert issubclass(models.ProgramGenerator, getattr(seq2seq_base, name))
        assert issubclass(models.QuestionReconstructor, getattr(seq2seq_base, name))


def test_from_config_reads_the_reference_keys(surface):
    """from_config of the product reads exactly the config keys the reference's does (attribute chains on
    `_C` in the source), and the product's Config look-alike provides every one of them."""
    import ast
    import textwrap

    from probnmn.config import Config

    cfg = Config()
    for cls_name, entry in surface["classes"].items():
        fc = entry["methods"].get("from_config")
        if not fc:
            continue
        src = textwrap.dedent(inspect.getsource(_product_class(cls_name).from_config.__func__))
        keys = set()
        for node in ast.walk(ast.parse(src)):
            if isinstance(node, ast.Attribute):
                chain, cur = [], node
                while isinstance(cur, ast.Attribute):
                    chain.append(cur.attr)
                    cur = cur.value
                if isinstance(cur, ast.Name) and cur.id == "_C":
                    keys.add(".".join(reversed(chain)))
        keys = {k for k in keys if not any(o != k and o.startswith(k + ".") for o in keys)}
        assert keys == set(fc["config_keys"]), (cls_name, keys, fc["config_keys"])
        for k in fc["config_keys"]:
            node = cfg
            for part in k.split("."):
                node = getattr(node, part)


def test_nmn_state_dict_keys_and_shapes(surface):
    want = surface["classes"]["NeuralModuleNetwork"]["state_dict"]
    net = models.NeuralModuleNetwork(Vocabulary.clevr())
    got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    assert len(want) == 222
    assert got == want  # same keys, same shapes, same order (what an optimizer built from parameters() sees)
    for name in ("AttentionModule", "QueryModule", "RelateModule", "SameModule", "ComparisonModule"):
        m = getattr(nmn_modules, name)(128)
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == surface["classes"][name]["state_dict"], name


def test_seq2seq_parameter_names_follow_allennlp(surface):
    """NOT pinned by /root/reference (allennlp 0.9.0 is absent): the names restate allennlp's module
    structure and are what the release checkpoints hold."""
    names = surface["allennlp_parameter_names"]
    vocab = Vocabulary.clevr()
    for cls in (models.ProgramGenerator, models.QuestionReconstructor):
        assert list(cls(vocab).state_dict()) == names["Seq2SeqBase"], cls.__name__
    assert list(models.ProgramPrior(vocab).state_dict()) == names["ProgramPrior"]
    sd = models.ProgramPrior(vocab).state_dict()
    assert sd["_output_layer.weight"].data_ptr() == sd["_embedder.token_embedder_programs.weight"].data_ptr()  # tied


def test_vocabulary_matches_the_fixture_namespaces():
    v, ns = Vocabulary.clevr(), namespaces()
    for name in ("programs", "answers"):
        assert [v.get_token_from_index(i, name) for i in range(v.get_vocab_size(name))] == ns[name]


@pytest.mark.gpu
def test_return_and_metric_keys_on_device(surface):
    """forward's return dicts and get_metrics' keys (needs the kernels, hence the MI355X)."""
    from probnmn.data.synthetic import synthetic_batch

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    nmn = models.NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=32).to(dev)
    pg, qr = models.ProgramGenerator(vocab).to(dev), models.QuestionReconstructor(vocab).to(dev)
    prior = models.ProgramPrior(vocab).to(dev)
    b = {k: v.to(dev) for k, v in synthetic_batch(vocab, 6, seed=1).items()}
    S = surface["classes"]
    nmn.train()
    out = nmn(b["image"], b["program"], b["answer"])
    assert sorted(out) == S["NeuralModuleNetwork"]["forward_keys_train"]
    assert sorted(out["metrics"]) == S["NeuralModuleNetwork"]["metrics_keys"] == sorted(nmn.get_metrics())
    nmn.eval()
    with torch.no_grad():
        assert sorted(nmn(b["image"], b["program"])) == S["NeuralModuleNetwork"]["forward_keys_eval"]
    loop_keys = set(S["Seq2SeqBase"]["methods"]["_forward_loop"]["dict_keys"])
    for m, src, tgt in ((pg, b["question"], b["program"]), (qr, b["program"], b["question"])):
        m.train()
        assert set(m(src, None, "sampling")) == loop_keys == {"predictions", "loss"}
        assert m.get_metrics() == {}
        m.eval()
        with torch.no_grad():
            assert set(m(src, tgt, "greedy")) == loop_keys
        want = set(S["Seq2SeqBase"]["methods"]["get_metrics"]["dict_keys"]) | set(
            surface["allennlp_parameter_names"]["Seq2SeqBase_metrics_from_allennlp"])
        assert set(m.get_metrics()) == want
    prior.eval()
    with torch.no_grad():
        assert set(prior(b["program"])) == set(S["ProgramPrior"]["methods"]["forward"]["dict_keys"])
    assert set(prior.get_metrics()) == set(S["ProgramPrior"]["methods"]["get_metrics"]["dict_keys"])
    nmn.train(), pg.train(), qr.train()
    qc = elbo.QuestionCodingElbo(pg, qr, prior)
    assert sorted(qc(b["question"])) == S["QuestionCodingElbo"]["forward_keys"]
    for objective in ("ours", "baseline"):
        je = elbo.JointTrainingElbo(pg, qr, prior, nmn, objective=objective)
        assert sorted(je(b["question"], b["image"], b["answer"])) == S["JointTrainingElbo"]["forward_keys_" + objective]
