"""Record the reference's class surface -- what scripts/train.py reaches through probnmn.trainers /
probnmn.evaluators -- into tests/golden/surface.json (test infrastructure; runs only in the
build container, where /root/reference exists).

    python oracle/make_surface.py

Two sources, stated per entry:
  "import"  the class was imported from /root/reference (with the stand-ins of oracle/make_golden.py for
            allennlp.data.Vocabulary, allennlp.training.metrics and yacs) and inspected / instantiated:
            signatures, defaults, forward's return keys, state_dict keys and shapes;
  "ast"     the file cannot be imported here (it subclasses allennlp 0.9.0, absent from this image) and
            was parsed instead: signatures and defaults from the ``def`` nodes, return / metric keys from
            the dict literals of ``forward`` / ``get_metrics``, config keys from ``from_config``.
The parameter names of the allennlp-built seq2seq models are NOT derivable from
/root/reference; they are recorded under "allennlp_parameter_names" as what they are -- a restatement of
allennlp 0.9.0's module structure -- and the test that uses them says so.
"""
import ast
import inspect
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import make_golden as mg  # noqa: E402


def _default(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return ast.unparse(node)


def _ast_signature(fn: ast.FunctionDef):
    args = fn.args
    names = [a.arg for a in args.args]
    defaults = [None] * (len(names) - len(args.defaults)) + [_default(d) for d in args.defaults]
    out = []
    for n, d, has in zip(names, defaults, [False] * (len(names) - len(args.defaults)) + [True] * len(args.defaults)):
        if n in ("self", "cls"):
            continue
        out.append({"name": n, "has_default": has, "default": d if has else None})
    return out


def _dict_keys(fn: ast.FunctionDef):
    """String keys of the dicts a method RETURNS: dict literals in ``return`` statements, assigned to a
    name like ``output_dict`` / ``all_metrics``, passed to ``<such a name>.update(...)``, and subscript
    stores on such names (dict literals used as call arguments, e.g. embedder inputs, are not results)."""
    def lit(d):
        return [k.value for k in d.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)]

    def result_name(n):
        return isinstance(n, ast.Name) and ("dict" in n.id or "metrics" in n.id)

    keys = []
    for node in ast.walk(fn):
        if isinstance(node, ast.Return) and isinstance(node.value, ast.Dict):
            keys += lit(node.value)
        if isinstance(node, (ast.Assign, ast.AnnAssign)) and isinstance(node.value, ast.Dict):
            targets = node.targets if isinstance(node, ast.Assign) else [node.target]
            if any(result_name(t) for t in targets):
                keys += lit(node.value)
        if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "update"
                and result_name(node.func.value)):
            for arg in node.args:
                if isinstance(arg, ast.Dict):
                    keys += lit(arg)
        if isinstance(node, ast.Subscript) and isinstance(node.ctx, ast.Store) and result_name(node.value):
            sl = node.slice
            if isinstance(sl, ast.Constant) and isinstance(sl.value, str):
                keys.append(sl.value)
    return sorted(set(keys))


def _config_keys(fn: ast.FunctionDef):
    keys = []
    for node in ast.walk(fn):
        if isinstance(node, ast.Attribute):
            chain, cur = [], node
            while isinstance(cur, ast.Attribute):
                chain.append(cur.attr)
                cur = cur.value
            if isinstance(cur, ast.Name) and cur.id == "_C":
                keys.append(".".join(reversed(chain)))
    # keep only the longest chains (drop prefixes)
    return sorted(k for k in set(keys) if not any(o != k and o.startswith(k + ".") for o in keys))


def ast_classes(relpath):
    tree = ast.parse(open(os.path.join(REF, relpath)).read())
    out = {}
    for node in tree.body:
        if not isinstance(node, ast.ClassDef):
            continue
        entry = {"source": "ast", "file": relpath, "bases": [ast.unparse(b) for b in node.bases], "methods": {}}
        for item in node.body:
            if isinstance(item, ast.FunctionDef) and (not item.name.startswith("_") or item.name in ("__init__", "_forward_loop")):
                m = {"signature": _ast_signature(item)}
                if item.name in ("forward", "_forward_loop", "get_metrics"):
                    m["dict_keys"] = _dict_keys(item)
                if item.name == "from_config":
                    m["config_keys"] = _config_keys(item)
                entry["methods"][item.name] = m
        out[node.name] = entry
    return out


def live_signature(fn):
    out = []
    for n, p in inspect.signature(fn).parameters.items():
        if n in ("self", "cls"):
            continue
        has = p.default is not inspect._empty
        d = p.default if has else None
        if isinstance(d, tuple):
            d = list(d)
        out.append({"name": n, "has_default": has, "default": d})
    return out


def main():
    mg._install_shims()
    ref_modules = mg._load("probnmn.modules.nmn_modules", "probnmn/modules/nmn_modules.py")
    ref_nmn = mg._load("probnmn.models.nmn", "probnmn/models/nmn.py")
    import types

    sys.modules["probnmn.models"] = types.ModuleType("probnmn.models")
    for n in ("ProgramGenerator", "ProgramPrior", "QuestionReconstructor", "NeuralModuleNetwork"):
        setattr(sys.modules["probnmn.models"], n, object)
    ref_elbo = mg._load("probnmn.modules.elbo", "probnmn/modules/elbo.py")

    surface = {"classes": {}}
    # ---- imported classes ----------------------------------------------------------------------
    for mod, names in ((ref_modules, ["AndModule", "OrModule", "AttentionModule", "QueryModule", "RelateModule",
                                      "SameModule", "ComparisonModule", "Flatten"]),
                       (ref_nmn, ["NeuralModuleNetwork"]),
                       (ref_elbo, ["Reinforce", "QuestionCodingElbo", "JointTrainingElbo"])):
        for name in names:
            cls = getattr(mod, name)
            entry = {"source": "import", "file": os.path.relpath(inspect.getsourcefile(cls), REF), "methods": {}}
            for m in ("__init__", "forward", "from_config", "get_metrics"):
                if m in cls.__dict__:
                    fn = cls.__dict__[m]
                    fn = fn.__func__ if isinstance(fn, classmethod) else fn
                    entry["methods"][m] = {"signature": live_signature(fn)}
            surface["classes"][name] = entry

    # module parameter names / shapes at dim 128
    for name in ("AttentionModule", "QueryModule", "RelateModule", "SameModule", "ComparisonModule"):
        m = getattr(ref_modules, name)(128)
        surface["classes"][name]["state_dict"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]

    # the network: state_dict keys + shapes with the CLEVR vocabulary, forward's keys in train / eval mode
    vocab = mg._Vocab(mg.namespaces())
    torch.manual_seed(0)
    net = ref_nmn.NeuralModuleNetwork(vocab)
    # (a list: the order is what an optimizer built from parameters() sees)
    surface["classes"]["NeuralModuleNetwork"]["state_dict"] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    small = ref_nmn.NeuralModuleNetwork(vocab, **mg.SMALL_DIMS)
    ns, programs, features, answers, _ = mg.small_network_inputs()
    small.train()
    with mg.torch14_integer_division():
        out = small(features[:4], programs[:4], answers[:4])
        surface["classes"]["NeuralModuleNetwork"]["forward_keys_train"] = sorted(out)
        surface["classes"]["NeuralModuleNetwork"]["metrics_keys"] = sorted(out["metrics"])
        small.eval()
        with torch.no_grad():
            out = small(features[:4], programs[:4])
        surface["classes"]["NeuralModuleNetwork"]["forward_keys_eval"] = sorted(out)
    surface["classes"]["NeuralModuleNetwork"]["methods"]["from_config"]["config_keys"] = _config_keys(
        next(n for n in ast.walk(ast.parse(open(os.path.join(REF, "probnmn/models/nmn.py")).read()))
             if isinstance(n, ast.FunctionDef) and n.name == "from_config"))

    # elbo return keys with duck-typed models
    Bn = 3
    duck = lambda: (lambda *a, **kw: {"predictions": torch.zeros(Bn, 3, dtype=torch.long), "loss": torch.rand(Bn)})  # noqa: E731
    for objective in ("ours", "baseline"):
        je = ref_elbo.JointTrainingElbo(duck(), duck(), duck(), duck(), objective=objective)
        surface["classes"]["JointTrainingElbo"]["forward_keys_" + objective] = sorted(je(None, None, None))
    surface["classes"]["QuestionCodingElbo"]["forward_keys"] = sorted(ref_elbo.QuestionCodingElbo(duck(), duck(), duck())(None))

    # ---- parsed classes (allennlp subclasses) ------------------------------------------------------
    for rel in ("probnmn/modules/seq2seq_base.py", "probnmn/models/program_generator.py",
                "probnmn/models/question_reconstructor.py", "probnmn/models/program_prior.py"):
        surface["classes"].update(ast_classes(rel))

    # ---- package exports ---------------------------------------------------------------------------
    tree = ast.parse(open(os.path.join(REF, "probnmn/models/__init__.py")).read())
    surface["models_all"] = next(ast.literal_eval(n.value) for n in tree.body
                                 if isinstance(n, ast.Assign) and n.targets[0].id == "__all__")

    # ---- allennlp-built parameter names (restated, see the module docstring) -----------------------
    lstm = ["_encoder._module.%s_l%d" % (n, l) for l in (0, 1) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    surface["allennlp_parameter_names"] = {
        "source": "allennlp 0.9.0 module structure (SimpleSeq2Seq, BasicTextFieldEmbedder, PytorchSeq2SeqWrapper); "
                  "not derivable from /root/reference -- SURVEY App. D",
        "Seq2SeqBase": ["_source_embedder.token_embedder_tokens.weight"] + lstm + [
            "_target_embedder.weight", "_decoder_cell.weight_ih", "_decoder_cell.weight_hh", "_decoder_cell.bias_ih",
            "_decoder_cell.bias_hh", "_output_projection_layer.weight", "_output_projection_layer.bias"],
        "ProgramPrior": ["_embedder.token_embedder_programs.weight"] + lstm + ["_projection_layer.weight", "_output_layer.weight"],
        # get_metrics() merges allennlp's BLEU metric dict into the three keys its own literal holds (seq2seq_base.py:367)
        "Seq2SeqBase_metrics_from_allennlp": ["BLEU"],
    }
    path = os.path.join(ROOT, "tests", "golden", "surface.json")
    with open(path, "w") as f:
        json.dump(surface, f, indent=1, sort_keys=True)
    print("wrote", path, "classes:", sorted(surface["classes"]))


if __name__ == "__main__":
    main()
