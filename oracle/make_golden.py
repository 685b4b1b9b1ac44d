"""Generate tests/golden/* by running the REAL reference (test infrastructure).

Runs only in the build container, where ``/root/reference`` exists; the GPU box never sees the
reference, it only reads the vectors this script wrote.  Nothing of the reference's source is
copied: the files are loaded from where they lie, with stand-ins for the three imports that are
absent from this image (``allennlp.data.Vocabulary``, ``allennlp.training.metrics``,
``yacs.config.CfgNode``) -- none of which takes part in the arithmetic being pinned.

    python oracle/make_golden.py            # rewrites tests/golden/

What is pinned:
  nmn_modules_full.npz   the 7 module kinds at D=128, 14x14 (outputs only; inputs/weights come
                         from oracle/detgen.py seeds recorded in the file)
  nmn_small.npz          full network at reduced dims: logits, predictions, loss, validity and
                         every parameter gradient of loss.mean()
  nmn_modules_full_28.npz, nmn_small_28.npz
                         the same at 28x28 maps (BASELINE config 5; the reference modules are
                         size-agnostic, nmn_modules.py:72-244, nmn.py:46-53,133); the network
                         fixture adds programs of up to 40 tokens (LONG_CASES)
  nmn_validity.json      program -> valid table from the reference's try/except interpreter
  elbo_known.json        Reinforce / ELBO outputs and the moving baseline over two calls

``SameModule`` is run under torch-1.4 semantics (the reference's pin): integer ``Tensor / int``
floors.  On torch >= 1.5 the literal code raises and the example is scored invalid instead.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import detgen, nmn_oracle  # noqa: E402


# ---- stand-ins for absent third-party imports ---------------------------------------------------
class _Vocab:
    def __init__(self, namespaces):
        self._itos = {k: dict(enumerate(v)) for k, v in namespaces.items()}
        self._stoi = {k: {t: i for i, t in enumerate(v)} for k, v in namespaces.items()}

    def get_index_to_token_vocabulary(self, namespace="tokens"):
        return self._itos[namespace]

    def get_token_to_index_vocabulary(self, namespace="tokens"):
        return self._stoi[namespace]

    def get_token_from_index(self, index, namespace="tokens"):
        return self._itos[namespace][int(index)]

    def get_token_index(self, token, namespace="tokens"):
        return self._stoi[namespace].get(token, self._stoi[namespace]["@@UNKNOWN@@"])

    def get_vocab_size(self, namespace="tokens"):
        return len(self._itos[namespace])


class _Average:
    def __init__(self):
        self.t, self.n = 0.0, 0

    def __call__(self, v):
        self.t += float(v)
        self.n += 1

    def get_metric(self, reset=False):
        r = self.t / self.n if self.n else 0.0
        if reset:
            self.t, self.n = 0.0, 0
        return r


class _BooleanAccuracy:
    def __init__(self):
        self.c, self.n = 0.0, 0.0

    def __call__(self, p, g, mask=None):
        self.c += float((p == g).sum())
        self.n += float(g.numel())

    def get_metric(self, reset=False):
        r = self.c / self.n if self.n else 0.0
        if reset:
            self.c, self.n = 0.0, 0.0
        return r


def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("allennlp")
    mod("allennlp.data", Vocabulary=_Vocab)
    mod("allennlp.training")
    mod("allennlp.training.metrics", Average=_Average, BooleanAccuracy=_BooleanAccuracy)
    mod("yacs")
    mod("yacs.config", CfgNode=dict)
    mod("probnmn")
    mod("probnmn.config", Config=object)
    mod("probnmn.modules")


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class torch14_integer_division:
    """Inside this context ``LongTensor / int`` floors, as under the reference's torch 1.4."""

    def __enter__(self):
        self._orig = torch.Tensor.__truediv__

        def div(a, b):
            if not a.is_floating_point() and isinstance(b, int):
                return torch.div(a, b, rounding_mode="floor")
            return self._orig(a, b)

        torch.Tensor.__truediv__ = div

    def __exit__(self, *exc):
        torch.Tensor.__truediv__ = self._orig


# ---- shared fixture definitions (also imported by the tests) -----------------------------------
def clevr_program_tokens():
    colors = ["blue", "brown", "cyan", "gray", "green", "purple", "red", "yellow"]
    toks = ["count", "exist", "greater_than", "less_than", "intersect", "union", "scene", "unique"]
    toks += ["equal_" + k for k in ("color", "integer", "material", "shape", "size")]
    toks += ["query_" + k for k in ("color", "material", "shape", "size")]
    toks += ["same_" + k for k in ("color", "material", "shape", "size")]
    toks += ["filter_color[%s]" % v for v in colors]
    toks += ["filter_material[%s]" % v for v in ("metal", "rubber")]
    toks += ["filter_shape[%s]" % v for v in ("cube", "cylinder", "sphere")]
    toks += ["filter_size[%s]" % v for v in ("large", "small")]
    toks += ["relate[%s]" % v for v in ("behind", "front", "left", "right")]
    return sorted(toks)


def namespaces():
    special = ["@@PADDING@@", "@@UNKNOWN@@", "@start@", "@end@"]
    answers = sorted(
        [str(i) for i in range(11)]
        + ["blue", "brown", "cyan", "gray", "green", "purple", "red", "yellow"]
        + ["metal", "rubber", "cube", "cylinder", "sphere", "large", "small", "yes", "no"]
    )
    return {
        "programs": special + clevr_program_tokens(),
        "questions": special + ["w%03d" % i for i in range(96)],
        "answers": answers + ["@@UNKNOWN@@"],
    }


VALIDITY_CASES = [
    "",
    "@end@",
    "scene",
    "intersect scene",
    "union",
    "count count scene",
    "count scene",
    "count",
    "equal_integer count scene count scene",
    "count filter_color[red] scene",
    "count filter_color[red] filter_shape[cube] scene",
    "query_color unique filter_shape[cube] relate[left] unique filter_color[red] scene",
    "exist filter_size[large] relate[behind] unique filter_material[metal] relate[front] unique filter_shape[sphere] scene",
    "greater_than count filter_color[blue] filter_shape[cube] scene count filter_size[small] scene",
    "count intersect filter_color[red] relate[left] unique filter_shape[cube] scene filter_size[large] relate[right] unique filter_material[rubber] scene",
    "count union filter_color[red] scene filter_shape[cube] scene",
    "query_shape unique same_color unique filter_size[small] filter_material[metal] scene",
    "equal_color query_color unique filter_shape[cube] scene query_color unique filter_size[large] relate[front] unique filter_material[metal] scene",
    "filter_color[red] scene",
    "relate[left] scene",
    "same_size scene",
    "query_color query_color scene",
    "filter_color[red] query_color scene",
    "equal_shape count scene scene",
    "equal_shape scene",
    "less_than filter_color[red] scene count scene",
    "intersect count scene count scene",
    "count intersect scene",
    "intersect intersect filter_color[red] scene filter_shape[cube] scene",
    "count filter_color[red] scene count filter_shape[cube] scene count scene",
    "@start@ count filter_color[red] scene @end@",
    "count @@UNKNOWN@@ filter_color[red] unique scene",
    "union filter_color[red] scene",
    "count same_shape filter_color[red] scene",
    "exist same_material unique relate[right] unique filter_size[large] scene",
    "query_size relate[behind] query_size scene",
]


# BASELINE config 5: "program length <= 40" -- deeper module chains than any CLEVR template
LONG_CASES = [
    # five hops (20 tokens)
    "query_color unique filter_shape[cube] relate[left] unique filter_color[red] relate[behind] unique "
    "filter_size[large] relate[front] unique filter_material[metal] relate[right] unique filter_shape[sphere] "
    "relate[left] unique filter_color[blue] scene",
    # comparison of two three-hop chains (27 tokens)
    "equal_material query_material unique filter_shape[cube] relate[left] unique filter_color[red] relate[behind] "
    "unique filter_size[large] scene query_material unique filter_color[green] relate[front] unique "
    "filter_material[metal] relate[right] unique filter_shape[sphere] relate[left] unique filter_size[small] scene",
    # and / or of long chains under a count, with a same-attribute hop (33 tokens)
    "count union filter_color[red] relate[left] unique filter_shape[cube] same_size unique filter_material[rubber] "
    "relate[front] unique filter_size[large] scene intersect filter_color[cyan] relate[behind] unique "
    "filter_shape[cylinder] scene filter_size[small] relate[right] unique filter_material[metal] relate[left] "
    "unique filter_color[yellow] filter_shape[sphere] scene",
    # 40 tokens: integer comparison of two counts over four-hop chains
    "less_than count filter_color[gray] relate[left] unique filter_shape[cube] relate[behind] unique "
    "filter_size[large] relate[front] unique filter_material[metal] relate[right] unique filter_color[purple] "
    "relate[left] unique filter_shape[cylinder] filter_size[small] scene "
    "count filter_shape[sphere] filter_size[small] relate[left] unique filter_color[brown] relate[behind] unique "
    "filter_material[rubber] relate[front] unique filter_shape[cylinder] relate[right] unique filter_color[green] "
    "relate[behind] unique filter_material[metal] filter_size[large] scene",
    # 40 tokens that end up invalid (a query fed by an encoding)
    "query_shape query_color unique filter_color[gray] relate[left] unique filter_shape[cube] relate[behind] unique "
    "filter_size[large] relate[front] unique filter_material[metal] relate[right] unique filter_color[purple] "
    "relate[left] unique filter_shape[sphere] relate[behind] unique filter_size[small] relate[front] unique "
    "filter_color[brown] relate[right] unique filter_material[rubber] relate[left] unique filter_shape[cylinder] "
    "relate[front] unique filter_size[large] relate[behind] unique filter_color[cyan] filter_color[green] filter_shape[cube] scene",
]
assert max(len(c.split()) for c in LONG_CASES) == 40


def encode_programs(cases, stoi, length=26):
    rows = []
    for case in cases:
        ids = [stoi[t] for t in case.split()]
        assert len(ids) <= length, case
        rows.append(ids + [0] * (length - len(ids)))
    return torch.tensor(rows, dtype=torch.long)


SMALL_DIMS = dict(
    image_feature_size=(16, 14, 14),
    module_channels=8,
    class_projection_channels=16,
    classifier_linear_size=32,
)


def small_dims(size=14):
    return dict(SMALL_DIMS, image_feature_size=(16, size, size))


def small_network_inputs(size=14):
    """size 14: the 36 validity cases at length 26; size 28 (config 5): + LONG_CASES, length 40."""
    ns = namespaces()
    stoi = {t: i for i, t in enumerate(ns["programs"])}
    if size == 14:
        programs = encode_programs(VALIDITY_CASES, stoi)
    else:
        programs = encode_programs(VALIDITY_CASES + LONG_CASES, stoi, length=40)
    B = programs.size(0)
    gen = detgen.rng(1234 if size == 14 else 1234 + size)
    features = torch.relu(detgen.normal(gen, (B, 16, size, size)))
    answers = torch.from_numpy(gen.integers(0, 28, size=(B,))).long()
    shapes = nmn_oracle.nmn_param_shapes(ns["programs"][4:], **small_dims(size))
    sd = detgen.fill_state_dict(shapes, seed=99)
    return ns, programs, features, answers, sd


def full_module_inputs(size=14):
    gen = detgen.rng(7 if size == 14 else 7 + size)
    feats = torch.relu(detgen.normal(gen, (1, 128, size, size)))
    feats2 = torch.relu(detgen.normal(gen, (1, 128, size, size)))
    attn = torch.sigmoid(detgen.normal(gen, (1, 1, size, size), 2.0))
    attn2 = torch.sigmoid(detgen.normal(gen, (1, 1, size, size), 2.0))
    toks = {
        "attention": "filter_color[red]",
        "query": "query_color",
        "relate": "relate[left]",
        "same": "same_shape",
        "comparison": "equal_color",
    }
    shapes = nmn_oracle.nmn_param_shapes(list(toks.values()))
    shapes = {k: v for k, v in shapes.items() if not k.startswith(("stem", "classifier"))}
    sd = detgen.fill_state_dict(shapes, seed=11)
    return feats, feats2, attn, attn2, toks, sd


# ---- generation ---------------------------------------------------------------------------------
def main():
    os.makedirs(OUT, exist_ok=True)
    _install_shims()
    ref_modules = _load("probnmn.modules.nmn_modules", "probnmn/modules/nmn_modules.py")
    ref_nmn = _load("probnmn.models.nmn", "probnmn/models/nmn.py")

    for size in (14, 28):
        _modules_golden(ref_modules, size)
        _network_golden(ref_nmn, size)
    _elbo_golden()
    print("wrote", sorted(os.listdir(OUT)))


def _modules_golden(ref_modules, size):
    # 1. the seven modules at full size ---------------------------------------------------------
    suffix = "" if size == 14 else "_%d" % size
    feats, feats2, attn, attn2, toks, sd = full_module_inputs(size)

    def build(cls, tok):
        m = cls(128)
        own = {k[len(tok) + 1 :]: v for k, v in sd.items() if k.startswith(tok + ".")}
        m.load_state_dict(own)
        return m

    with torch.no_grad(), torch14_integer_division():
        gold = {
            "and_1_1": ref_modules.AndModule()(attn, attn2),
            "or_1_1": ref_modules.OrModule()(attn, attn2),
            "and_1_128": ref_modules.AndModule()(attn, feats),
            "or_128_128": ref_modules.OrModule()(feats, feats2),
            "attention": build(ref_modules.AttentionModule, toks["attention"])(feats, attn),
            "query": build(ref_modules.QueryModule, toks["query"])(feats, attn),
            "relate": build(ref_modules.RelateModule, toks["relate"])(feats, attn),
            "same": build(ref_modules.SameModule, toks["same"])(feats, attn),
            "comparison": build(ref_modules.ComparisonModule, toks["comparison"])(feats, feats2),
        }
        mine = {
            "and_1_1": nmn_oracle.and_module(attn, attn2),
            "or_1_1": nmn_oracle.or_module(attn, attn2),
            "and_1_128": nmn_oracle.and_module(attn, feats),
            "or_128_128": nmn_oracle.or_module(feats, feats2),
            "attention": nmn_oracle.attention_module(sd, toks["attention"], feats, attn),
            "query": nmn_oracle.query_module(sd, toks["query"], feats, attn),
            "relate": nmn_oracle.relate_module(sd, toks["relate"], feats, attn),
            "same": nmn_oracle.same_module(sd, toks["same"], feats, attn),
            "comparison": nmn_oracle.comparison_module(sd, toks["comparison"], feats, feats2),
        }
    for k in gold:
        assert torch.equal(gold[k], mine[k]), "oracle != reference for module " + k
    np.savez_compressed(
        os.path.join(OUT, "nmn_modules_full%s.npz" % suffix), **{k: v.numpy() for k, v in gold.items()}
    )


def _network_golden(ref_nmn, size):
    # 2. + 3. full network at reduced dims, validity table ------------------------------------------
    suffix = "" if size == 14 else "_%d" % size
    ns, programs, features, answers, sd = small_network_inputs(size)
    vocab = _Vocab(ns)
    torch.manual_seed(0)
    net = ref_nmn.NeuralModuleNetwork(vocab, **small_dims(size))
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net.train()
    with torch14_integer_division():
        out = net(features, programs, answers)
        out["loss"].mean().backward()
        net.eval()
        with torch.no_grad():
            out_noans = net(features, programs)
    ref_grads = {
        k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()
    }

    sd_req = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mine = nmn_oracle.nmn_forward(sd_req, vocab._itos["programs"], features, programs, answers)
    mine["loss"].mean().backward()
    mine_noans = nmn_oracle.nmn_forward(sd, vocab._itos["programs"], features, programs, None)
    assert torch.equal(mine["predictions"], out["predictions"])
    assert torch.equal(mine["loss"].detach(), out["loss"].detach())
    assert torch.equal(mine_noans["loss"], out_noans["loss"])
    for k, g in ref_grads.items():
        og = sd_req[k].grad if sd_req[k].grad is not None else torch.zeros_like(sd_req[k])
        assert torch.allclose(og, g, rtol=0, atol=0), "oracle grad != reference grad for " + k

    # recover the reference's validity from its outputs (pred == 28 <=> invalid)
    valid = (out["predictions"] != 28).long()
    assert torch.equal(valid, mine["valid"])
    arrays = {
        "predictions": out["predictions"].numpy(),
        "loss": out["loss"].detach().numpy(),
        "loss_without_answers": out_noans["loss"].numpy(),
        "valid": valid.numpy(),
        "logits": mine["logits"].detach().numpy(),
    }
    for k, g in ref_grads.items():
        arrays["grad::" + k] = g.numpy()
    np.savez_compressed(os.path.join(OUT, "nmn_small%s.npz" % suffix), **arrays)
    cases = VALIDITY_CASES if size == 14 else VALIDITY_CASES + LONG_CASES
    with open(os.path.join(OUT, "nmn_validity%s.json" % suffix), "w") as f:
        json.dump({c: int(v) for c, v in zip(cases, valid.tolist())}, f, indent=1)


def _elbo_golden():
    # 4. REINFORCE / ELBO known answers ----------------------------------------------------------
    sys.modules["probnmn.models"] = types.ModuleType("probnmn.models")
    for n in ("ProgramGenerator", "ProgramPrior", "QuestionReconstructor", "NeuralModuleNetwork"):
        setattr(sys.modules["probnmn.models"], n, object)
    ref_elbo = _load("probnmn.modules.elbo", "probnmn/modules/elbo.py")
    e = ref_elbo._ElboWithReinforce(beta=0.1, baseline_decay=0.99)
    record = {"calls": []}
    for _ in range(2):
        logq = torch.tensor([-1.0, -2.0], requires_grad=True)
        rec = torch.tensor([-3.0, -1.0], requires_grad=True)
        reward = torch.tensor([0.5, 1.5])
        o = e._forward(logq, rec, reward)
        (-o["elbo"]).backward()
        record["calls"].append(
            {
                **{k: float(v) for k, v in o.items()},
                "baseline_after": float(e._reinforce._reinforce_baseline),
                "dneg_elbo_dlogq": logq.grad.tolist(),
                "dneg_elbo_drec": rec.grad.tolist(),
            }
        )

    # JointTrainingElbo / QuestionCodingElbo with duck-typed models returning fixed tensors
    gen = detgen.rng(5)
    Bn = 6
    fixed = {
        "pg_loss": torch.from_numpy(gen.uniform(0.5, 3.0, Bn).astype(np.float32)),
        "qr_loss": torch.from_numpy(gen.uniform(0.5, 3.0, Bn).astype(np.float32)),
        "prior_loss": torch.from_numpy(gen.uniform(0.5, 3.0, Bn).astype(np.float32)),
        "nmn_loss": torch.from_numpy(gen.uniform(0.5, 3.5, Bn).astype(np.float32)),
    }
    leaves = {k: v.clone().requires_grad_(True) for k, v in fixed.items()}

    def duck(key):
        return lambda *a, **kw: {"predictions": torch.zeros(Bn, 3, dtype=torch.long), "loss": leaves[key]}

    record["fixed_losses"] = {k: v.tolist() for k, v in fixed.items()}
    for objective in ("ours", "baseline"):
        for v in leaves.values():
            v.grad = None
        je = ref_elbo.JointTrainingElbo(
            duck("pg_loss"), duck("qr_loss"), duck("prior_loss"), duck("nmn_loss"),
            beta=0.1, gamma=1.0, baseline_decay=0.99, objective=objective,
        )
        o = je(None, None, None)
        nmn_loss = o.pop("nmn_loss")
        (1.0 * nmn_loss - o["elbo"]).backward()
        record["joint_" + objective] = {
            **{k: float(v) for k, v in o.items()},
            "nmn_loss": float(nmn_loss),
            "baseline_after": float(je._reinforce._reinforce_baseline),
            "grads": {k: (v.grad.tolist() if v.grad is not None else None) for k, v in leaves.items()},
        }
    for v in leaves.values():
        v.grad = None
    qe = ref_elbo.QuestionCodingElbo(
        duck("pg_loss"), duck("qr_loss"), duck("prior_loss"), beta=0.1, baseline_decay=0.99
    )
    o = qe(None)
    (-o["elbo"]).backward()
    record["question_coding"] = {
        **{k: float(v) for k, v in o.items()},
        "baseline_after": float(qe._reinforce._reinforce_baseline),
        "grads": {k: (v.grad.tolist() if v.grad is not None else None) for k, v in leaves.items()},
    }
    with open(os.path.join(OUT, "elbo_known.json"), "w") as f:
        json.dump(record, f, indent=1)


if __name__ == "__main__":
    main()
