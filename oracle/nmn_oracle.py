"""CPU restatement of the Neural Module Network path (test infrastructure -- see oracle/__init__).

Functional style over a ``state_dict`` with the reference's key names, plain
torch CPU fp32 ops, autograd for the backward.  Follows:

* modules ............ reference probnmn/modules/nmn_modules.py:25-27, 43-45, 72-87, 111-123,
                       144-168, 194-208, 231-244
* network forward .... reference probnmn/models/nmn.py:183-275
* parameter shapes ... reference probnmn/models/nmn.py:67-115

The one deliberate deviation: ``SameModule`` uses integer floor division for the arg-max row
(``idx // size``), which is what the reference's pinned torch 1.4 computes for
``the_idx[0,0,0,0] / size``; under torch >= 1.5 the literal expression is a float and raises.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

PLACEHOLDERS = {"@@PADDING@@", "@@UNKNOWN@@", "@start@", "@end@", "unique"}
BINARY = {"intersect", "union", "less_than", "greater_than"}
INVALID_LOSS = 3.33  # reference nmn.py:260,269


def module_kind(token: str) -> Optional[str]:
    """reference nmn.py:87-111 (test order preserved)."""
    if token in PLACEHOLDERS:
        return None
    if token == "scene":
        return "scene"
    if token == "intersect":
        return "and"
    if token == "union":
        return "or"
    if "equal" in token or token in {"less_than", "greater_than"}:
        return "comparison"
    if "query" in token or token in {"exist", "count"}:
        return "query"
    if "relate" in token:
        return "relate"
    if "same" in token:
        return "same"
    return "attention"


def nmn_param_shapes(
    program_tokens: Sequence[str],
    image_feature_size: Tuple[int, int, int] = (1024, 14, 14),
    module_channels: int = 128,
    class_projection_channels: int = 1024,
    classifier_linear_size: int = 1024,
    num_answers: int = 28,
) -> Dict[str, Tuple[int, ...]]:
    """Every parameter of the reference network by state_dict key (nmn.py:67-115)."""
    C, H, W = image_feature_size
    D = module_channels
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k):
        s[name + ".weight"] = (cout, cin, k, k)
        s[name + ".bias"] = (cout,)

    conv("stem.0", D, C, 3)
    conv("stem.2", D, D, 3)
    conv("classifier.0", class_projection_channels, D, 1)
    s["classifier.4.weight"] = (classifier_linear_size, class_projection_channels * H * W // 4)
    s["classifier.4.bias"] = (classifier_linear_size,)
    s["classifier.6.weight"] = (num_answers, classifier_linear_size)
    s["classifier.6.bias"] = (num_answers,)
    for tok in program_tokens:
        kind = module_kind(tok)
        if kind in ("attention", "query"):
            conv(tok + ".conv1", D, D, 3)
            conv(tok + ".conv2", D, D, 3)
            if kind == "attention":
                conv(tok + ".conv3", 1, D, 1)
        elif kind == "relate":
            for i in range(1, 6):
                conv(tok + ".conv%d" % i, D, D, 3)
            conv(tok + ".conv6", 1, D, 1)
        elif kind == "same":
            conv(tok + ".conv", 1, D + 1, 1)
        elif kind == "comparison":
            conv(tok + ".projection", D, 2 * D, 1)
            conv(tok + ".conv1", D, D, 3)
            conv(tok + ".conv2", D, D, 3)
    return s


# ---- the seven modules ---------------------------------------------------------------------
def _conv(sd, name, x, padding=0, dilation=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding, dilation=dilation)


def and_module(a, b):  # nmn_modules.py:25-27
    return torch.min(a, b)


def or_module(a, b):  # nmn_modules.py:43-45
    return torch.max(a, b)


def attention_module(sd, tok, feats, attn):  # nmn_modules.py:82-87
    dim = sd[tok + ".conv1.weight"].shape[1]
    x = feats * attn.repeat(1, dim, 1, 1)
    x = F.relu(_conv(sd, tok + ".conv1", x, 1))
    x = F.relu(_conv(sd, tok + ".conv2", x, 1))
    return torch.sigmoid(_conv(sd, tok + ".conv3", x))


def query_module(sd, tok, feats, attn):  # nmn_modules.py:119-123
    dim = sd[tok + ".conv1.weight"].shape[1]
    x = feats * attn.repeat(1, dim, 1, 1)
    x = F.relu(_conv(sd, tok + ".conv1", x, 1))
    return F.relu(_conv(sd, tok + ".conv2", x, 1))


def relate_module(sd, tok, feats, attn):  # nmn_modules.py:160-168
    dim = sd[tok + ".conv1.weight"].shape[1]
    x = feats * attn.repeat(1, dim, 1, 1)
    for i, d in enumerate((1, 2, 4, 8, 1), start=1):
        x = F.relu(_conv(sd, tok + ".conv%d" % i, x, padding=d, dilation=d))
    return torch.sigmoid(_conv(sd, tok + ".conv6", x))


def same_module(sd, tok, feats, attn):  # nmn_modules.py:200-208 (torch-1.4 integer division)
    size = attn.size(2)
    _, idx = F.max_pool2d(attn, size, return_indices=True)
    flat = idx[0, 0, 0, 0]
    row = torch.div(flat, size, rounding_mode="floor")
    col = flat % size
    picked = feats.index_select(2, row.view(1)).index_select(3, col.view(1))
    x = feats * picked.repeat(1, 1, size, size)
    x = torch.cat([x, attn], dim=1)
    return torch.sigmoid(_conv(sd, tok + ".conv", x))


def comparison_module(sd, tok, in1, in2):  # nmn_modules.py:240-244
    x = torch.cat([in1, in2], 1)
    x = F.relu(_conv(sd, tok + ".projection", x))
    x = F.relu(_conv(sd, tok + ".conv1", x, 1))
    return F.relu(_conv(sd, tok + ".conv2", x, 1))


_UNARY = {
    "attention": attention_module,
    "query": query_module,
    "relate": relate_module,
    "same": same_module,
}


# ---- network ---------------------------------------------------------------------------------
def stem(sd, features):  # nmn.py:67-72,183
    x = F.relu(_conv(sd, "stem.0", features, 1))
    return F.relu(_conv(sd, "stem.2", x, 1))


def classifier(sd, x):  # nmn.py:75-83
    x = F.relu(_conv(sd, "classifier.0", x))
    x = F.max_pool2d(x, kernel_size=2, stride=2)
    x = x.reshape(x.size(0), -1)
    x = F.relu(F.linear(x, sd["classifier.4.weight"], sd["classifier.4.bias"]))
    return F.linear(x, sd["classifier.6.weight"], sd["classifier.6.bias"])


def execute_program(sd, index_to_token, feat_input, program: Sequence[int], module_channels: int):
    """One example; right-to-left, one side register (nmn.py:199-238).  Returns (output, valid)."""
    output = feat_input
    saved_output = None
    try:
        for i in reversed([int(t) for t in program]):
            tok = index_to_token[i]
            kind = module_kind(tok)
            if kind is None:
                continue
            if kind == "scene":
                saved_output = output
                output = torch.ones_like(feat_input)[:, :1, :, :]
                continue
            if kind == "and":
                output = and_module(output, saved_output)
            elif kind == "or":
                output = or_module(output, saved_output)
            elif kind == "comparison":
                output = comparison_module(sd, tok, output, saved_output)
            else:
                output = _UNARY[kind](sd, tok, feat_input, output)
        if output.size(1) != module_channels:
            raise ValueError("program must end with an encoding")
        return output, 1
    except Exception:  # the reference uses a bare except (nmn.py:235)
        return torch.zeros_like(feat_input), 0


def nmn_forward(
    sd: Dict[str, torch.Tensor],
    index_to_token: Dict[int, str],
    features: torch.Tensor,
    programs: torch.Tensor,
    answers: Optional[torch.Tensor] = None,
    unknown_answer_index: int = 28,
):
    """reference nmn.py:139-275.  Returns dict(predictions, loss, valid, logits, final)."""
    feat_volume = stem(sd, features)
    D = feat_volume.size(1)
    outs: List[torch.Tensor] = []
    valid: List[int] = []
    progs = programs.cpu().numpy()
    for n in range(feat_volume.size(0)):
        out, ok = execute_program(sd, index_to_token, feat_volume[n].unsqueeze(0), progs[n], D)
        outs.append(out)
        valid.append(ok)
    final = torch.cat(outs, 0)
    logits = classifier(sd, final)
    logprobs = F.log_softmax(logits, dim=-1)
    best_logprob, predictions = torch.max(logprobs, dim=1)
    valid_t = torch.tensor(valid)
    predictions = predictions.clone()
    predictions[valid_t == 0] = unknown_answer_index
    if answers is not None:
        loss = F.cross_entropy(logits, answers, reduction="none")
    else:
        loss = -best_logprob
    loss = loss.clone()
    loss[valid_t == 0] = INVALID_LOSS
    return {
        "predictions": predictions,
        "loss": loss,
        "valid": valid_t,
        "logits": logits,
        "final": final,
        "stem": feat_volume,
    }
