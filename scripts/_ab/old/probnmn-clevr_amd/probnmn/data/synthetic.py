"""CLEVR-shaped synthetic batches (no dataset, no network): the batch dict schema of the
reference's datasets (reference: probnmn/data/datasets.py:91-95,137-142,222-228) with the shapes,
dtypes and value ranges of real preprocessed CLEVR, drawn from numpy's Philox generator.

  image        (B, C, H, W) float32   relu(N(0,1))  -- ResNet-101 stage-3 output is post-ReLU
  question     (B, 45)      int64     length ~ U{5..43}, tokens ~ U{4..Vq-1}, zero right-padded
  program      (B, 26)      int64     one of eight CLEVR template shapes,
                                      prefix order, random filter / relate / query arguments
  answer       (B,)         int64     U{0..27}
  supervision  (B,)         int64     Bernoulli(0.5)

``deep=True`` (BASELINE config 5, "program length <= 40 ... deeper module chains"): programs are drawn
from the eight CLEVR shapes and eight deeper ones of 16-40 tokens (up to ten hops, comparisons and
unions of multi-hop chains), zero-padded to 40.
"""
from typing import Dict, List, Optional

import numpy as np
import torch

_COLORS = ["blue", "brown", "cyan", "gray", "green", "purple", "red", "yellow"]
_FILTERS = (["filter_color[%s]" % c for c in _COLORS] + ["filter_material[metal]", "filter_material[rubber]"]
            + ["filter_shape[cube]", "filter_shape[cylinder]", "filter_shape[sphere]"]
            + ["filter_size[large]", "filter_size[small]"])
_RELATES = ["relate[behind]", "relate[front]", "relate[left]", "relate[right]"]
_QUERIES = ["query_color", "query_material", "query_shape", "query_size"]
_SAMES = ["same_color", "same_material", "same_shape", "same_size"]
_EQUALS = ["equal_color", "equal_material", "equal_shape", "equal_size"]
_INT_CMP = ["greater_than", "less_than", "equal_integer"]

NUM_TEMPLATES = 8


def template_program(t: int, rng: np.random.Generator) -> List[str]:
    F = lambda: _FILTERS[rng.integers(len(_FILTERS))]  # noqa: E731
    R = lambda: _RELATES[rng.integers(len(_RELATES))]  # noqa: E731
    Q = lambda: _QUERIES[rng.integers(len(_QUERIES))]  # noqa: E731
    if t == 0:  # count the objects matching two filters
        return [["count", "exist"][rng.integers(2)], F(), F(), "scene"]
    if t == 1:  # one hop
        return [Q(), "unique", F(), R(), "unique", F(), F(), "scene"]
    if t == 2:  # two hops
        return [Q(), "unique", F(), R(), "unique", F(), R(), "unique", F(), "scene"]
    if t == 3:  # three hops
        return [Q(), "unique", F(), R(), "unique", F(), R(), "unique", F(), R(), "unique", F(), "scene"]
    if t == 4:  # integer comparison of two counts
        return [_INT_CMP[rng.integers(3)], "count", F(), F(), "scene", "count", F(), "scene"]
    if t == 5:  # logical and / or of two one-hop chains
        return ["count", ["intersect", "union"][rng.integers(2)], F(), R(), "unique", F(), "scene",
                F(), R(), "unique", F(), "scene"]
    if t == 6:  # same-attribute
        return [Q(), "unique", _SAMES[rng.integers(4)], "unique", F(), F(), "scene"]
    if t == 7:  # attribute comparison of two objects
        k = rng.integers(4)
        return [_EQUALS[k], _QUERIES[k], "unique", F(), "scene", _QUERIES[k], "unique", F(), R(), "unique", F(), "scene"]
    raise ValueError(t)


NUM_DEEP_TEMPLATES = 8
DEEP_PROGRAM_LENGTH = 40


def deep_template_program(t: int, rng: np.random.Generator) -> List[str]:
    """Deeper module chains than any CLEVR template, at most 40 tokens (prefix order)."""
    F = lambda: _FILTERS[rng.integers(len(_FILTERS))]  # noqa: E731
    R = lambda: _RELATES[rng.integers(len(_RELATES))]  # noqa: E731
    Q = lambda: _QUERIES[rng.integers(len(_QUERIES))]  # noqa: E731

    def chain(hops: int, filters: int = 1) -> List[str]:
        """`filters` filters on the scene, then `hops` x (unique, relate, filter), outermost first."""
        out: List[str] = []
        for _ in range(hops):
            out += [F(), R(), "unique"]
        return out + [F() for _ in range(filters)] + ["scene"]

    if t == 0:
        return [Q(), "unique"] + chain(4)
    if t == 1:
        return [Q(), "unique"] + chain(6, 2)
    if t == 2:
        return [["count", "exist"][rng.integers(2)]] + chain(10, 2)
    if t == 3:  # attribute comparison of the ends of two three-hop chains
        k = rng.integers(4)
        return [_EQUALS[k], _QUERIES[k], "unique"] + chain(3) + [_QUERIES[k], "unique"] + chain(3, 2)
    if t == 4:  # integer comparison of two counts over five-hop chains (40 tokens)
        return [_INT_CMP[rng.integers(3)], "count"] + chain(5, 2) + ["count"] + chain(5, 3)
    if t == 5:  # and / or of two five-hop chains
        return ["count", ["intersect", "union"][rng.integers(2)]] + chain(5, 2) + chain(5)
    if t == 6:  # same-attribute between hops
        return [Q(), "unique"] + chain(2)[:-1] + [_SAMES[rng.integers(4)], "unique"] + chain(3, 2)
    if t == 7:  # a deep stack of filters over a two-hop chain
        return [["count", "exist"][rng.integers(2)]] + [F() for _ in range(20)] + chain(3, 6)
    raise ValueError(t)


def synthetic_batch(
    vocabulary,
    batch_size: int,
    image_feature_size=(1024, 14, 14),
    seed: int = 0,
    question_length: int = 45,
    program_length: Optional[int] = None,
    with_image: bool = True,
    device: Optional[torch.device] = None,
    deep: bool = False,
) -> Dict[str, torch.Tensor]:
    rng = np.random.Generator(np.random.Philox(seed))
    stoi = vocabulary.get_token_to_index_vocabulary("programs")
    vq = vocabulary.get_vocab_size("questions")
    num_answers = vocabulary.get_vocab_size("answers") - 1
    if program_length is None:
        program_length = DEEP_PROGRAM_LENGTH if deep else 26

    programs = np.zeros((batch_size, program_length), np.int64)
    templates = rng.integers(0, NUM_TEMPLATES + (NUM_DEEP_TEMPLATES if deep else 0), batch_size)
    for i, t in enumerate(templates):
        toks = template_program(int(t), rng) if t < NUM_TEMPLATES else deep_template_program(int(t) - NUM_TEMPLATES, rng)
        ids = [stoi[tok] for tok in toks]
        programs[i, : len(ids)] = ids
    questions = np.zeros((batch_size, question_length), np.int64)
    lengths = rng.integers(5, 44, batch_size)
    for i, n in enumerate(lengths):
        questions[i, :n] = rng.integers(4, vq, n)
    batch = {
        "question": torch.from_numpy(questions),
        "program": torch.from_numpy(programs),
        "answer": torch.from_numpy(rng.integers(0, num_answers, batch_size).astype(np.int64)),
        "supervision": torch.from_numpy((rng.random(batch_size) < 0.5).astype(np.int64)),
    }
    if with_image:
        c, h, w = image_feature_size
        img = rng.standard_normal((batch_size, c, h, w), dtype=np.float32)
        np.maximum(img, 0.0, out=img)
        batch["image"] = torch.from_numpy(img)
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch
