"""Token <-> index tables for the three CLEVR namespaces.

The reference takes an ``allennlp.data.Vocabulary`` (reference: probnmn/models/nmn.py:5,
probnmn/modules/seq2seq_base.py:4) and uses five of its methods.  AllenNLP is not part of
this build, so this class offers exactly that subset with the same call signatures:

    get_index_to_token_vocabulary(namespace) -> {int: str}
    get_token_to_index_vocabulary(namespace) -> {str: int}
    get_token_from_index(index, namespace)   -> str
    get_token_index(token, namespace)        -> int      (OOV -> @@UNKNOWN@@ index)
    get_vocab_size(namespace)                -> int
    Vocabulary.from_files(directory)

Index conventions (reference: scripts/preprocess/build_vocabulary.py:114-149):
padded namespaces ("programs", "questions") put @@PADDING@@ at 0 and then follow file order, so
@@UNKNOWN@@=1, @start@=2, @end@=3; the non-padded "answers" namespace follows file order from 0
with @@UNKNOWN@@ as its last entry.
"""
import os
from typing import Dict, Iterable, List

PADDING = "@@PADDING@@"
UNKNOWN = "@@UNKNOWN@@"
START = "@start@"
END = "@end@"

_COLORS = ["blue", "brown", "cyan", "gray", "green", "purple", "red", "yellow"]
_MATERIALS = ["metal", "rubber"]
_SHAPES = ["cube", "cylinder", "sphere"]
_SIZES = ["large", "small"]
_RELATIONS = ["behind", "front", "left", "right"]


def clevr_program_tokens() -> List[str]:
    """The 40 CLEVR v1.0 function tokens in sorted order."""
    toks = ["count", "exist", "greater_than", "less_than", "intersect", "union", "scene", "unique"]
    toks += ["equal_" + k for k in ("color", "integer", "material", "shape", "size")]
    toks += ["query_" + k for k in ("color", "material", "shape", "size")]
    toks += ["same_" + k for k in ("color", "material", "shape", "size")]
    toks += ["filter_color[%s]" % v for v in _COLORS]
    toks += ["filter_material[%s]" % v for v in _MATERIALS]
    toks += ["filter_shape[%s]" % v for v in _SHAPES]
    toks += ["filter_size[%s]" % v for v in _SIZES]
    toks += ["relate[%s]" % v for v in _RELATIONS]
    return sorted(toks)


def clevr_answer_tokens() -> List[str]:
    """The 28 CLEVR answers in sorted order."""
    toks = [str(i) for i in range(11)]
    toks += _COLORS + _MATERIALS + _SHAPES + _SIZES + ["yes", "no"]
    return sorted(toks)


class Vocabulary:
    def __init__(self, namespaces: Dict[str, List[str]]):
        self._itos: Dict[str, Dict[int, str]] = {}
        self._stoi: Dict[str, Dict[str, int]] = {}
        for name, tokens in namespaces.items():
            self._itos[name] = dict(enumerate(tokens))
            self._stoi[name] = {t: i for i, t in enumerate(tokens)}

    # ---- constructors -------------------------------------------------------------------
    @classmethod
    def from_token_lists(
        cls, programs: Iterable[str], questions: Iterable[str], answers: Iterable[str]
    ) -> "Vocabulary":
        special = [PADDING, UNKNOWN, START, END]
        return cls(
            {
                "programs": special + list(programs),
                "questions": special + list(questions),
                "answers": list(answers) + [UNKNOWN],
            }
        )

    @classmethod
    def clevr(cls, num_question_tokens: int = 96) -> "Vocabulary":
        """CLEVR-shaped vocabulary without the dataset: the real 40 program tokens and 28
        answers, and ``num_question_tokens`` placeholder question words (the true question
        vocabulary cannot be derived without the data; its size is a parameter)."""
        words = ["w%03d" % i for i in range(num_question_tokens)]
        return cls.from_token_lists(clevr_program_tokens(), words, clevr_answer_tokens())

    @classmethod
    def from_files(cls, directory: str) -> "Vocabulary":
        non_padded = set()
        npn = os.path.join(directory, "non_padded_namespaces.txt")
        if os.path.exists(npn):
            with open(npn) as f:
                non_padded = {line.strip() for line in f if line.strip()}
        namespaces: Dict[str, List[str]] = {}
        for fname in sorted(os.listdir(directory)):
            if not fname.endswith(".txt") or fname == "non_padded_namespaces.txt":
                continue
            name = fname[: -len(".txt")]
            with open(os.path.join(directory, fname)) as f:
                tokens = [line.rstrip("\n") for line in f if line.rstrip("\n") != ""]
            namespaces[name] = tokens if name in non_padded else [PADDING] + tokens
        return cls(namespaces)

    def save_to_files(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        non_padded = []
        for name, itos in self._itos.items():
            tokens = [itos[i] for i in range(len(itos))]
            if tokens and tokens[0] == PADDING:
                tokens = tokens[1:]
            else:
                non_padded.append(name)
            with open(os.path.join(directory, name + ".txt"), "w") as f:
                f.writelines(t + "\n" for t in tokens)
        with open(os.path.join(directory, "non_padded_namespaces.txt"), "w") as f:
            f.write("\n".join(non_padded))

    # ---- the AllenNLP subset ------------------------------------------------------------
    def get_index_to_token_vocabulary(self, namespace: str = "tokens") -> Dict[int, str]:
        return self._itos[namespace]

    def get_token_to_index_vocabulary(self, namespace: str = "tokens") -> Dict[str, int]:
        return self._stoi[namespace]

    def get_token_from_index(self, index: int, namespace: str = "tokens") -> str:
        return self._itos[namespace][int(index)]

    def get_token_index(self, token: str, namespace: str = "tokens") -> int:
        table = self._stoi[namespace]
        if token in table:
            return table[token]
        return table[UNKNOWN]

    def get_vocab_size(self, namespace: str = "tokens") -> int:
        return len(self._itos[namespace])
