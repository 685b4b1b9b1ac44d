import json

import pytest
import os

from probnmn.runtime import program_compiler as pc
from probnmn.vocabulary import Vocabulary

from fixtures import LONG_CASES, VALIDITY_CASES, namespaces


def test_vocabulary_matches_reference_layout():
    v = Vocabulary.clevr()
    ns = namespaces()
    assert [v.get_token_from_index(i, "programs") for i in range(44)] == ns["programs"]
    assert v.get_vocab_size("programs") == 44
    assert v.get_vocab_size("answers") == 29
    assert v.get_token_index("@@UNKNOWN@@", "answers") == 28
    assert [v.get_token_index(t, "programs") for t in ("@@PADDING@@", "@@UNKNOWN@@", "@start@", "@end@")] == [0, 1, 2, 3]
    assert v.get_token_index("no-such-token", "programs") == 1


def test_vocabulary_file_roundtrip(tmp_path):
    v = Vocabulary.clevr()
    v.save_to_files(str(tmp_path))
    w = Vocabulary.from_files(str(tmp_path))
    for ns in ("programs", "questions", "answers"):
        assert w.get_index_to_token_vocabulary(ns) == v.get_index_to_token_vocabulary(ns)


def test_module_table_counts():
    v = Vocabulary.clevr()
    kinds = [pc.classify_token(t) for t in v.get_token_to_index_vocabulary("programs")]
    # reference nmn.py:98-111 -> 15 Attention, 6 Query, 4 Relate, 4 Same, 7 Comparison
    assert kinds.count(pc.ATT) == 15
    assert kinds.count(pc.QUERY) == 6
    assert kinds.count(pc.REL) == 4
    assert kinds.count(pc.SAME) == 4
    assert kinds.count(pc.CMP) == 7
    assert kinds.count(pc.AND) == 1 and kinds.count(pc.OR) == 1 and kinds.count(pc.SCENE) == 1


@pytest.mark.parametrize("fixture,cases", [("nmn_validity.json", VALIDITY_CASES),
                                           ("nmn_validity_28.json", VALIDITY_CASES + LONG_CASES)])
def test_validity_matches_reference_interpreter(golden_dir, fixture, cases):
    """The static rules reproduce the reference's try/except verdict on every golden case (the second
    table: the reference at 28x28 maps, with programs of up to 40 tokens -- BASELINE config 5)."""
    with open(os.path.join(golden_dir, fixture)) as f:
        table = json.load(f)
    v = Vocabulary.clevr()
    comp = pc.ProgramCompiler(v.get_index_to_token_vocabulary("programs"), module_channels=8)
    assert set(table) == set(cases)
    rows = []
    for case, valid in table.items():
        ids = [v.get_token_index(t, "programs") for t in case.split()]
        got = comp.compile(ids + [0] * 3)
        assert got.valid == bool(valid), case
        rows.append(ids + [0] * (40 - len(ids)))
    import numpy as np

    batch = comp.compile_batch(np.asarray(rows, dtype=np.int64))  # the library's batch compiler, length 40
    assert [p.valid for p in batch] == [bool(x) for x in table.values()]


def test_operand_order_and_values():
    v = Vocabulary.clevr()
    comp = pc.ProgramCompiler(v.get_index_to_token_vocabulary("programs"))
    ids = [v.get_token_index(t, "programs") for t in
           "greater_than count filter_color[blue] scene count filter_size[small] scene".split()]
    p = comp.compile(ids)
    assert p.valid and [c.kind for c in p.calls] == [pc.ATT, pc.QUERY, pc.ATT, pc.QUERY, pc.CMP]
    cmp_call = p.calls[-1]
    # first operand = most recently computed chain (left subtree), second = saved register
    assert cmp_call.a == 5 and cmp_call.b == 3
    assert p.calls[0].a == pc.ONES and p.calls[0].b == pc.FEAT
    assert p.result == 6
    assert comp.compile([999]).valid is False
    assert comp.compile([]).valid and comp.compile([]).result == pc.FEAT


def test_batch_compiler_in_the_library_agrees_with_the_python_rules(golden_dir):
    """pnmn_compile_programs (host routine of the C-ABI library, what compile_batch uses for programs it
    has not seen) against ProgramCompiler._compile: validity, result value and every call, on the
    golden validity cases, on template programs and on random token soup."""
    import numpy as np

    from probnmn.data.synthetic import synthetic_batch

    v = Vocabulary.clevr()
    for channels in (8, 128):
        comp = pc.ProgramCompiler(v.get_index_to_token_vocabulary("programs"), module_channels=channels)
        rows = []
        for case in VALIDITY_CASES:  # (a list of space-separated programs)
            ids = [v.get_token_index(t, "programs") for t in case.split()]
            rows.append(ids + [0] * (26 - len(ids)))
        rows += synthetic_batch(v, 200, seed=5, with_image=False)["program"].tolist()
        rng = np.random.Generator(np.random.Philox(9))
        soup = rng.integers(0, 44, (600, 26))
        soup[rng.random((600, 26)) < 0.5] = 0          # mostly short programs
        soup[:50, 3] = 999                             # out-of-vocabulary token
        rows += soup.tolist()
        arr = np.asarray(rows, dtype=np.int64)
        got = comp.compile_batch(arr)
        n_valid = 0
        for row, g in zip(arr.tolist(), got):
            want = comp._compile(tuple(row))
            assert g.valid == want.valid and g.result == want.result, row
            assert g.calls == want.calls, row
            n_valid += g.valid
        assert 200 < n_valid < len(rows)  # both verdicts are exercised
        # second call: everything comes from the cache, same objects
        again = comp.compile_batch(arr)
        assert all(a is b for a, b in zip(got, again))
