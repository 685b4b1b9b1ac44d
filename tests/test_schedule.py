"""Host-side scheduler invariants (no GPU): every primitive's inputs are produced at a lower level,
every buffer a launch writes is written by exactly one item of that launch, arena blocks do not
overlap, and records are bit-compatible with the C structs."""
import os
import numpy as np

from probnmn import _hip
from probnmn.data.synthetic import synthetic_batch
from probnmn.runtime import program_compiler as pc
from probnmn.runtime.schedule import BatchScheduler, Buffers, WeightTables
from probnmn.vocabulary import Vocabulary

from fixtures import VALIDITY_CASES, encode_programs

DT = {"conv": _hip.CONV_ITEM, "dot": _hip.DOT1_ITEM, "same": _hip.SAME_ITEM, "minmax": _hip.MINMAX_ITEM,
      "maskbwd": _hip.MASKBWD_ITEM, "wgrad_item": _hip.WGRAD_ITEM, "wgrad_job": _hip.WGRAD_JOB}
BUF = Buffers(params=1 << 40, grads=2 << 40, wt=3 << 40, act=4 << 40, gact=5 << 40, feat=6 << 40,
              gfeat=7 << 40, final=8 << 40, gfinal=9 << 40, ones=10 << 40)
HW, C = 196, 128


def _scheduler():
    v = Vocabulary.clevr()
    comp = pc.ProgramCompiler(v.get_index_to_token_vocabulary("programs"))
    V = 44
    tb = WeightTables(np.arange(V * 6).reshape(V, 6) * 200000, np.arange(V * 6).reshape(V, 6) * 200000 + 150000,
                      np.arange(V * 6).reshape(V, 6) * 200000, np.arange(V) * 1000 + 10 ** 8, np.arange(V) * 1000 + 10 ** 8 + 500)
    return v, comp, BatchScheduler(HW, C, tb, DT)


def _check_plan(plan, n_examples):
    written_at = {}  # address -> forward level that produces it
    for l in plan.forward:
        rec = plan.records[l.kind][l.begin:l.end]
        outs = rec["out"]
        assert len(np.unique(outs)) == len(outs), "two items of one launch write the same buffer"
        ins = [rec[f] for f in {"conv": ("in", "mask"), "proj": ("in", "in2"), "dot": ("in",),
                                "same": ("feats", "attn"), "minmax": ("a", "b")}[l.kind]]
        for arr in ins:
            for a in arr:
                a = int(a)
                if a == 0 or a == BUF.ones or (BUF.feat <= a < BUF.feat + (1 << 39)):
                    continue
                assert a in written_at and written_at[a] < l.level, (l.kind, l.level, hex(a))
        for o in outs:
            written_at[int(o)] = l.level
    # forward and backward cover the same primitives
    nf = sum(l.end - l.begin for l in plan.forward)
    assert nf == plan.n_prims
    nb = sum(l.end - l.begin for ph in plan.backward for l in ph if l.kind in ("dgrad", "dot_bwd", "same_bwd", "minmax_bwd"))
    nb += sum(l.end - l.begin for ph in plan.backward for l in ph if l.kind == "pdgrad") // 2
    assert nb == plan.n_prims
    # accumulate-mode launches never write one buffer twice
    for ph in plan.backward:
        for l in ph:
            if l.kind in ("pdgrad", "dgrad"):
                outs = plan.records[l.kind][l.begin:l.end]["out"]
                outs = outs[outs != 0]  # masked convs add into dFEAT / d(attention) with atomics instead
                assert len(np.unique(outs)) == len(outs)
    # arena addresses stay inside the arena
    for k in ("conv", "proj", "dot"):
        o = plan.records[k]["out"].astype(np.int64)
        inside = (o >= BUF.act) & (o < BUF.act + plan.arena_floats * 4)
        final = (o >= BUF.final) & (o < BUF.final + n_examples * HW * C * 4)
        assert np.all(inside | final)
    # weight-gradient jobs partition the items, each job within one weight
    for k, key in (("wg3", None), ("wgp", None)):
        jobs, items = plan.wgrad_jobs[k], plan.records[k]
        covered = np.zeros(len(items), int)
        for j in jobs:
            covered[j["item_begin"]:j["item_end"]] += 1
            assert 0 < j["item_end"] - j["item_begin"] <= 8
        assert np.all(covered == 1)


def test_golden_programs_plan():
    v, comp, s = _scheduler()
    progs = encode_programs(VALIDITY_CASES, v.get_token_to_index_vocabulary("programs")).numpy()
    compiled = comp.compile_batch(progs)
    plan = s.plan(compiled, BUF)
    assert plan.arena_floats == s.arena_floats(compiled)
    _check_plan(plan, len(compiled))
    # empty / placeholder-only programs return the stem output itself
    assert set(plan.feat_result_examples.tolist()) == {0, 1}
    # each case alone (ragged, single-structure batches; some have no primitives at all)
    for case in VALIDITY_CASES:
        one = comp.compile_batch(encode_programs([case, case], v.get_token_to_index_vocabulary("programs")).numpy())
        _check_plan(s.plan(one, BUF), 2)


def test_synthetic_batch_plan_and_counts():
    v, comp, s = _scheduler()
    batch = synthetic_batch(v, 256, seed=1000, with_image=False)
    compiled = comp.compile_batch(batch["program"].numpy())
    assert all(p.valid for p in compiled)
    plan = s.plan(compiled, BUF)
    _check_plan(plan, 256)
    n3 = len(plan.records["conv"])
    assert len(plan.records["dgrad"]) == n3 == len(plan.records["wg3"])
    assert len(plan.records["pdgrad"]) == 2 * len(plan.records["proj"])
    # the two data gradients of a level's projections share ONE launch (they add into different values' gradients:
    # _check_plan holds that no launch writes a buffer twice)
    for phase in plan.backward:
        assert sum(1 for l in phase if l.kind == "pdgrad") <= 1
    # 3x3 conv count per template: T1 6, T2 13, T3 18, T4 25, T5 12, T6 20, T7 6, T8 17
    assert 256 * 6 <= n3 <= 256 * 25
    depth = max(l.level for l in plan.forward)
    assert depth <= 40


def test_invalid_batch_is_empty_plan():
    v, comp, s = _scheduler()
    progs = encode_programs(["scene", "count"], v.get_token_to_index_vocabulary("programs")).numpy()
    plan = s.plan(comp.compile_batch(progs), BUF)
    assert plan.n_prims == 0 and plan.arena_floats == 0 and not plan.forward and not plan.backward


def _same_plan(a, b):
    assert a.n_prims == b.n_prims and a.arena_floats == b.arena_floats
    assert np.array_equal(a.feat_result_examples, b.feat_result_examples)
    assert set(a.records) == set(b.records)
    for k in a.records:
        assert a.records[k].dtype == b.records[k].dtype and a.records[k].shape == b.records[k].shape, k
        assert a.records[k].tobytes() == b.records[k].tobytes(), k
    for k in ("wg3", "wgp"):
        assert a.wgrad_jobs[k].tobytes() == b.wgrad_jobs[k].tobytes(), k
    assert a.forward == b.forward and a.backward == b.backward
    assert [tuple(int(x) for x in g) for g in (a.wgrad_groups or [])] == [tuple(int(x) for x in g) for g in (b.wgrad_groups or [])]


def test_library_planner_equals_the_numpy_planner(monkeypatch):
    """``pnmn_plan_batch`` (csrc/host_plan.hip) against the whole-array numpy formulation: every record of every
    launch bit for bit, the launch order, the weight-gradient jobs and groups -- on the validity cases (ragged,
    invalid, placeholder-only programs), on synthetic CLEVR batches incl. the deep 40-token shapes, with and
    without the fused mask backward, several weight-gradient groups, and without the weight sort."""
    from probnmn.data.synthetic import deep_template_program  # noqa: F401  (the deep shapes are part of deep=True batches)

    v, comp, s = _scheduler()
    t2i = v.get_token_to_index_vocabulary("programs")
    batches = [comp.compile_batch(encode_programs(VALIDITY_CASES, t2i).numpy())]
    for seed, n, deep in ((1, 1, False), (2, 65, False), (3, 300, False), (4, 64, True)):
        kw = {"deep": True, "program_length": 40} if deep else {}
        b = synthetic_batch(v, n, seed=seed, with_image=False, **kw)
        batches.append(comp.compile_batch(b["program"].numpy()))
    for compiled in batches:
        _same_plan(s.plan(compiled, BUF), s.plan_numpy(compiled, BUF))
    mixed = batches[2]
    s.fuse_mask_bwd = False
    _same_plan(s.plan(mixed, BUF), s.plan_numpy(mixed, BUF))
    s.fuse_mask_bwd = True
    s.sole_writer_rmw = False
    _same_plan(s.plan(mixed, BUF), s.plan_numpy(mixed, BUF))
    s.sole_writer_rmw = True
    s.wgrad_groups, s.wgrad_chunk = 3, 2
    _same_plan(s.plan(mixed, BUF), s.plan_numpy(mixed, BUF))
    s.sort_by_weight = False
    _same_plan(s.plan(mixed, BUF), s.plan_numpy(mixed, BUF))


def test_the_numpy_planner_is_not_part_of_the_product():
    """runtime/schedule.py is the specification the library's planner is compared with; the engine configures the
    library's planner through PlannerConfig alone (one home for the configuration)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import probnmn.runtime.engine, probnmn.models.nmn; "
            "assert 'probnmn.runtime.schedule' not in sys.modules" % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "probnmn-clevr_amd"))
    subprocess.check_call([sys.executable, "-c", code])
