"""The library's trunk planner (csrc/host_trunk.hip: programs -> launched module programs in one call) against the
Python planner it replaces on the critical path (program_compiler + schedule.py, itself checked against the numpy
specification in test_schedule.py) -- no GPU: `launch = 0` plans into host memory.  Same validity, same templates
(hence same arena layout), same records word for word, same launch order forward and backward."""
import numpy as np

from probnmn import _hip
from probnmn.data.synthetic import synthetic_batch

from fixtures import VALIDITY_CASES, encode_programs
from test_schedule import BUF, C, HW, _scheduler

FAKE_BASE = 0x10000
RECORD_DTYPE = {"conv": _hip.CONV_ITEM, "proj": _hip.CONV_ITEM, "dgrad": _hip.CONV_ITEM, "pdgrad": _hip.CONV_ITEM,
                "dot": _hip.DOT1_ITEM, "same": _hip.SAME_ITEM, "minmax": _hip.MINMAX_ITEM, "maskbwd": _hip.MASKBWD_ITEM,
                "wg3": _hip.WGRAD_ITEM, "wgp": _hip.WGRAD_ITEM}


def _planner(s, comp):
    t = s.tables
    keep = [np.ascontiguousarray(comp.kinds, dtype=np.int32)] + [np.ascontiguousarray(a, dtype=np.int64)
                                                                 for a in (t.w3, t.b3, t.wt3, t.dotw, t.dotb)]
    cfg = np.zeros(1, _hip.TRUNK_CONFIG)
    cfg[0] = tuple(a.ctypes.data for a in keep) + (keep[0].size, C, 14, 14, s.wgrad_chunk, s.wgrad_groups,
                                                    int(s.fuse_mask_bwd), int(s.sole_writer_rmw), 1, 0)
    out = np.zeros(1, np.uint64)
    _hip.check(_hip.lib().pnmn_trunk_planner_create(cfg.ctypes.data, out.ctypes.data), "create")
    return int(out[0])


def _rows(spec):
    lst = _hip.LaunchList()
    for op, n, a in spec:
        lst.add(op, n, a)
    return np.array(lst._rows, dtype=np.uint64).reshape(-1, 8)


def _run(planner, programs, capacity=1 << 40, conv_cus=0):
    programs = np.ascontiguousarray(programs, dtype=np.int64)
    B = programs.shape[0]
    fwd_tail, bwd_head, bwd_tail = _rows([(_hip.OP_MAXPOOL_FWD, B, 111)]), _rows([(_hip.OP_ZERO, 0, 222), (_hip.OP_MAXPOOL_BWD, B, 333)]), _rows([(_hip.OP_WGRAD, 5, 444)])
    bwd = np.zeros((2048, 8), np.uint64)
    valid = np.zeros(B, np.uint8)
    io = np.zeros(1, _hip.TRUNK_IO)
    io[0] = (programs.ctypes.data, BUF.params, BUF.grads, BUF.wt, BUF.act, BUF.gact, BUF.feat, BUF.gfeat, BUF.final, BUF.gfinal,
             BUF.ones, capacity, fwd_tail.ctypes.data, bwd_head.ctypes.data, bwd_tail.ctypes.data, bwd.ctypes.data,
             valid.ctypes.data, 0, B, programs.shape[1], 1, 2, 1, bwd.shape[0], 1, 0, 0, 0, 0, 0, 0, 0, 0, conv_cus, 0, 0, 0, 0, (0, 0, 0, 0))
    rc = _hip.lib().pnmn_trunk_plan_and_launch(planner, io.ctypes.data, None)
    out = io[0]
    if rc != 0:
        return rc, out, None, None, valid, None
    fwd = np.zeros((2048, 8), np.uint64)
    n_fwd = _hip.lib().pnmn_trunk_last_forward(planner, fwd.ctypes.data, fwd.shape[0])
    assert n_fwd == out["n_fwd"]
    words = np.zeros(4 << 20, np.uint64)
    nbytes = _hip.lib().pnmn_trunk_last_records_bytes(planner, words.ctypes.data, words.size)
    assert 0 <= nbytes <= words.size * 8
    return rc, out, fwd[:n_fwd].view(_hip.LAUNCH).reshape(-1), bwd[: out["n_bwd"]].view(_hip.LAUNCH).reshape(-1), valid, words


def _records(words, launch, kind):
    dt = RECORD_DTYPE[kind]
    off = int(launch["a"]) - FAKE_BASE
    assert off >= 0 and off % 8 == 0
    return words[off // 8: off // 8 + int(launch["n"]) * dt.itemsize // 8].view(dt)


FWD = {"minmax": (_hip.OP_MINMAX_FWD, (HW, C)), "same": (_hip.OP_SAME_FWD, (HW,)), "dot": (_hip.OP_DOT_FWD, (HW,)),
       "proj": (_hip.OP_CONV, (14, 14, 2, 1, C, C, 1, 1)), "conv": (_hip.OP_CONV, (14, 14, 1, 9, C, C, 1, 1))}
BWD = {"minmax_bwd": (_hip.OP_MINMAX_BWD, (HW, C), "minmax"), "same_bwd": (_hip.OP_SAME_BWD, (HW,), "same"),
       "dot_bwd": (_hip.OP_DOT_BWD, (HW,), "dot"), "pdgrad": (_hip.OP_CONV, (14, 14, 1, 1, C, C, 1, 0), "pdgrad"),
       "dgrad": (_hip.OP_CONV, (14, 14, 1, 9, C, C, 1, 0), "dgrad"), "maskbwd": (_hip.OP_MASK_BWD, (HW,), "maskbwd")}


def _same_launch(got, op, n, p):
    assert int(got["op"]) == op and int(got["n"]) == n, (got, op, n)
    assert tuple(int(v) for v in got["p"][: len(p)]) == tuple(p) and not any(got["p"][len(p):])


def _compare(planner, comp, s, programs):
    compiled = comp.compile_batch(programs)
    plan = s.plan(compiled, BUF)
    rc, out, fwd, bwd, valid, words = _run(planner, programs)
    assert rc == 0
    assert valid.tolist() == [int(p.valid) for p in compiled]
    assert out["n_prims"] == plan.n_prims and out["arena_floats"] == plan.arena_floats
    assert out["n_invalid"] == sum(1 for p in compiled if not p.valid)
    assert out["n_feat_result"] == plan.feat_result_examples.size
    # ---- forward: [SET_ROWS] + the level-ordered launches + the caller's tail
    i = 0
    n_rows = int(out["n_invalid"]) + int(out["n_feat_result"])
    if n_rows:
        _same_launch(fwd[0], _hip.OP_SET_ROWS, n_rows, ())
        items = words[(int(fwd[0]["a"]) - FAKE_BASE) // 8:][: 3 * n_rows].view(_hip.AXPY_ITEM)
        zeroed = sorted(int(d - BUF.final) // (HW * C * 4) for sname, d in zip(items["src"], items["dst"]) if sname == 0)
        assert zeroed == [e for e, p in enumerate(compiled) if not p.valid]
        copied = [(int(a - BUF.feat), int(d - BUF.final)) for a, d in zip(items["src"], items["dst"]) if a != 0]
        assert copied == [(int(e) * HW * C * 4,) * 2 for e in plan.feat_result_examples.tolist()]
        assert set(items["n"].tolist()) == {HW * C}
        i = 1
    for l in plan.forward:
        op, p = FWD[l.kind]
        _same_launch(fwd[i], op, l.end - l.begin, p)
        got = _records(words, fwd[i], l.kind)
        assert got.tobytes() == plan.records[l.kind][l.begin:l.end].tobytes(), (l.kind, l.level)
        i += 1
    _same_launch(fwd[i], _hip.OP_MAXPOOL_FWD, len(compiled), ())
    assert i + 1 == len(fwd)
    # ---- backward: head + ZERO(gact block) + [ACCUMULATE] + phases + deferred weight gradients + tail
    assert int(bwd[0]["op"]) == _hip.OP_ZERO and int(bwd[1]["op"]) == _hip.OP_MAXPOOL_BWD
    i = 2
    if plan.arena_floats:
        assert int(bwd[i]["op"]) == _hip.OP_ZERO and int(bwd[i]["a"]) == BUF.gact and int(bwd[i]["b"]) == plan.arena_floats * 4
        i += 1
    if plan.feat_result_examples.size:
        _same_launch(bwd[i], _hip.OP_ACCUMULATE, plan.feat_result_examples.size, ())
        items = words[(int(bwd[i]["a"]) - FAKE_BASE) // 8:][: 3 * plan.feat_result_examples.size].view(_hip.AXPY_ITEM)
        moved = [(int(a - BUF.gfinal), int(d - BUF.gfeat)) for a, d in zip(items["src"], items["dst"])]
        assert moved == [(int(e) * HW * C * 4,) * 2 for e in plan.feat_result_examples.tolist()]
        i += 1
    for phase in plan.backward:
        for l in phase:
            op, p, rk = BWD[l.kind]
            _same_launch(bwd[i], op, l.end - l.begin, p)
            assert _records(words, bwd[i], rk).tobytes() == plan.records[rk][l.begin:l.end].tobytes(), (l.kind, l.level)
            i += 1
    if s.fuse_mask_bwd == 2 and len(plan.records["maskbwd"]):  # deferred d(feats) of the masked convs: one gather
        n_mb = len(plan.records["maskbwd"])
        _same_launch(bwd[i], _hip.OP_FEAT_GATHER, n_mb, (len(compiled), HW))
        assert int(bwd[i]["b"]) == BUF.gfeat
        got = words[(int(bwd[i]["a"]) - FAKE_BASE) // 8:][: n_mb * 5].view(_hip.MASKBWD_ITEM)
        assert got.tobytes() == plan.records["maskbwd"].tobytes()
        assert np.all(np.diff(got["dfeats"].astype(np.int64)) >= 0)  # sorted by the map they add into
        i += 1
    for key, ntaps, cin_blocks in (("wg3", 9, 1), ("wgp", 1, 2)):
        jobs = plan.wgrad_jobs[key]
        if len(jobs):
            _same_launch(bwd[i], _hip.OP_WGRAD, len(jobs), (14, 14, ntaps, cin_blocks, 1, C, C, 0))
            items = words[(int(bwd[i]["a"]) - FAKE_BASE) // 8:][: len(plan.records[key]) * 6].view(_hip.WGRAD_ITEM)
            assert items.tobytes() == plan.records[key].tobytes()
            got_jobs = words[(int(bwd[i]["b"]) - FAKE_BASE) // 8:][: len(jobs) * 3].view(_hip.WGRAD_JOB)
            assert got_jobs.tobytes() == jobs.tobytes()
            i += 1
    assert out["bwd_piece_cut"] == i
    _same_launch(bwd[i], _hip.OP_WGRAD, 5, ())
    assert i + 1 == len(bwd)
    return plan


def test_native_planner_equals_the_python_planner_on_the_golden_programs():
    v, comp, s = _scheduler()
    planner = _planner(s, comp)
    progs = encode_programs(VALIDITY_CASES, v.get_token_to_index_vocabulary("programs")).numpy()
    plan = _compare(planner, comp, s, progs)
    assert plan.n_prims > 0 and plan.feat_result_examples.size > 0
    _hip.lib().pnmn_trunk_planner_destroy(planner)


def test_native_planner_equals_the_python_planner_on_synthetic_batches_and_keeps_its_caches():
    v, comp, s = _scheduler()
    planner = _planner(s, comp)
    for seed, n, deep in ((1, 64, False), (2, 257, False), (1, 64, False), (3, 40, True)):
        b = synthetic_batch(v, n, seed=seed, with_image=False, deep=deep)
        _compare(planner, comp, s, b["program"].numpy())
    _hip.lib().pnmn_trunk_planner_destroy(planner)


def test_native_planner_reports_a_too_small_arena_and_an_empty_batch():
    v, comp, s = _scheduler()
    planner = _planner(s, comp)
    b = synthetic_batch(v, 16, seed=4, with_image=False)["program"].numpy()
    rc, out, *_ = _run(planner, b, capacity=10)
    assert rc == _hip.EAGAIN and out["arena_floats"] > 10
    rc, out, fwd, bwd, valid, _ = _run(planner, np.zeros((3, 5), np.int64))  # three empty programs: result = the feature map
    assert rc == 0 and valid.tolist() == [1, 1, 1] and out["n_feat_result"] == 3 and out["n_prims"] == 0
    assert [int(l["op"]) for l in fwd] == [_hip.OP_SET_ROWS, _hip.OP_MAXPOOL_FWD]
    _hip.lib().pnmn_trunk_planner_destroy(planner)


def test_no_accumulating_launch_writes_one_map_twice():
    """the projections' two data gradients of a level share ONE launch whose items add into their target maps
    with a non-atomic read-modify-write.  Invariant of every plan: inside one such launch all target maps are distinct --
    checked on synthetic batches (shallow and deep) and on hand-built programs in which ``saved_output`` feeds two
    comparisons (the second use lies a level deeper) and in which a comparison takes one value for both operands."""
    v, comp, s = _scheduler()
    stoi = v.get_token_to_index_vocabulary("programs")

    def row(*tokens, width=24):
        r = [stoi[t] for t in tokens]
        return r + [0] * (width - len(r))

    hand = np.array([
        # executed right to left: scene saves FEAT ... equal_size(q, saved) -> equal_color(that, saved): `saved` is read twice
        row("equal_color", "equal_size", "query_size", "filter_shape[cube]", "scene", "query_color", "filter_color[red]", "scene"),
        # a comparison of the saved value with itself is not expressible (output always moves on), but equal operands of two
        # DIFFERENT items are: two copies of one program
        row("equal_shape", "query_shape", "filter_size[large]", "scene", "query_shape", "filter_size[small]", "scene"),
        row("equal_shape", "query_shape", "filter_size[large]", "scene", "query_shape", "filter_size[small]", "scene"),
    ], dtype=np.int64)
    batches = [hand] + [np.asarray(synthetic_batch(v, n, seed=seed, with_image=False, deep=deep)["program"].numpy())
                        for seed, n, deep in ((5, 200, False), (6, 120, True), (7, 64, True))]
    checked = 0
    for programs in batches:
        compiled = comp.compile_batch(programs)
        plan = s.plan(compiled, BUF)
        for phase in plan.backward:
            for l in phase:
                if l.kind != "pdgrad":
                    continue
                rec = plan.records["pdgrad"][l.begin:l.end]
                targets = rec["out"]
                assert len(np.unique(targets)) == len(targets), (l.level, targets)
                checked += 1
    assert checked >= 4, checked
