"""Execution engine of the Neural Module Network trunk on one MI355X.

trunk = stem -> per-example module programs -> classifier conv1x1 + ReLU + max-pool + flatten
(reference: probnmn/models/nmn.py:183-241 and :76-79).  One :class:`NMNEngine` belongs to one
``NeuralModuleNetwork``; it owns the parameter arena, the transposed-weight arena used by the data
gradients, the per-step activation / gradient arenas and the scheduler, and drives the kernels of
``libprobnmn_hip.so`` on torch's current stream.  All device memory is torch-allocated and reused
from step to step; the engine never synchronises with the host.

Forward and backward are both explicit kernel schedules (no autograd inside): backward walks the
forward levels in reverse; every weight gradient of the module convs is deferred into one big
grouped launch at the end, where all (example, conv) pairs that share a weight are accumulated by
the same workgroups.

The module programs are planned and launched by the library's trunk planner (csrc/host_trunk.hip:
programs -> compiled -> templates -> records -> upload -> launches in ONE call); this file puts the
program-independent launches around them (stem, classifier conv, pooling) and owns the buffers.
``probnmn.runtime.schedule`` is the planner's specification in numpy: tests/test_trunk_planner.py
compares the two word for word; nothing here launches from it.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from probnmn import _hip
from probnmn.runtime import program_compiler as pc
from probnmn.runtime.arena import ParamArena
from probnmn.runtime.planner_config import PlannerConfig, WeightTables

C = _hip.CHANNELS

_DT = {
    "conv": _hip.CONV_ITEM,
    "dot": _hip.DOT1_ITEM,
    "same": _hip.SAME_ITEM,
    "minmax": _hip.MINMAX_ITEM,
    "maskbwd": _hip.MASKBWD_ITEM,
    "wgrad_item": _hip.WGRAD_ITEM,
    "wgrad_job": _hip.WGRAD_JOB,
}


class _Pack:
    """Several numpy record arrays -> one device byte buffer (a single H2D copy per step)."""

    def __init__(self):
        self._chunks: List[Tuple[int, np.ndarray]] = []
        self._offsets: Dict[str, int] = {}
        self._itemsize: Dict[str, int] = {}
        self._size = 0
        self.device_buf: Optional[torch.Tensor] = None

    def add(self, name: str, rec: np.ndarray) -> None:
        raw = np.ascontiguousarray(rec).view(np.uint8).reshape(-1)
        self._size += (-self._size) % 64  # (the padding bytes are never read)
        self._offsets[name] = self._size
        self._itemsize[name] = rec.dtype.itemsize
        if raw.size:
            self._chunks.append((self._size, raw))
        self._size += raw.size

    def upload(self, device: torch.device) -> None:
        # (the pieces go straight into the pinned staging slot: no concatenated host copy in between)
        self.device_buf = _hip.pieces_to_device(self._chunks, max(self._size, 64), device)

    def ptr(self, name: str, index: int = 0) -> int:
        return self.device_buf.data_ptr() + self._offsets[name] + index * self._itemsize[name]


class _NativePlan:
    """What callers read of a step planned by the library (bench.py, tests): counts only."""

    __slots__ = ("n_prims", "arena_floats", "n_forward_launches", "n_backward_launches")

    def __init__(self, n_prims, arena_floats, n_fwd, n_bwd):
        self.n_prims, self.arena_floats = n_prims, arena_floats
        self.n_forward_launches, self.n_backward_launches = n_fwd, n_bwd


class _State:
    __slots__ = ("plan", "pack", "fixed", "B", "generation", "valid_examples", "features", "backward_rows", "step_pack", "touched")


class NMNEngine:
    def __init__(self, net, image_feature_size, module_channels: int, class_projection_channels: int):
        cin, H, W = image_feature_size
        if module_channels != C:
            raise NotImplementedError(
                "the gfx950 kernels are built for module_channels == 128 (got %d)" % module_channels)
        if (H, W) not in ((14, 14), (28, 28)):
            raise NotImplementedError("the gfx950 kernels are built for 14x14 and 28x28 feature maps (got %dx%d)" % (H, W))
        if cin % C or class_projection_channels % C:
            raise NotImplementedError("channel counts must be multiples of 128")
        self.net = net
        self.cin, self.H, self.W = cin, H, W
        self.HW = H * W
        # 28x28 maps: conv / weight-gradient workgroups cover one of four 7-row bands of an item, and a
        # weight-gradient slab is 64 x 64 channels instead of 64 x 128 (csrc/conv_wgrad.hip)
        self.banded = (H, W) != (14, 14)
        self.cproj = class_projection_channels
        self.arena: Optional[ParamArena] = None
        self.direct_grads = False  # True: gradients stay in the arena, nothing is handed to autograd
        self.generation = 0
        self._ws: Dict[str, torch.Tensor] = {}
        self._fixed_cache: Dict[int, Dict[str, np.ndarray]] = {}
        self.compiler = pc.ProgramCompiler(
            net.vocabulary.get_index_to_token_vocabulary("programs"), module_channels)
        self.last_plan: Optional[_NativePlan] = None
        # launches of a pass are collected and issued by ONE library call (pnmn_run_launches)
        self._list: Optional[_hip.LaunchList] = None
        # data parallel: called with k when every kernel that writes gradient piece k of the arena has been
        # queued (see grad_pieces); a trainer points it at its EarlyReducer.piece_ready
        self.on_grad_piece = None
        # event behind an optimiser step of these parameters that runs on another stream than its caller's
        # (trainers/joint_training.py: overlap_optimizer); every entry point that reads parameters waits for it
        self.params_ready = None
        # CUs the trunk's conv launches are planned for: 0 = all of them; a trainer that runs the trunk on its own stream
        # beside other work sets the number it may count on (JointTrainingStep: 192 -- the seq2seq passes' multi-CU
        # kernels hold 64-96 CUs, and a launch cut for 256 workgroups then takes two rounds)
        self.conv_cus = 0
        self.last_counts = (0, 0)  # 3x3 / projection records of the last natively planned batch
        self.wgrad_cus = 0  # the same for the weight-gradient launches: at most that many (persistent) workgroups
        self._planner = None
        self._native_fixed: Dict[tuple, dict] = {}

    def _conv(self, ptr, n, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu):
        self._list.add(_hip.OP_CONV, n, ptr, c=self.conv_cus,
                       p=(self.H, self.W, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu))

    def _op(self, op: int, n: int, a: int, b: int = 0, c: int = 0, p=()) -> None:
        self._list.add(op, n, a, b, c, p)

    def _begin_list(self) -> None:
        # (a list left over from a pass that raised is dropped here)
        self._list = _hip.LaunchList()

    # ---- launch trace (measurement: bench.py's roofline passes, scripts/conv_launch_table.py) ----------
    def begin_trace(self) -> None:
        """From here to ``end_trace`` the library brackets every conv / weight-gradient launch it issues -- the stem's
        list, the trunk planner's forward list, the backward list: the shipped host path -- with events on its stream
        (``pnmn_launch_trace_begin``)."""
        _hip.check(_hip.lib().pnmn_launch_trace_begin(), "launch_trace_begin")

    def end_trace(self) -> list:
        """[(kernel family, call site, algorithmic FLOPs, ms, algorithmic bytes, kernel launches)] of the traced
        launches, in issue order.  FLOPs and bytes are the library's (csrc/host_run.hip reads the items back); the call
        site is named here from the call's shape and place in the step."""
        lib = _hip.lib()
        cap = 4096
        out = np.zeros(cap, _hip.LAUNCH_TIMING)
        n = np.zeros(1, np.int32)
        rc = lib.pnmn_launch_trace_end(out.ctypes.data, cap, n.ctypes.data)
        if rc == _hip.EAGAIN:  # (the library keeps the trace: read it again with room for all of it)
            cap = int(n[0])
            out = np.zeros(cap, _hip.LAUNCH_TIMING)
            rc = lib.pnmn_launch_trace_end(out.ctypes.data, cap, n.ctypes.data)
        _hip.check(rc, "launch_trace_end")
        t = out[: int(n[0])]
        cin_chunks, cls_blocks = self.cin // C, self.cproj // C
        sites = []
        for r in t:
            p = r["p"]
            if r["op"] == _hip.OP_CONV:
                chunks, ntaps, in_stride, out_stride, relu = int(p[2]), int(p[3]), int(p[4]), int(p[5]), int(p[7])
                if ntaps == 9:
                    what = "stem conv1" if chunks == cin_chunks and chunks > 1 else ("module conv" if relu else "module dgrad")
                elif chunks == 2 and relu:
                    what = "projection"
                elif relu:
                    what = "classifier conv"
                else:
                    what = "classifier dgrad" if in_stride == self.cproj and chunks == cls_blocks and (cls_blocks > 1 or not sites or sites[-1] == "classifier wgrad") else "projection dgrad"
                if what == "module conv" and sites and sites[-1] == "stem conv1":
                    what = "stem conv2"
                if what == "module dgrad" and sites and sites[-1] == "stem conv2 wgrad":
                    what = "stem conv2 dgrad"
            else:
                ntaps, blocks, out_blocks = int(p[2]), int(p[3]), int(p[4])
                if ntaps == 1:
                    what = "projection wgrad" if blocks == 2 else "classifier wgrad"
                elif blocks == cin_chunks and blocks > 1:
                    what = "stem conv1 wgrad"
                else:
                    what = "module wgrad"
            sites.append(what)
        # the backward's tail is [stem conv2 wgrad, stem conv2 dgrad, stem conv1 wgrad]: same shapes as the modules'
        for i in range(2, len(sites)):
            if sites[i] == "stem conv1 wgrad" and sites[i - 1] == "module dgrad" and sites[i - 2] == "module wgrad":
                sites[i - 1], sites[i - 2] = "stem conv2 dgrad", "stem conv2 wgrad"
        return [("conv_nhwc" if r["op"] == _hip.OP_CONV else "conv_wgrad", w, float(r["flops"]), float(r["ms"]), float(r["bytes"]), 1)
                for r, w in zip(t, sites)]

    def _flush_list(self, st: int, what: str) -> None:
        self._list.run(st, what)
        self._list = None

    # ---- parameters ---------------------------------------------------------------------------
    def trunk_named_parameters(self):
        return [(n, p) for n, p in self.net.named_parameters()
                if not n.startswith(("classifier.4.", "classifier.6."))]

    def ensure_arena(self) -> ParamArena:
        device = self.net.stem[0].weight.device
        if device.type != "cuda":
            raise _hip.HipLibraryError(
                "NeuralModuleNetwork parameters are on %s: the HIP path needs a ROCm device "
                "(call .to('cuda')); there is no CPU fallback" % device)
        _hip.lib()
        if self.arena is None or self.arena.device != device or not self.arena.intact():
            self.arena = ParamArena(self.trunk_named_parameters(), device)
            self.arena_generation = getattr(self, "arena_generation", 0) + 1
            self._build_tables()
            self._ws.clear()
            self._fixed_cache.clear()
            # everything that holds the old arena's absolute addresses goes with it: the uploaded records and launch rows
            # of the native path (keyed on workspace pointers the allocator may hand out again) and the planner's offset
            # tables (a stale hit read and wrote freed memory)
            self._native_fixed.clear()
            self._drop_planner()
        return self.arena

    def grad_pieces(self):
        """The trunk gradient arena as contiguous float ranges in the order backward completes them: piece 0 =
        classifier conv + every module (final once the deferred module / projection weight gradients are queued:
        ~48 of 52 MB), piece 1 = the stem (final with the last kernel of the trunk backward).  Parameters sit in
        ``named_parameters`` order: stem.0, stem.2, classifier.0, then one child per program token."""
        a = self.ensure_arena()
        cut = a.offsets["classifier.0.weight"]
        assert all(a.offsets[n] < cut for n in a.names if n.startswith("stem.")) and \
            all(a.offsets[n] >= cut for n in a.names if not n.startswith("stem."))
        return [(a, cut, a.total), (a, 0, cut)]

    def _build_tables(self) -> None:
        a = self.arena
        vocab = self.net.vocabulary.get_index_to_token_vocabulary("programs")
        V = max(vocab) + 1
        w3 = -np.ones((V, 6), np.int64)
        b3 = -np.ones((V, 6), np.int64)
        wt3 = -np.ones((V, 6), np.int64)
        dotw = -np.ones(V, np.int64)
        dotb = -np.ones(V, np.int64)
        wt_items = []  # (src name, cout, cin, ntaps, wt offset)
        wt_cursor = 0

        def add_wt(name):
            nonlocal wt_cursor
            p = a.param(name)
            co, ci, kh, kw = p.shape
            off = wt_cursor
            wt_cursor += p.numel()
            wt_items.append((name, co, ci, kh * kw, off))
            return off

        for idx, tok in vocab.items():
            kind = self.compiler.kinds[idx]
            if kind in (pc.ATT, pc.QUERY, pc.REL, pc.CMP):
                names = {pc.ATT: [None, "conv1", "conv2"], pc.QUERY: [None, "conv1", "conv2"],
                         pc.REL: [None, "conv1", "conv2", "conv3", "conv4", "conv5"],
                         pc.CMP: ["projection", "conv1", "conv2"]}[kind]
                for j, nm in enumerate(names):
                    if nm is None:
                        continue
                    w3[idx, j] = a.offsets["%s.%s.weight" % (tok, nm)]
                    b3[idx, j] = a.offsets["%s.%s.bias" % (tok, nm)]
                    wt3[idx, j] = add_wt("%s.%s.weight" % (tok, nm))
            head = {pc.ATT: "conv3", pc.REL: "conv6", pc.SAME: "conv"}.get(kind)
            if head:
                dotw[idx] = a.offsets["%s.%s.weight" % (tok, head)]
                dotb[idx] = a.offsets["%s.%s.bias" % (tok, head)]
        # parameters (arena name indices) a program token's module owns, and those every valid program uses: what a
        # backward pass hands a gradient to (ParamArena.touched)
        index = {n: i for i, n in enumerate(a.names)}
        self._token_params = {idx: np.array([i for n, i in index.items() if n.startswith(tok + ".")], np.int64)
                              for idx, tok in vocab.items()}
        self._always_params = np.array([i for n, i in index.items() if n.startswith(("stem.", "classifier."))], np.int64)
        self._stem_params = np.array([i for n, i in index.items() if n.startswith("stem.")], np.int64)
        self._classifier_params = np.array([i for n, i in index.items() if n.startswith("classifier.")], np.int64)
        self._reach_cache = {}
        a.touched = np.zeros(len(a.names), bool)
        self.wt_stem2 = add_wt("stem.2.weight")
        self.wt_cls0 = add_wt("classifier.0.weight")
        self.tables = WeightTables(w3, b3, wt3, dotw, dotb)
        self.wt = torch.zeros(wt_cursor, dtype=torch.float32, device=a.device)
        rec = np.zeros(len(wt_items), _hip.WTRANS_ITEM)
        for i, (name, co, ci, nt, off) in enumerate(wt_items):
            rec[i]["src"] = a.flat.data_ptr() + a.offsets[name] * 4
            rec[i]["dst"] = self.wt.data_ptr() + off * 4
            rec[i]["cout"], rec[i]["cin"], rec[i]["ntaps"] = co, ci, nt
        self._wt_records = _hip.to_device(rec, a.device)
        self._wt_count = len(wt_items)
        self.ones = torch.ones(self.HW, dtype=torch.float32, device=a.device)
        self.planner_config = PlannerConfig(wgrad_chunk=2 if self.banded else 8, wgrad_groups=1)

    # ---- workspaces ---------------------------------------------------------------------------
    def _buf(self, name: str, numel: int) -> torch.Tensor:
        t = self._ws.get(name)
        if t is None or t.numel() < numel:
            t = torch.empty(int(numel * 1.25) if t is not None else numel, dtype=torch.float32,
                            device=self.arena.device)
            self._ws[name] = t
        return t

    def _fixed_records(self, B: int, ws: Dict[str, torch.Tensor], xin_ptrs: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        """Work lists of the stem and classifier convs: depend only on B and buffer addresses.  ``xin_ptrs``: the input
        maps are rows of a resident feature store (one device address per example, other rows every step): the cached
        lists are completed with them for the two kernels that read the input, stem conv1 and its weight gradient."""
        key = (B, self.conv_cus) + tuple(ws[k].data_ptr() for k in sorted(ws))
        hit = self._fixed_cache.get(key)
        if hit is not None:
            return self._with_inputs(hit, xin_ptrs)
        a = self.arena
        HW = self.HW
        e = np.arange(B, dtype=np.int64)
        m128, mcin, mcls = HW * C * 4, HW * self.cin * 4, HW * self.cproj * 4
        P = a.flat.data_ptr()
        G = a.grad.data_ptr()

        def po(name):
            return P + a.offsets[name] * 4

        def go(name):
            return G + a.offsets[name] * 4

        def conv(inp, w, b, out, gate=None):
            r = np.zeros(B, _hip.CONV_ITEM)
            r["in"], r["weight"], r["out"], r["dilation"] = inp, w, out, 1
            if b is not None:
                r["bias"] = b
            if gate is not None:
                r["gate"] = gate
            return r

        def wg(x, dy, gate=None):
            r = np.zeros(B, _hip.WGRAD_ITEM)
            r["x"], r["dy"], r["dilation"] = x, dy, 1
            if gate is not None:
                r["gate"] = gate
            return r

        def jobs(dw, db, yblocks, taps=9):
            # items per job: a launch of ceil(B / chunk) * yblocks workgroups costs ceil(. / 256) rounds of
            # `chunk` items each (+ a fraction of an item for the atomic add of the job's slab into the shared
            # weight gradient); 260 workgroups cost two rounds, 208 one -- take the chunk with the shortest makespan
            flush = 0.5
            if self.banded:  # four bands per item, twice the slabs per weight
                yblocks *= 2
                sizes = range(1, 9)
            elif taps == 9:
                # the streamed 3x3 kernel (csrc/conv_wgrad_stream.h): slabs of 64 x 64 channels, the flush a tenth of an
                # item's 35 us -- a 64-row shard goes out as 64 one-item jobs on 256 workgroups (stem conv2: 126 -> ~45 us)
                yblocks *= 2
                sizes, flush = range(1, 33), 0.3
            else:
                sizes = range(4, 33)
            cus = self.conv_cus or 256  # (a trunk that shares the chip: rounds of the CUs it can count on)
            chunk = min(sizes, key=lambda c: (-(-(-(-B // c) * yblocks) // cus)) * (c + flush))
            starts = np.arange(0, B, chunk)
            j = np.zeros(starts.size, _hip.WGRAD_JOB)
            j["dw"], j["dbias"] = dw, db
            j["item_begin"], j["item_end"] = starts, np.minimum(starts + chunk, B)
            return j

        xin = ws["xin"].data_ptr() if "xin" in ws else 0  # (0: resident rows, filled in per step by _with_inputs)
        s1, s2 = ws["stem1"].data_ptr(), ws["feat"].data_ptr()
        gs1, gs2 = ws["gstem1"].data_ptr(), ws["gfeat"].data_ptr()
        fin, gfin = ws["final"].data_ptr(), ws["gfinal"].data_ptr()
        cls, gcls = ws["cls"].data_ptr(), ws["gcls"].data_ptr()
        out = {
            "stem1": conv(xin + e * mcin, po("stem.0.weight"), po("stem.0.bias"), s1 + e * m128),
            "stem2": conv(s1 + e * m128, po("stem.2.weight"), po("stem.2.bias"), s2 + e * m128),
            "cls": conv(fin + e * m128, po("classifier.0.weight"), po("classifier.0.bias"), cls + e * mcls),
            "cls_dgrad": conv(gcls + e * mcls, self.wt.data_ptr() + self.wt_cls0 * 4, None, gfin + e * m128),
            "stem2_dgrad": conv(gs2 + e * m128, self.wt.data_ptr() + self.wt_stem2 * 4, None, gs1 + e * m128,
                                gate=s2 + e * m128),
            "cls_wg": wg(fin + e * m128, gcls + e * mcls),
            "cls_wg_jobs": jobs(go("classifier.0.weight"), go("classifier.0.bias"), 2 * self.cproj // C, taps=1),
            "stem2_wg": wg(s1 + e * m128, gs2 + e * m128, gate=s2 + e * m128),
            "stem2_wg_jobs": jobs(go("stem.2.weight"), go("stem.2.bias"), 2),
            "stem1_wg": wg(xin + e * mcin, gs1 + e * m128, gate=s1 + e * m128),
            "stem1_wg_jobs": jobs(go("stem.0.weight"), go("stem.0.bias"), 2 * self.cin // C),
        }
        self._fixed_cache[key] = out
        return self._with_inputs(out, xin_ptrs)

    @staticmethod
    def _with_inputs(fixed: Dict[str, np.ndarray], xin_ptrs: Optional[np.ndarray]) -> Dict[str, np.ndarray]:
        if xin_ptrs is None:
            return fixed
        out = dict(fixed)
        out["stem1"] = fixed["stem1"].copy()
        out["stem1"]["in"] = xin_ptrs
        out["stem1_wg"] = fixed["stem1_wg"].copy()
        out["stem1_wg"]["x"] = xin_ptrs
        return out

    # ---- forward --------------------------------------------------------------------------------
    def wait_params(self, device) -> None:
        if self.params_ready is not None:
            torch.cuda.current_stream(device).wait_event(self.params_ready)

    def begin_forward(self, features: torch.Tensor, need_backward: bool, rows: Optional[torch.Tensor] = None):
        """The part of the forward pass that does not depend on the programs: layout change of the
        input features and the two stem convolutions.  A trainer whose programs are still being
        produced (joint training: sampled on the device, scheduled on the host) launches this first
        and hands the returned token to ``run_forward_tokens`` -- the GPU then has ~3 ms more work queued
        while the host compiles and schedules the sampled programs.  ``rows`` (int64, on the device): run on
        examples ``features[rows]`` without materialising that gather (0.8 MB per example): the layout kernel reads
        through the index."""
        a = self.ensure_arena()
        dev = a.device
        self.wait_params(dev)
        if features.device != dev:
            raise _hip.HipLibraryError("features on %s but the network is on %s" % (features.device, dev))
        from probnmn.data.feature_store import ResidentRows

        resident = isinstance(features, ResidentRows)
        if resident and rows is not None:
            raise ValueError("rows of a resident batch: pass features.subset(rows) instead")
        B = features.size(0) if rows is None else int(rows.numel())
        if rows is not None and (rows.dtype != torch.long or rows.device != dev or not rows.is_contiguous()):
            raise ValueError("rows must be a contiguous int64 tensor on the network's device")
        if tuple(features.shape[1:]) != (self.cin, self.H, self.W):
            raise ValueError("expected features (B,%d,%d,%d), got %s" % (self.cin, self.H, self.W, tuple(features.shape)))
        # features already in the kernels' layout (a `channels_last` tensor, e.g. from
        # probnmn.data.feature_store: the ingest kernel writes NHWC): used in place, no layout pass
        # ... or rows of a feature store that lives in HBM in that layout (DeviceFeatureStore): nothing is gathered or
        # copied, stem conv1 and its weight gradient get one pointer per example
        nhwc = resident or (features.dtype == torch.float32 and not features.is_contiguous()
                            and features.is_contiguous(memory_format=torch.channels_last))
        subset = rows is not None
        if nhwc and rows is not None:  # (a row subset of an NHWC batch: gather it, the layout pass is what reads through rows)
            features, rows = features[rows], None
            nhwc = (not features.is_contiguous()) and features.is_contiguous(memory_format=torch.channels_last)
        if not nhwc:
            features = features.contiguous().float()
        HW = self.HW
        st = _hip.stream_ptr(dev)
        self.generation += 1

        names = ["xin", "stem1", "feat", "final", "cls"]
        sizes = [B * HW * self.cin, B * HW * C, B * HW * C, B * HW * C, B * HW * self.cproj]
        if need_backward:
            names += ["gstem1", "gfeat", "gfinal", "gcls"]
            sizes += [B * HW * C, B * HW * C, B * HW * C, B * HW * self.cproj]
        if nhwc:
            names, sizes = names[1:], sizes[1:]
        ws = {n: self._buf(n, s) for n, s in zip(names, sizes)}
        if nhwc and not resident:
            ws["xin"] = features.permute(0, 2, 3, 1).reshape(-1)  # a view of the caller's storage
        if not need_backward:  # records still reference gradient buffers; point them somewhere valid
            for n in ("gstem1", "gfeat", "gfinal", "gcls"):
                ws[n] = ws["stem1"]
        fixed = self._fixed_records(B, ws, features.pointers() if resident else None)
        pack = _Pack()
        for k in ("stem1", "stem2"):
            pack.add(k, fixed[k])
        pack.upload(dev)
        self._begin_list()
        if not nhwc:
            self._op(_hip.OP_NCHW_TO_NHWC, B, features.data_ptr(), b=ws["xin"].data_ptr(),
                     c=0 if rows is None else rows.data_ptr(), p=(self.cin, HW))  # (c: the layout kernel reads through rows)
        self._conv(pack.ptr("stem1"), B, self.cin // C, 9, self.cin, C, 1, 1)
        self._conv(pack.ptr("stem2"), B, 1, 9, C, C, 1, 1)
        self._flush_list(st, "stem")
        return {"B": B, "ws": ws, "fixed": fixed, "need_backward": need_backward, "generation": self.generation,
                "features": features, "pack": pack, "rows": rows, "subset": subset, "resident": resident}

    # ---- the library's trunk planner ------------------------------------------------------------------
    def _native_planner(self) -> int:
        if self._planner is None:
            t = self.tables
            kinds = np.ascontiguousarray(self.compiler.kinds, dtype=np.int32)
            arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (t.w3, t.b3, t.wt3, t.dotw, t.dotb)]
            cfg = np.zeros(1, _hip.TRUNK_CONFIG)
            cfg[0] = (kinds.ctypes.data,) + tuple(a.ctypes.data for a in arrs) + (
                kinds.size, C, self.H, self.W, self.planner_config.wgrad_chunk, self.planner_config.wgrad_groups,
                int(self.planner_config.fuse_mask_bwd), int(self.planner_config.sole_writer_rmw),
                int(self.planner_config.sort_by_weight), 0)
            out = np.zeros(1, np.uint64)
            _hip.check(_hip.lib().pnmn_trunk_planner_create(cfg.ctypes.data, out.ctypes.data), "trunk_planner_create")
            self._planner = int(out[0])
            self._planner_io = np.zeros(1, _hip.TRUNK_IO)
            self._planner_bwd = np.zeros((1024, 8), np.uint64)
            self._planner_valid = np.zeros(4096, np.uint8)
        return self._planner

    def _drop_planner(self) -> None:
        if getattr(self, "_planner", None):
            _hip.lib().pnmn_trunk_planner_destroy(self._planner)
        self._planner = None

    def __del__(self):
        try:
            self._drop_planner()
        except Exception:
            pass

    def _native_fixed_rows(self, B: int, ws: Dict[str, torch.Tensor], fixed, dev) -> dict:
        """What the planner's lists have around the module programs: depends only on B and the workspace addresses,
        so the records are uploaded ONCE and the launch rows are kept (pooled / d(pooled) are patched in per step)."""
        key = (B, self.conv_cus, self.wgrad_cus) + tuple(ws[k].data_ptr() for k in sorted(ws))
        hit = self._native_fixed.get(key)
        if hit is not None:
            return hit
        if len(self._native_fixed) > 8:
            self._native_fixed.clear()
        a = self.arena
        H, W, HW = self.H, self.W, self.HW
        pack = _Pack()
        for k in ("cls", "cls_dgrad", "stem2_dgrad", "cls_wg", "cls_wg_jobs", "stem2_wg", "stem2_wg_jobs", "stem1_wg", "stem1_wg_jobs"):
            pack.add(k, fixed[k])
        pack.upload(dev)

        def rows(build):
            lst = _hip.LaunchList()
            build(lst)
            return np.array(lst._rows, dtype=np.uint64).reshape(-1, 8)

        def fwd_tail(l):
            l.add(_hip.OP_CONV, B, pack.ptr("cls"), c=self.conv_cus, p=(H, W, 1, 1, C, self.cproj, self.cproj // C, 1))
            l.add(_hip.OP_MAXPOOL_FWD, B, ws["cls"].data_ptr(), 0, 0, (H, W, self.cproj))  # b = pooled: per step

        def bwd_head(l):
            l.add(_hip.OP_ZERO, 0, a.grad.data_ptr(), a.total * 4)
            l.add(_hip.OP_ZERO, 0, ws["gfeat"].data_ptr(), B * HW * C * 4)
            l.add(_hip.OP_TRANSPOSE_WEIGHTS, self._wt_count, self._wt_records.data_ptr())
            l.add(_hip.OP_MAXPOOL_BWD, B, ws["cls"].data_ptr(), 0, ws["gcls"].data_ptr(), (H, W, self.cproj))  # b = d(pooled)
            l.add(_hip.OP_WGRAD, len(fixed["cls_wg_jobs"]), pack.ptr("cls_wg"), pack.ptr("cls_wg_jobs"),
                  p=(H, W, 1, 1, self.cproj // C, C, self.cproj, self.wgrad_cus))
            l.add(_hip.OP_CONV, B, pack.ptr("cls_dgrad"), c=self.conv_cus, p=(H, W, self.cproj // C, 1, self.cproj, C, 1, 0))

        def bwd_tail(l):
            l.add(_hip.OP_WGRAD, len(fixed["stem2_wg_jobs"]), pack.ptr("stem2_wg"), pack.ptr("stem2_wg_jobs"), p=(H, W, 9, 1, 1, C, C, self.wgrad_cus))
            l.add(_hip.OP_CONV, B, pack.ptr("stem2_dgrad"), c=self.conv_cus, p=(H, W, 1, 9, C, C, 1, 0))
            l.add(_hip.OP_WGRAD, len(fixed["stem1_wg_jobs"]), pack.ptr("stem1_wg"), pack.ptr("stem1_wg_jobs"),
                  p=(H, W, 9, self.cin // C, 1, self.cin, C, self.wgrad_cus))

        # rows patched per step, by name: d(pooled) of the backward head's max-pool, the stem conv1 weight gradient's items
        hit = {"pack": pack, "fwd_tail": rows(fwd_tail), "bwd_head": rows(bwd_head), "bwd_tail": rows(bwd_tail), "dpooled_row": 3,
               "stem1_wg_row": 2}
        self._native_fixed[key] = hit
        return hit

    def run_forward_tokens(self, features: torch.Tensor, programs: np.ndarray, need_backward: bool, started=None):
        """``run_forward`` from the token matrix (HOST int64 [B, T]) through the library's trunk planner: one call
        compiles, plans, uploads and launches.  Returns (pooled, backward state, validity of every program)."""
        if started is None:
            started = self.begin_forward(features, need_backward)
        elif (started["generation"] != self.generation or started["need_backward"] != need_backward
              or (not started["subset"] and started["B"] != features.size(0))):
            raise ValueError("begin_forward token does not belong to this forward pass")
        a = self.ensure_arena()
        self.wait_params(a.device)
        dev = a.device
        B, ws, fixed = started["B"], started["ws"], started["fixed"]
        programs = np.ascontiguousarray(programs, dtype=np.int64)
        if programs.ndim != 2 or programs.shape[0] != B:
            raise ValueError("programs must be (%d, length), got shape %s" % (B, programs.shape))
        st = _hip.stream_ptr(dev)
        planner = self._native_planner()
        rows = self._native_fixed_rows(B, ws, fixed, dev)
        step_pack = None
        if started.get("resident"):  # (the stem weight gradient's items name this step's rows: uploaded per step)
            step_pack = _Pack()
            step_pack.add("stem1_wg", fixed["stem1_wg"])
            step_pack.upload(dev)
            rows["bwd_tail"][rows["stem1_wg_row"], 0] = step_pack.ptr("stem1_wg")
        H, W, HW = self.H, self.W, self.HW
        pooled = torch.empty(B, self.cproj * (H // 2) * (W // 2), dtype=torch.float32, device=dev)
        rows["fwd_tail"][1, 1] = pooled.data_ptr()
        if B > self._planner_valid.size:
            self._planner_valid = np.zeros(2 * B, np.uint8)
        act = self._ws.get("act")
        if act is None:
            act = self._buf("act", 1 << 20)
        io = self._planner_io
        while True:
            gact = self._buf("gact", act.numel()) if need_backward else act
            io[0] = (programs.ctypes.data, a.flat.data_ptr(), a.grad.data_ptr(), self.wt.data_ptr(), act.data_ptr(), gact.data_ptr(),
                     ws["feat"].data_ptr(), ws["gfeat"].data_ptr(), ws["final"].data_ptr(), ws["gfinal"].data_ptr(),
                     self.ones.data_ptr(), min(act.numel(), gact.numel()),
                     rows["fwd_tail"].ctypes.data, rows["bwd_head"].ctypes.data, rows["bwd_tail"].ctypes.data,
                     self._planner_bwd.ctypes.data, self._planner_valid.ctypes.data, 0,
                     B, programs.shape[1], rows["fwd_tail"].shape[0], rows["bwd_head"].shape[0], rows["bwd_tail"].shape[0],
                     self._planner_bwd.shape[0], int(need_backward), 1, 0, 0, 0, 0, 0, 0, 0, self.conv_cus, self.wgrad_cus,
                     0, 0, 0, (0, 0, 0, 0))
            rc = _hip.lib().pnmn_trunk_plan_and_launch(planner, io.ctypes.data, st)
            if rc != _hip.EAGAIN:
                break
            act = self._buf("act", int(io[0]["arena_floats"]))  # (grows by a quarter beyond what is asked for)
        _hip.check(rc, "trunk plan_and_launch")
        out = io[0]
        valid = self._planner_valid[:B].copy()
        self.last_plan = _NativePlan(int(out["n_prims"]), int(out["arena_floats"]), int(out["n_fwd"]), int(out["n_bwd"]))
        self.last_counts = (int(out["n_conv"]), int(out["n_proj"]))
        state = None
        if need_backward:
            state = _State()
            state.plan, state.pack, state.fixed, state.B = self.last_plan, rows["pack"], fixed, B
            state.features = started["features"]  # (the stem's weight gradient reads the input again)
            state.step_pack = step_pack
            state.generation = self.generation
            state.touched = self._touched_from_tokens(out["touched_tokens"], bool(valid.any()))
            n = int(out["n_bwd"])
            state.backward_rows = (self._planner_bwd[:n].copy(), int(out["bwd_piece_cut"]), rows["dpooled_row"])
        return pooled, state, valid

    def _touched_from_tokens(self, words, any_valid: bool) -> np.ndarray:
        """``_touched_by`` from what the library's planner reports of the batch it just compiled (pnmn_trunk_io.touched_tokens:
        the program tokens some valid program's result depends on) -- no per-program work on the host: the Python walk below
        cost 25 us per NEW program, 8 ms per 1024-question iteration, between the samples' arrival and the trunk's launch."""
        mask = np.zeros(len(self.arena.names), bool)
        mask[self._classifier_params] = True
        if any_valid:
            mask[self._stem_params] = True
            for w in range(4):
                bits = int(words[w])
                while bits:
                    low = bits & -bits
                    owned = self._token_params.get(64 * w + low.bit_length() - 1)
                    if owned is not None and owned.size:
                        mask[owned] = True
                    bits ^= low
        return mask

    def _touched_by(self, programs: np.ndarray, valid: np.ndarray) -> np.ndarray:
        """Arena parameters that get a gradient from a backward pass over these programs, as the reference's autograd would
        hand them out (nmn.py:197-241): the classifier conv always (every example's zero / final map goes through it: its
        gradient is a tensor even when every program is invalid -- all zeros then); the stem and the modules of the calls the
        VALID programs' RESULTS depend on.  The interpreter is a two-register machine: a chain whose value is overwritten by a
        later ``scene`` before a binary module reads it is executed but not part of the loss graph -- its modules keep
        ``grad = None`` and torch.optim.Adam starts no state for them; an invalid program's partial graph is dropped whole
        (nmn.py:235-241).  One mask per distinct program, cached by the token row's bytes."""
        mask = np.zeros(len(self.arena.names), bool)
        mask[self._classifier_params] = True
        rows = programs[valid.astype(bool)]
        if rows.size:
            mask[self._stem_params] = True
            cache = self._reach_cache
            if len(cache) > 200000:
                cache.clear()
            for row in rows:
                key = row.tobytes()
                hit = cache.get(key)
                if hit is None:
                    hit = cache[key] = self._reachable_params(row)
                mask[hit] = True
        return mask

    def _reachable_params(self, tokens: np.ndarray) -> np.ndarray:
        """Indices (into the arena's names) of the parameters of the modules one valid program's result depends on."""
        prog = self.compiler.compile(tokens.tolist())
        owned = []
        if prog.valid:
            calls = prog.calls
            seen, todo = set(), [prog.result]
            while todo:
                v = todo.pop()
                if v < 2 or v in seen:  # FEAT / ONES, or a call already walked
                    continue
                seen.add(v)
                c = calls[v - 2]
                p = self._token_params.get(int(c.token))
                if p is not None and p.size:
                    owned.append(p)
                todo += [c.a, c.b]
        return np.unique(np.concatenate(owned)) if owned else np.zeros(0, np.int64)

    # ---- backward -------------------------------------------------------------------------------
    def run_backward(self, state: _State, dpooled: torch.Tensor):
        if state.generation != self.generation:
            raise RuntimeError(
                "NeuralModuleNetwork.forward was called again before backward of the previous call: "
                "the engine keeps one step's activations (run evaluation passes under torch.no_grad())")
        a = self.arena
        self.last_touched = state.touched  # (diagnostics / smoke(): which parameters this backward pass reaches)
        if a.touched is not None and state.touched is not None:
            a.touched |= state.touched
        chk = _hip.check
        dev = a.device
        st = _hip.stream_ptr(dev)
        dpooled = dpooled.contiguous()
        # d(pooled) was produced (and is owned) by the stream of the fully connected layers; when the trunk runs on its own
        # stream the launches below read it there AFTER this function has returned and autograd has dropped the tensor --
        # without this the allocator may hand its memory to the next main-stream allocation while the max-pool backward
        # has not run yet (seen as NMN gradients off by ~1e-3 in one run in four of the 1024-row side-stream test)
        dpooled.record_stream(torch.cuda.current_stream(dev))

        # the whole list came out of the planner (zeroing of the gradient buffers included): d(pooled) is the only
        # thing that was not known then
        rows, piece_cut, where = state.backward_rows
        rows[where, 1] = dpooled.data_ptr()
        lib_run = _hip.lib().pnmn_run_launches
        _hip.mark("trunk backward begins (dpooled ready)")
        if self.on_grad_piece is not None and 0 < piece_cut < rows.shape[0]:
            chk(lib_run(rows.ctypes.data, piece_cut, st), "trunk backward")
            self.on_grad_piece(0)  # classifier conv + all module gradients are final behind these launches
            chk(lib_run(rows[piece_cut:].ctypes.data, rows.shape[0] - piece_cut, st), "trunk backward (stem)")
        else:
            chk(lib_run(rows.ctypes.data, rows.shape[0], st), "trunk backward")
        if self.on_grad_piece is not None:
            if not 0 < piece_cut < rows.shape[0]:
                self.on_grad_piece(0)
            self.on_grad_piece(1)
        if self.direct_grads:
            a.attach_grads()
            return [None] * len(a.names)
        return [a.grad_view(n) for n in a.names]
