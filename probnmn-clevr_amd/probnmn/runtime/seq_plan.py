"""The seq2seq half of a question-coding / joint-training iteration as a static launch plan.

The reference runs ProgramGenerator, QuestionReconstructor and ProgramPrior through AllenNLP's ``SimpleSeq2Seq`` loop under
autograd (probnmn/trainers/question_coding_trainer.py:128-160, joint_training_trainer.py:150-190,
probnmn/modules/seq2seq_base.py:101-276, probnmn/models/program_prior.py:80-155).  ``Seq2SeqBase`` of this package keeps
that call structure with one autograd node per kernel -- right for evaluation, the drop-in graft and the kernel tests, but
at 128 questions per GPU the iteration is then bound by the HOST: ~40 autograd-function applications, ~60 small torch ops
and 20 library GEMM calls per iteration cost 5 of its 6.5 ms (profiles/r06a_b128_host.txt).

For given row counts and token widths the passes are a FIXED sequence of launches over FIXED shapes.  ``Seq2SeqPlan``
therefore builds, once per shape signature,

* a workspace of persistent device buffers (every activation, saved tensor, gradient and scratch block of the passes);
* three forward call lists -- generator (encoder over [unsupervised ; supervised] questions, sampling + teacher-forced
  decoder pair, losses), reconstructor (encoder over [sampled ; ground-truth] programs, teacher-forced decoder, loss),
  prior (LSTM language model over the samples, no gradient) -- and one backward list; an entry is a bound C-ABI entry
  point of libprobnmn_hip.so with its final argument tuple;
* ONE autograd node (``_PlanNode``) whose outputs are the per-row losses and whose backward replays the backward list:
  seq_nll backward -> output-projection data gradient -> the three decoders' backward in one launch -> encoder-output
  gradients -> the two encoders' layers, with every product over all time steps on ``pnmn_gemm`` and EVERY weight
  gradient of a model deferred into one grouped GEMM launch at the end.  Parameter gradients are written straight into
  one flat buffer per model whose slices become the parameters' ``.grad``.

Per iteration the host then replays ~100 prepared calls (a few microseconds each) and touches no torch op in the passes.
Arithmetic is the eager path's kernel for kernel (same recurrent kernels, same losses); the GEMMs are this library's
instead of hipBLASLt's, so results agree with ``Seq2SeqBase.forward`` to fp32 round-off (tests/test_seq_plan_gpu.py)."""
from typing import Dict, List, Optional

import numpy as np
import torch

from probnmn import _hip


#: workgroup slots of the chip for a launch of split-K products (256 CUs, the 64 KB workgroups of pnmn_gemm sit two to a CU)
#: and the shortest chunk (k tiles) a product is cut into
SPLIT_SLOTS = int(__import__("os").environ.get("PNMN_PLAN_SPLIT_SLOTS", "512"))
SPLIT_MIN_KTILES = int(__import__("os").environ.get("PNMN_PLAN_SPLIT_MIN", "8"))
#: an encoder's two LSTM layers as a wavefront, independent encoder passes in one launch (pnmn_lstm_stack_*); False: a launch
#: per layer with the input projection as a GEMM in between (A/B aid, and what batches too large for the chip fall back to)
USE_STACK = True
#: encoder passes per wavefront launch, forward / backward (0: a launch per layer), and decoder passes per backward launch --
#: A/B aids (env PNMN_PLAN_FWD_ENC / PNMN_PLAN_BWD_ENC / PNMN_PLAN_DEC_GROUP); the defaults are what measured best beside
#: the NMN trunk at 128 questions (DESIGN 5 "Round 6")
import os as _os
STACK_FWD_ENCODERS = int(_os.environ.get("PNMN_PLAN_FWD_ENC", "2"))
STACK_PG_ENCODER = int(_os.environ.get("PNMN_PLAN_PG_ENC", "1"))
STACK_BWD_ENCODERS = int(_os.environ.get("PNMN_PLAN_BWD_ENC", "0"))
DECODER_BWD_GROUP = int(_os.environ.get("PNMN_PLAN_DEC_GROUP", "3"))
#: workgroups a GEMM launch of the plan may occupy (0: one per tile)
GEMM_WORKGROUPS = int(_os.environ.get("PNMN_PLAN_GEMM_WGS", "0"))
#: parameter gradients of the passes on an auxiliary stream beside the backward chains (0: on the one stream, at the end)
USE_AUX_STREAM = _os.environ.get("PNMN_PLAN_AUX", "0") != "0"
_AUX_STREAMS: Dict = {}


def _aux_stream(dev: torch.device) -> "torch.cuda.Stream":
    """ONE auxiliary stream per device for every plan of the process (HIP multiplexes a process's streams onto four hardware
    queues: every further stream shifts which of them share one -- see trainers.joint_training.shared_stream)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _AUX_STREAMS:
        _AUX_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _AUX_STREAMS[key]


class PlanUnsupported(Exception):
    """The models / shapes are outside what the plan is built for: the trainer keeps the eager passes."""


class _Calls(list):
    def add(self, name: str, *args) -> None:
        self.append((getattr(_hip.lib(), name), args, name))

    def run(self) -> None:
        for fn, args, name in self:
            rc = fn(*args)
            if rc:
                _hip.check(rc, name)


def _supported(model) -> bool:
    lstm, cell = model._encoder._module, model._decoder_cell
    return (lstm.hidden_size == 256 and lstm.num_layers == 2 and lstm.input_size == 256 and cell.hidden_size == 256
            and cell.input_size == 512 and model._output_projection_layer.weight.size(0) <= 128
            and model._source_embedder.embedding.weight.size(0) <= 128)


class _Model:
    """Parameter handles of one Seq2SeqBase in the plan's terms, and its flat gradient buffer."""

    def __init__(self, model, dev):
        self.model = model
        lstm, cell, proj = model._encoder._module, model._decoder_cell, model._output_projection_layer
        self.emb_src, self.emb_tgt = model._source_embedder.embedding.weight, model._target_embedder.weight
        self.lstm, self.cell, self.proj = lstm, cell, proj
        self.params = [self.emb_src] + [getattr(lstm, "%s_l%d" % (n, layer)) for layer in (0, 1)
                                        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] \
            + [self.emb_tgt, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, proj.weight, proj.bias]
        for p in self.params:
            if not p.is_contiguous() or p.device != dev or p.dtype != torch.float32:
                raise PlanUnsupported("parameter layout")
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = [self.gflat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)]
        self.g = {id(p): g for p, g in zip(self.params, self.grads)}
        self.signature = tuple(p.data_ptr() for p in self.params)

    def grad(self, p) -> torch.Tensor:
        return self.g[id(p)]

    def attach(self) -> None:
        for p, g in zip(self.params, self.grads):
            p.grad = g


class Seq2SeqPlan:
    """See the module docstring.  ``n`` unsupervised (sampled) rows, ``m`` supervised rows, both > 0; ``tq`` / ``tp``: the
    widths of the batch's question / program matrices."""

    def __init__(self, pg, qr, prior, dev: torch.device, n: int, m: int, tq: int, tp: int, with_prior: bool = True):
        if not (_supported(pg) and _supported(qr)) or n <= 0 or m <= 0:
            raise PlanUnsupported("model shapes")
        if with_prior:
            pl = prior._encoder._module
            if pl.hidden_size != 256 or pl.num_layers != 2 or pl.input_size != 256:
                raise PlanUnsupported("prior shapes")
        lib = _hip.lib()
        self.dev, self.n, self.m, self.tq, self.tp = dev, n, m, tq, tp
        self.stream = _hip.stream_ptr(dev)
        self.pg, self.qr, self.prior = _Model(pg, dev), _Model(qr, dev), prior
        self.with_prior = with_prior
        B = n + m
        self.B = B
        D = pg._max_decoding_steps
        self.D = D
        if int(lib.pnmn_attn_lstm_pair_workspace_bytes(n, m, 0)) <= 0 or int(lib.pnmn_attn_lstm_multi_workspace_bytes(B, 0)) <= 0 \
                or int(lib.pnmn_lstm_seq_workspace_bytes(B, 0)) <= 0:
            raise PlanUnsupported("batch does not fit the multi-CU recurrent kernels in one launch")
        self._bufs: Dict[str, torch.Tensor] = {}
        self._keep: List = []  # numpy records the call tuples point into
        self.anchor = torch.zeros((), device=dev, requires_grad=True)
        self.fwd_pg_enc, self.fwd_pg, self.fwd_qr, self.fwd_prior, self.bwd = _Calls(), _Calls(), _Calls(), _Calls(), _Calls()
        self._derived_sig = None
        self._build()

    # ---- workspace ---------------------------------------------------------------------------------------------------------
    def buf(self, name: str, *shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        if name in self._bufs:
            raise KeyError(name)
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
        self._bufs[name] = t
        return t

    def bytes_buf(self, name: str, nbytes: int, zero: bool = False) -> torch.Tensor:
        return self.buf(name, max(int(nbytes), 16), dtype=torch.uint8, zero=zero)

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._bufs[name]

    # ---- pieces ------------------------------------------------------------------------------------------------------------
    def _gemm(self, calls: _Calls, name: str, descs) -> None:
        """descs: list of dicts(a, b, c, M, N, K, lda, ldb, ldc, ta, tb, acc, bias, split, shift_t, h0, ld_h0, colsum, colsum2)."""
        lib = _hip.lib()
        for lo in range(0, len(descs), _hip.GEMM_MAX):
            part = descs[lo:lo + _hip.GEMM_MAX]
            rec = np.zeros(len(part), _hip.GEMM_DESC)
            # split-K of the problems that ask for it ("auto"), chosen for the LAUNCH: chunks of one common length (in 32-wide
            # k tiles) for every problem of the launch, so that the launch is a few ROUNDS of equal workgroups over the chip's
            # 512 slots (two 64 KB workgroups per CU) -- a chunk count per problem that merely filled the chip once left the CUs
            # holding two long chunks running after the others had finished (eight weight gradients of 512-question passes:
            # 78 TFLOP/s).  The length is the one with the least modelled time among 8 .. 128 k tiles: rounds x chunk length,
            # a started last round counting 0.6 (its workgroups have their CU to themselves), plus the partial tiles' trip
            # through memory (64 KB written and read back per chunk: ~4 k tiles' worth of a workgroup's time).  2114 equal
            # chunks are 4.13 rounds -- five in practice; 2040 are four.
            tiles = [((d["M"] + 127) // 128) * ((d["N"] + 127) // 128) for d in part]
            ktiles = [(d["K"] + 31) // 32 for d in part]
            auto = [i for i, d in enumerate(part) if d.get("split") == "auto"]
            fixed_units = sum(t for i, t in enumerate(tiles) if i not in auto)
            chunk_len, best = SPLIT_MIN_KTILES, None
            for cand in range(SPLIT_MIN_KTILES, 129):
                splits = [max(1, min(-(-ktiles[i] // cand), 64)) for i in auto]
                units = fixed_units + sum(tiles[i] * sp for i, sp in zip(auto, splits))
                longest = max([-(-ktiles[i] // sp) for i, sp in zip(auto, splits)] + [ktiles[i] for i in range(len(part)) if i not in auto] + [1])
                rounds = units // SPLIT_SLOTS + (0.6 if units % SPLIT_SLOTS else 0.0)
                cost = rounds * longest + 4.0 * sum(tiles[i] * sp for i, sp in zip(auto, splits) if sp > 1) / SPLIT_SLOTS
                if best is None or cost < best:
                    chunk_len, best = cand, cost
            for i, d in enumerate(part):
                r = rec[i]
                r["a"], r["b"], r["c"] = d["a"], d["b"], d["c"]
                r["lda"], r["ldb"], r["ldc"] = d["lda"], d["ldb"], d["ldc"]
                r["M"], r["N"], r["K"] = d["M"], d["N"], d["K"]
                r["flags"] = (_hip.GEMM_A_T if d.get("ta") else 0) | (_hip.GEMM_B_T if d.get("tb") else 0) | (_hip.GEMM_ACC if d.get("acc") else 0)
                r["bias"] = d.get("bias", 0)
                split = d.get("split", 1)
                if split == "auto":
                    split = max(1, min(-(-ktiles[i] // chunk_len), 64))
                r["split_k"] = split
                if split > 1:
                    ws = self.bytes_buf("%s.ws%d" % (name, lo + i), lib.pnmn_gemm_workspace_bytes(d["M"], d["N"], split), zero=True)
                    r["workspace"] = ws.data_ptr()
                r["shift_t"], r["shift_h0"], r["ld_h0"] = d.get("shift_t", 0), d.get("h0", 0), d.get("ld_h0", 0)
                r["colsum"], r["colsum2"] = d.get("colsum", 0), d.get("colsum2", 0)
            self._keep.append(rec)
            calls.add("pnmn_gemm_cus", rec.ctypes.data, len(rec), GEMM_WORKGROUPS, self.stream)

    def _encoder_prepare(self, calls: _Calls, tag: str, mm: Optional[_Model], derived, tokens: torch.Tensor, width: int, rows: int,
                         drop_first: bool, emb: torch.Tensor, pad_idx, lstm, want_last: bool = True) -> Dict:
        """Buffers of one encoder pass + what precedes its recurrence: token_prep and the per-token table of layer 1."""
        st = self.stream
        model = mm.model if mm is not None else self.prior
        pad, bos, eos = model._pad_index, model._start_index, model._end_index
        T = width + 2 - int(drop_first)
        V = emb.size(0)
        f = self.buf
        e = dict(tag=tag, mm=mm, derived=derived, lstm=lstm, T=T, rows=rows, V=V, pad_idx=-1 if pad_idx is None else pad_idx, emb=emb,
                 want_last=want_last,
                 src=f(tag + ".src", rows, T, dtype=torch.long), fmask=f(tag + ".fmask", rows, T), last=f(tag + ".last", rows, dtype=torch.int32),
                 table=f(tag + ".table", V, 1024), hs1=f(tag + ".hs1", rows, T, 256), cs1=f(tag + ".cs1", rows, T, 256),
                 act1=f(tag + ".act1", rows, T, 1024), hs2=f(tag + ".hs2", rows, T, 256), cs2=f(tag + ".cs2", rows, T, 256),
                 act2=f(tag + ".act2", rows, T, 1024))
        calls.add("pnmn_token_prep", tokens.data_ptr(), tokens.stride(0), rows, width, pad, bos, eos, int(drop_first), e["src"].data_ptr(),
                  e["fmask"].data_ptr(), e["last"].data_ptr(), st)
        calls.add("pnmn_token_table_fwd", emb.data_ptr(), lstm.weight_ih_l0.data_ptr(), lstm.weight_ih_l0.stride(0),
                  derived["l0.b"].data_ptr(), V, 256, 1024, e["table"].data_ptr(), st)
        if want_last:
            e["enc"], e["h"] = f(tag + ".enc", rows, T, 256), f(tag + ".h", rows, 256)
        return e

    def _stack_jobs(self, encs: List[Dict], backward: bool) -> np.ndarray:
        jobs = np.zeros(2 * len(encs), _hip.LSTM_STACK_JOB)
        for k, e in enumerate(encs):
            d = e["derived"]
            if not backward:
                a, b = jobs[2 * k], jobs[2 * k + 1]
                a["xp"], a["tokens"], a["token_stride"], a["w_hh"] = e["table"].data_ptr(), e["src"].data_ptr(), e["src"].stride(0), d["l0.hh"].data_ptr()
                a["hs"], a["cs"], a["act"], a["dep"] = e["hs1"].data_ptr(), e["cs1"].data_ptr(), e["act1"].data_ptr(), -1
                b["w_hh"], b["w_ih"], b["bias"] = d["l1.hh"].data_ptr(), d["l1.ih"].data_ptr(), d["l1.b"].data_ptr()
                b["hs"], b["cs"], b["act"], b["dep"] = e["hs2"].data_ptr(), e["cs2"].data_ptr(), e["act2"].data_ptr(), 2 * k
            else:  # layer 2 first (top), layer 1 below it
                a, b = jobs[2 * k], jobs[2 * k + 1]
                a["dhs"], a["act"], a["cs"], a["w_hh"], a["dgates"], a["dep"] = (e["dhs2"].data_ptr(), e["act2"].data_ptr(), e["cs2"].data_ptr(),
                                                                                 d["l1.hhT"].data_ptr(), e["dg2"].data_ptr(), -1)
                b["act"], b["cs"], b["w_hh"], b["w_ih"], b["dgates"], b["dep"] = (e["act1"].data_ptr(), e["cs1"].data_ptr(), d["l0.hhT"].data_ptr(),
                                                                                 d["l1.ihT"].data_ptr(), e["dg1"].data_ptr(), 2 * k)
            for j in (a, b):
                j["B"], j["T"] = e["rows"], e["T"]
        return jobs

    def _stack_launches(self, calls: _Calls, name: str, encs: List[Dict], backward: bool, most: Optional[int] = None) -> List[Dict]:
        """The recurrences of several encoder passes as wavefront launches: all of them in one launch when their workgroups
        fit the chip together, else greedily as many as fit per launch; returns the encoders no stack launch takes (their
        layers go out one by one)."""
        lib, left, group = _hip.lib(), [], []
        if most is None:
            most = STACK_BWD_ENCODERS if backward else STACK_FWD_ENCODERS
        use = USE_STACK and most > 0 and all("l1.ih" in e["derived"] for e in encs)

        def flush():
            if not group:
                return
            jobs = self._stack_jobs(group, backward)
            ws = self.bytes_buf("%s.stack_ws%d" % (name, len(self._keep)), lib.pnmn_lstm_stack_workspace_bytes(jobs.ctypes.data, len(jobs), int(backward)))
            self._keep.append(jobs)
            calls.add("pnmn_lstm_stack_bwd" if backward else "pnmn_lstm_stack_fwd", jobs.ctypes.data, len(jobs), ws.data_ptr(), self.stream)
            del group[:]

        for e in encs:
            if not use:
                left.append(e)
                continue
            trial = self._stack_jobs(group + [e], backward)
            if len(group) < most and 2 * (len(group) + 1) <= _hip.LSTM_STACK_JOBS and int(lib.pnmn_lstm_stack_workspace_bytes(trial.ctypes.data, len(trial), int(backward))) > 0:
                group.append(e)
                continue
            flush()
            alone = self._stack_jobs([e], backward)
            if int(lib.pnmn_lstm_stack_workspace_bytes(alone.ctypes.data, 2, int(backward))) > 0:
                group.append(e)
            else:
                left.append(e)
        flush()
        return left

    def _encoders_fwd(self, calls: _Calls, name: str, encs: List[Dict], most: Optional[int] = None) -> None:
        lib, st = _hip.lib(), self.stream
        for e in self._stack_launches(calls, name, encs, False, most):
            d, lstm, rows, T = e["derived"], e["lstm"], e["rows"], e["T"]
            xp2 = self.buf(e["tag"] + ".xp2", rows, T, 1024)
            ws = self.bytes_buf(e["tag"] + ".lstm_ws", lib.pnmn_lstm_seq_workspace_bytes(rows, 0))
            calls.add("pnmn_lstm_seq_fwd", e["table"].data_ptr(), e["src"].data_ptr(), e["src"].stride(0), d["l0.hh"].data_ptr(),
                      e["hs1"].data_ptr(), e["cs1"].data_ptr(), e["act1"].data_ptr(), rows, T, 256, ws.data_ptr(), st)
            self._gemm(calls, e["tag"] + ".xp2g", [dict(a=e["hs1"].data_ptr(), b=lstm.weight_ih_l1.data_ptr(), c=xp2.data_ptr(), M=rows * T,
                                                        N=1024, K=256, lda=256, ldb=256, ldc=1024, tb=1, bias=d["l1.b"].data_ptr())])
            calls.add("pnmn_lstm_seq_fwd", xp2.data_ptr(), None, 0, d["l1.hh"].data_ptr(), e["hs2"].data_ptr(), e["cs2"].data_ptr(),
                      e["act2"].data_ptr(), rows, T, 256, ws.data_ptr(), st)
        for e in encs:
            if e["want_last"]:
                calls.add("pnmn_mask_last_fwd", e["hs2"].data_ptr(), e["fmask"].data_ptr(), e["last"].data_ptr(), e["rows"], e["T"], 256,
                          e["enc"].data_ptr(), e["h"].data_ptr(), st)

    def _encoders_bwd(self, calls: _Calls, name: str, encs: List[Dict]) -> None:
        """The encoders' backward CHAIN.  encs: encoder dicts with "denc" / "dh" set (gradients of the masked outputs and of
        the last states)."""
        lib, st = _hip.lib(), self.stream
        f = self.buf
        for e in encs:
            tag, rows, T = e["tag"], e["rows"], e["T"]
            e["dhs2"], e["dg2"], e["dg1"] = f(tag + ".dhs2", rows, T, 256), f(tag + ".dg2", rows, T, 1024), f(tag + ".dg1", rows, T, 1024)
            calls.add("pnmn_mask_last_bwd", e["denc"].data_ptr(), e["dh"].data_ptr(), e["fmask"].data_ptr(), e["last"].data_ptr(), rows, T, 256,
                      e["dhs2"].data_ptr(), st)
        for e in self._stack_launches(calls, name, encs, True):
            d, lstm, tag, rows, T = e["derived"], e["lstm"], e["tag"], e["rows"], e["T"]
            dhs1 = f(tag + ".dhs1", rows, T, 256)
            ws = self.bytes_buf(tag + ".lstm_bws", lib.pnmn_lstm_seq_workspace_bytes(rows, 1))
            calls.add("pnmn_lstm_seq_bwd", e["dhs2"].data_ptr(), e["act2"].data_ptr(), e["cs2"].data_ptr(), d["l1.hhT"].data_ptr(),
                      e["dg2"].data_ptr(), rows, T, 256, ws.data_ptr(), st)
            self._gemm(calls, tag + ".dx", [dict(a=e["dg2"].data_ptr(), b=lstm.weight_ih_l1.data_ptr(), c=dhs1.data_ptr(), M=rows * T, N=256,
                                                 K=1024, lda=1024, ldb=256, ldc=256, split="auto")])
            calls.add("pnmn_lstm_seq_bwd", dhs1.data_ptr(), e["act1"].data_ptr(), e["cs1"].data_ptr(), d["l0.hhT"].data_ptr(),
                      e["dg1"].data_ptr(), rows, T, 256, ws.data_ptr(), st)

    def _encoders_param_grads(self, calls: _Calls, deferred: List, encs: List[Dict]) -> None:
        """The encoders' parameter gradients from what the chain left behind (dgates of both layers): the table's rows and the
        bias sums as launches, everything GEMM-shaped appended to ``deferred``."""
        lib, st = _hip.lib(), self.stream
        f = self.buf
        for e in encs:
            mm, lstm, tag, rows, T, V = e["mm"], e["lstm"], e["tag"], e["rows"], e["T"], e["V"]
            dtable = f(tag + ".dtable", V, 1024)
            ews = self.bytes_buf(tag + ".emb_ws", lib.pnmn_embedding_grad_workspace_bytes(rows, T, V))
            dg1, dg2, g = e["dg1"], e["dg2"], mm.grad
            calls.add("pnmn_embedding_grad", dg1.data_ptr(), e["src"].data_ptr(), e["src"].stride(0), rows, T, 1024, V, 0, 0, -1, 0,
                      dtable.data_ptr(), ews.data_ptr(), st)
            calls.add("pnmn_token_table_bwd", dtable.data_ptr(), e["emb"].data_ptr(), lstm.weight_ih_l0.data_ptr(), lstm.weight_ih_l0.stride(0),
                      V, 256, 1024, e["pad_idx"], g(e["emb"]).data_ptr(), g(lstm.weight_ih_l0).data_ptr(), 0, g(lstm.bias_ih_l0).data_ptr(),
                      g(lstm.bias_hh_l0).data_ptr(), st)
            K = rows * T
            # (layer 2's bias gradients = the column sums of dgates2: the weight-gradient product that reads dgates2 as its
            # transposed operand adds them up on the way -- pnmn_gemm_desc.colsum)
            deferred += [
                dict(a=dg2.data_ptr(), b=e["hs2"].data_ptr(), c=g(lstm.weight_hh_l1).data_ptr(), M=1024, N=256, K=K, lda=1024, ldb=256,
                     ldc=256, ta=1, split="auto", shift_t=T, colsum=g(lstm.bias_ih_l1).data_ptr(), colsum2=g(lstm.bias_hh_l1).data_ptr()),
                dict(a=dg2.data_ptr(), b=e["hs1"].data_ptr(), c=g(lstm.weight_ih_l1).data_ptr(), M=1024, N=256, K=K, lda=1024, ldb=256,
                     ldc=256, ta=1, split="auto"),
                dict(a=dg1.data_ptr(), b=e["hs1"].data_ptr(), c=g(lstm.weight_hh_l0).data_ptr(), M=1024, N=256, K=K, lda=1024, ldb=256,
                     ldc=256, ta=1, split="auto", shift_t=T),
            ]

    def _decoder_side(self, tag: str, rows: int, T: int, S: int, base: Dict[str, torch.Tensor], row0: int) -> Dict[str, torch.Tensor]:
        """Views of a model's concatenated decoder buffers for one pass: rows*T sequence rows starting at flat row ``row0``."""
        v = {}
        R = rows * T
        for k, w in (("hs", 256), ("cs", 256), ("cx", 256), ("act", 1024), ("dhs", 256), ("dg", 1024), ("dctx", 256)):
            v[k] = base[k][row0:row0 + R].view(rows, T, w)
        for k in ("probs", "dscore", "weights"):
            v[k] = self.buf("%s.%s" % (tag, k), rows, T, S)
        v.update(rows=rows, T=T, S=S, row0=row0, R=R)
        return v

    # ---- the plan ----------------------------------------------------------------------------------------------------------
    def _build(self) -> None:
        lib, st = _hip.lib(), self.stream
        n, m, B, D, tq, tp = self.n, self.m, self.B, self.D, self.tq, self.tp
        pg, qr = self.pg, self.qr
        dpg, dqr = pg.model._derived(), qr.model._derived()
        if dpg is None or dqr is None:
            raise PlanUnsupported("derived parameters")
        f = self.buf
        pad, bos, eos = pg.model._pad_index, pg.model._start_index, pg.model._end_index
        # ---- generator: encoder over [unsupervised ; supervised] questions -------------------------------------------------
        ques = f("ques", B, tq, dtype=torch.long)
        prog_sup = f("prog_sup", m, tp, dtype=torch.long)
        e_pg = self._encoder_prepare(self.fwd_pg_enc, "pg.e", pg, dpg, ques, tq, B, True, pg.emb_src,
                                     pg.model._source_embedder.embedding.padding_idx, pg.lstm)
        self._encoders_fwd(self.fwd_pg_enc, "pg.e", [e_pg], most=STACK_PG_ENCODER)
        #: workgroups of the generator's encoder launch (0: a launch per layer): a trainer that runs the NMN's stem beside it
        #: cuts the stem's launches for the CUs this leaves (JointTrainingStep)
        self.pg_encoder_workgroups = 16 * (-(-B // 16)) if any(n == "pnmn_lstm_stack_fwd" for _, _, n in self.fwd_pg_enc) else 0
        S = e_pg["T"]
        if S > 64 or D > 64:
            raise PlanUnsupported("more than 64 source positions / decoding steps")
        # ---- generator: sampling decode of rows [0, n) and teacher-forced decode of rows [n, B) in one launch ----------------
        Tt = tp + 1  # teacher-forced steps over [@start@, program, @end@]
        Rs, Rt = n * D, m * Tt
        Vp = pg.proj.weight.size(0)
        tgt = f("pg.tgt", m, tp + 2, dtype=torch.long)
        table_d = f("pg.d.table", Vp, 1024)
        base = {k: f("pg.d." + k, Rs + Rt, w) for k, w in (("hs", 256), ("cs", 256), ("cx", 256), ("act", 1024), ("dhs", 256),
                                                           ("dg", 1024), ("dctx", 256))}
        side_s = self._decoder_side("pg.s", n, D, S, base, 0)
        side_t = self._decoder_side("pg.t", m, Tt, S, base, Rs)
        raw, z = f("pg.raw", n, D, dtype=torch.long), f("pg.z", n, D, dtype=torch.long)
        logits, dlogits = f("pg.logits", Rs + Rt, Vp), f("pg.dlogits", Rs + Rt, Vp)
        loss_s, loss_t = f("pg.loss_s", n), f("pg.loss_t", m)
        lse_s, lse_t = f("pg.lse_s", n, D), f("pg.lse_t", m, Tt)
        c = self.fwd_pg
        c.add("pnmn_token_prep", prog_sup.data_ptr(), prog_sup.stride(0), m, tp, pad, bos, eos, 0, tgt.data_ptr(), None, None, st)
        cell = pg.cell
        w_e_ptr = cell.weight_ih.data_ptr() + 4 * 256  # columns [256, 512): the embedding half of cat(attended, embedded)
        c.add("pnmn_token_table_fwd", pg.emb_tgt.data_ptr(), w_e_ptr, 512, dpg["d.b"].data_ptr(), Vp, 256, 1024, table_d.data_ptr(), st)
        jobs = np.zeros(2, _hip.DECODER_FWD_JOB)
        enc, fmask, h = e_pg["enc"], e_pg["fmask"], e_pg["h"]
        for j, sd, r0 in ((jobs[0], side_s, 0), (jobs[1], side_t, n)):
            j["etable"], j["enc"], j["mask"], j["h0"] = table_d.data_ptr(), enc[r0:].data_ptr(), fmask[r0:].data_ptr(), h[r0:].data_ptr()
            j["w_c"], j["w_hh"] = dpg["d.c"].data_ptr(), dpg["d.hh"].data_ptr()
            j["hs"], j["cs"], j["act"], j["ctx"], j["probs"] = (sd[k].data_ptr() for k in ("hs", "cs", "act", "cx", "probs"))
            j["B"], j["T"], j["S"], j["start_index"] = sd["rows"], sd["T"], S, bos
        js = jobs[0]
        js["w_p"], js["b_p"], js["tokens"], js["V"], js["sample"] = pg.proj.weight.data_ptr(), pg.proj.bias.data_ptr(), raw.data_ptr(), Vp, 1
        js["pad_index"], js["unk_index"] = pad, pg.model._unk_index
        jobs[1]["in_tokens"], jobs[1]["in_token_stride"] = tgt.data_ptr(), tgt.stride(0)
        self.pair_jobs = jobs
        pws = self.bytes_buf("pg.pair_ws", lib.pnmn_attn_lstm_pair_workspace_bytes(n, m, 0))
        c.add("pnmn_attn_lstm_fwd_multi_pair", jobs[0:1].ctypes.data, jobs[1:2].ctypes.data, 256, pws.data_ptr(), st)
        c.add("pnmn_trim_predictions", raw.data_ptr(), n, D, eos, z.data_ptr(), st)
        # (the output projection and the losses follow in `fwd_pg_finish`: the samples are what the host waits for)
        self.fwd_pg_finish = _Calls()
        c = self.fwd_pg_finish
        self._gemm(c, "pg.logits_g", [dict(a=base["hs"].data_ptr(), b=pg.proj.weight.data_ptr(), c=logits.data_ptr(), M=Rs + Rt, N=Vp,
                                           K=256, lda=256, ldb=256, ldc=Vp, tb=1, bias=pg.proj.bias.data_ptr())])
        c.add("pnmn_seq_nll_fwd", logits.data_ptr(), D * Vp, raw.data_ptr(), D, z.data_ptr(), D, pad, loss_s.data_ptr(), lse_s.data_ptr(),
              n, D, Vp, 1e-12, st)
        lt = logits[Rs:]
        c.add("pnmn_seq_nll_fwd", lt.data_ptr(), Tt * Vp, tgt.data_ptr() + 8, tgt.stride(0), tgt.data_ptr() + 8, tgt.stride(0), pad,
              loss_t.data_ptr(), lse_t.data_ptr(), m, Tt, Vp, 1e-13, st)
        # ---- reconstructor: encoder over [sampled ; ground-truth] programs, teacher-forced decode over the questions ----------
        Wq = max(D, tp)
        source = f("qr.source", B, Wq, dtype=torch.long)
        segs = np.zeros(2, _hip.TOKEN_SEG)
        segs[0]["src"], segs[0]["row_stride"], segs[0]["rows"], segs[0]["width"] = z.data_ptr(), D, n, D
        segs[1]["src"], segs[1]["row_stride"], segs[1]["rows"], segs[1]["width"] = prog_sup.data_ptr(), tp, m, tp
        self._keep.append(segs)
        c = self.fwd_qr
        c.add("pnmn_token_rows", segs.ctypes.data, 2, source.data_ptr(), Wq, 0, st)
        e_qr = self._encoder_prepare(c, "qr.e", qr, dqr, source, Wq, B, True, qr.emb_src, qr.model._source_embedder.embedding.padding_idx,
                                     qr.lstm)
        # the prior reads the same samples: its two LSTM layers ride in the reconstructor encoder's launch (its projections
        # and loss follow in `fwd_prior`, behind the trunk's launch)
        e_pr = None
        if self.with_prior:
            pr = self.prior
            dpr = pr._derived()
            if dpr is None:
                raise PlanUnsupported("prior derived parameters")
            e_pr = self._encoder_prepare(c, "pr.e", None, dpr, z, D, n, False, pr._embedder.embedding.weight,
                                         pr._embedder.embedding.padding_idx, pr._encoder._module, want_last=False)
        self._encoders_fwd(c, "qr.e", [e_qr] + ([e_pr] if e_pr is not None else []))
        Sq = e_qr["T"]
        Tq = tq + 1
        if Sq > 64 or Tq > 64:
            raise PlanUnsupported("more than 64 positions in the reconstructor")
        Vq = qr.proj.weight.size(0)
        qtgt = f("qr.tgt", B, tq + 2, dtype=torch.long)
        table_q = f("qr.d.table", Vq, 1024)
        qbase = {k: f("qr.d." + k, B * Tq, w) for k, w in (("hs", 256), ("cs", 256), ("cx", 256), ("act", 1024), ("dhs", 256), ("dg", 1024),
                                                            ("dctx", 256))}
        side_q = self._decoder_side("qr.q", B, Tq, Sq, qbase, 0)
        qlogits, qdlogits = f("qr.logits", B * Tq, Vq), f("qr.dlogits", B * Tq, Vq)
        loss_q, lse_q = f("qr.loss", B), f("qr.lse", B, Tq)
        c.add("pnmn_token_prep", ques.data_ptr(), ques.stride(0), B, tq, pad, bos, eos, 0, qtgt.data_ptr(), None, None, st)
        qcell = qr.cell
        c.add("pnmn_token_table_fwd", qr.emb_tgt.data_ptr(), qcell.weight_ih.data_ptr() + 4 * 256, 512, dqr["d.b"].data_ptr(), Vq, 256, 1024,
              table_q.data_ptr(), st)
        qws = self.bytes_buf("qr.dec_ws", lib.pnmn_attn_lstm_multi_workspace_bytes(B, 0))
        c.add("pnmn_attn_lstm_fwd_multi", None, table_q.data_ptr(), e_qr["enc"].data_ptr(), e_qr["fmask"].data_ptr(), e_qr["h"].data_ptr(),
              dqr["d.c"].data_ptr(), dqr["d.hh"].data_ptr(), None, None, side_q["hs"].data_ptr(), side_q["cs"].data_ptr(),
              side_q["act"].data_ptr(), side_q["cx"].data_ptr(), side_q["probs"].data_ptr(), None, B, Tq, Sq, 0, 256, 0, pad,
              qr.model._unk_index, bos, 0, 0, qtgt.data_ptr(), qtgt.stride(0), qws.data_ptr(), st)
        self._gemm(c, "qr.logits_g", [dict(a=qbase["hs"].data_ptr(), b=qr.proj.weight.data_ptr(), c=qlogits.data_ptr(), M=B * Tq, N=Vq, K=256,
                                           lda=256, ldb=256, ldc=Vq, tb=1, bias=qr.proj.bias.data_ptr())])
        c.add("pnmn_seq_nll_fwd", qlogits.data_ptr(), Tq * Vq, qtgt.data_ptr() + 8, qtgt.stride(0), qtgt.data_ptr() + 8, qtgt.stride(0), pad,
              loss_q.data_ptr(), lse_q.data_ptr(), B, Tq, Vq, 1e-13, st)
        # ---- prior: LSTM language model over the samples (no gradient: its loss only enters the detached reward) ---------------
        if self.with_prior:
            c = self.fwd_prior
            emb = pr._embedder.embedding.weight
            Vz = emb.size(0)
            Tz = e_pr["T"]
            proj, plog = f("pr.proj", n * Tz, 256), f("pr.logits", n * Tz, Vz)
            loss_p, lse_p = f("pr.loss", n), f("pr.lse", n, Tz - 1)
            # (the padded steps are not zeroed as PytorchSeq2SeqWrapper does: they only meet padded targets, whose weight is zero)
            self._gemm(c, "pr.proj_g", [dict(a=e_pr["hs2"].data_ptr(), b=pr._projection_layer.weight.data_ptr(), c=proj.data_ptr(), M=n * Tz,
                                             N=256, K=256, lda=256, ldb=256, ldc=256, tb=1)])
            self._gemm(c, "pr.out_g", [dict(a=proj.data_ptr(), b=emb.data_ptr(), c=plog.data_ptr(), M=n * Tz, N=Vz, K=256, lda=256, ldb=256,
                                            ldc=Vz, tb=1)])
            src = e_pr["src"]
            c.add("pnmn_seq_nll_fwd", plog.data_ptr(), Tz * Vz, src.data_ptr() + 8, src.stride(0), src.data_ptr() + 8, src.stride(0),
                  pr._pad_index, loss_p.data_ptr(), lse_p.data_ptr(), n, Tz - 1, Vz, 1e-13, st)
            self.prior_sig = tuple(p.data_ptr() for p in pr.parameters())
        # ---- backward ------------------------------------------------------------------------------------------------------------
        c = self.bwd
        self.d_rows = {"s": f("d.loss_s", n), "t": f("d.loss_t", m), "q": f("d.loss_q", B)}
        c.add("pnmn_seq_nll_bwd", logits.data_ptr(), D * Vp, raw.data_ptr(), D, z.data_ptr(), D, pad, lse_s.data_ptr(),
              self.d_rows["s"].data_ptr(), dlogits.data_ptr(), D * Vp, n, D, Vp, 1e-12, st)
        c.add("pnmn_seq_nll_bwd", lt.data_ptr(), Tt * Vp, tgt.data_ptr() + 8, tgt.stride(0), tgt.data_ptr() + 8, tgt.stride(0), pad,
              lse_t.data_ptr(), self.d_rows["t"].data_ptr(), dlogits[Rs:].data_ptr(), Tt * Vp, m, Tt, Vp, 1e-13, st)
        c.add("pnmn_seq_nll_bwd", qlogits.data_ptr(), Tq * Vq, qtgt.data_ptr() + 8, qtgt.stride(0), qtgt.data_ptr() + 8, qtgt.stride(0), pad,
              lse_q.data_ptr(), self.d_rows["q"].data_ptr(), qdlogits.data_ptr(), Tq * Vq, B, Tq, Vq, 1e-13, st)
        self._gemm(c, "dhs_g", [
            dict(a=dlogits.data_ptr(), b=pg.proj.weight.data_ptr(), c=base["dhs"].data_ptr(), M=Rs + Rt, N=256, K=Vp, lda=Vp, ldb=256, ldc=256),
            dict(a=qdlogits.data_ptr(), b=qr.proj.weight.data_ptr(), c=qbase["dhs"].data_ptr(), M=B * Tq, N=256, K=Vq, lda=Vq, ldb=256, ldc=256)])
        # the three decoders' backward in one launch
        denc_pg, dh_pg = f("pg.denc", B, S, 256), f("pg.dh", B, 256)
        denc_qr, dh_qr = f("qr.denc", B, Sq, 256), f("qr.dh", B, 256)
        bj = np.zeros(3, _hip.DECODER_BWD_JOB)
        for j, sd, mdl, der, e, r0, dh in ((bj[0], side_s, pg, dpg, e_pg, 0, dh_pg), (bj[1], side_t, pg, dpg, e_pg, n, dh_pg),
                                           (bj[2], side_q, qr, dqr, e_qr, 0, dh_qr)):
            j["dhs"], j["act"], j["cs"], j["hs"], j["probs"] = (sd[k].data_ptr() for k in ("dhs", "act", "cs", "hs", "probs"))
            j["enc"], j["mask"], j["h0"] = e["enc"][r0:].data_ptr(), e["fmask"][r0:].data_ptr(), e["h"][r0:].data_ptr()
            j["w_c_t"], j["w_hh_t"] = der["d.cT"].data_ptr(), der["d.hhT"].data_ptr()
            j["dgates"], j["dctx"], j["dscore"], j["weights"] = (sd[k].data_ptr() for k in ("dg", "dctx", "dscore", "weights"))
            j["dh0"] = dh[r0:].data_ptr()
            j["B"], j["T"], j["S"] = sd["rows"], sd["T"], sd["S"]
        self._keep.append(bj)
        if DECODER_BWD_GROUP >= 3:
            gws = self.bytes_buf("group_ws", lib.pnmn_attn_lstm_group3_workspace_bytes(n, m, B, 1))
            c.add("pnmn_attn_lstm_bwd_multi_group3", bj[0:1].ctypes.data, bj[1:2].ctypes.data, bj[2:3].ctypes.data, 256, gws.data_ptr(), st)
        else:
            singles = [0, 1, 2]
            if DECODER_BWD_GROUP == 2:
                gws = self.bytes_buf("pair_bws", lib.pnmn_attn_lstm_pair_workspace_bytes(n, m, 1))
                c.add("pnmn_attn_lstm_bwd_multi_pair", bj[0:1].ctypes.data, bj[1:2].ctypes.data, 256, gws.data_ptr(), st)
                singles = [2]
            for k in singles:
                j = bj[k]
                sws = self.bytes_buf("single_bws%d" % k, lib.pnmn_attn_lstm_multi_workspace_bytes(int(j["B"]), 1))
                c.add("pnmn_attn_lstm_bwd_multi", *(int(j[fld]) for fld in ("dhs", "act", "cs", "hs", "probs", "enc", "mask", "h0", "w_c_t", "w_hh_t",
                                                                            "dgates", "dctx", "dscore", "weights", "dh0")),
                      int(j["B"]), int(j["T"]), int(j["S"]), 256, sws.data_ptr(), st)
        for sd, e, r0, denc in ((side_s, e_pg, 0, denc_pg), (side_t, e_pg, n, denc_pg), (side_q, e_qr, 0, denc_qr)):
            c.add("pnmn_attn_denc", sd["weights"].data_ptr(), sd["dscore"].data_ptr(), sd["dctx"].data_ptr(), sd["hs"].data_ptr(),
                  e["h"][r0:].data_ptr(), denc[r0:].data_ptr(), sd["rows"], sd["T"], sd["S"], 256, st)
        # Parameter gradients leave the chain: nothing in backward waits for them, so they go to an auxiliary stream behind
        # events -- the decoders' right behind the decoder launch, the encoders' behind their layers -- and fill the CUs the
        # latency-bound chains (this one and the NMN trunk's on its stream) leave idle; every kernel there is one of this
        # library's that never waits for another workgroup.  PNMN_PLAN_AUX=0: all on the one stream, at the end.
        main_stream = self.stream
        self.bwd_a, self.bwd_b, self.aux_a, self.aux_b = c, _Calls(), _Calls(), _Calls()
        self.aux_stream = _aux_stream(self.dev) if USE_AUX_STREAM else None
        if self.aux_stream is not None:
            self.stream = self.aux_stream.cuda_stream
        ca = self.aux_a if self.aux_stream is not None else _Calls()
        deferred: List[dict] = []
        for mdl, der, tab, tag, sides in ((pg, dpg, table_d, "pg", ((side_s, raw, D, 1, 0), (side_t, tgt, tp + 2, 0, 0))),
                                          (qr, dqr, table_q, "qr", ((side_q, qtgt, tq + 2, 0, 0),))):
            V = tab.size(0)
            dtab = f(tag + ".d.dtable", V, 1024)
            g = mdl.grad
            for k, (sd, toks, tstride, shift, _) in enumerate(sides):
                ews = self.bytes_buf("%s.d.emb_ws%d" % (tag, k), lib.pnmn_embedding_grad_workspace_bytes(sd["rows"], sd["T"], V))
                ca.add("pnmn_embedding_grad", sd["dg"].data_ptr(), toks.data_ptr(), tstride, sd["rows"], sd["T"], 1024, V, shift, bos, -1,
                       1 if k else 0, dtab.data_ptr(), ews.data_ptr(), self.stream)
            w_ih = mdl.cell.weight_ih
            ca.add("pnmn_token_table_bwd", dtab.data_ptr(), mdl.emb_tgt.data_ptr(), w_ih.data_ptr() + 4 * 256, 512, V, 256, 1024, -1,
                   g(mdl.emb_tgt).data_ptr(), g(w_ih).data_ptr() + 4 * 256, 512, g(mdl.cell.bias_ih).data_ptr(), g(mdl.cell.bias_hh).data_ptr(),
                   self.stream)
            bs = base if mdl is pg else qbase
            R = bs["hs"].size(0)
            lg, dlg = (logits, dlogits) if mdl is pg else (qlogits, qdlogits)
            deferred.append(dict(a=dlg.data_ptr(), b=bs["hs"].data_ptr(), c=g(mdl.proj.weight).data_ptr(), M=V, N=256, K=R, lda=V, ldb=256,
                                 ldc=256, ta=1, split="auto", colsum=g(mdl.proj.bias).data_ptr()))
            deferred.append(dict(a=bs["dg"].data_ptr(), b=bs["cx"].data_ptr(), c=g(w_ih).data_ptr(), M=1024, N=256, K=R, lda=1024, ldb=256,
                                 ldc=512, ta=1, split="auto"))
            for k, (sd, _, _, _, _) in enumerate(sides):
                e, r0 = (e_pg, (0 if sd is side_s else n)) if mdl is pg else (e_qr, 0)
                deferred.append(dict(a=sd["dg"].data_ptr(), b=sd["hs"].data_ptr(), c=g(mdl.cell.weight_hh).data_ptr(), M=1024, N=256,
                                     K=sd["R"], lda=1024, ldb=256, ldc=256, ta=1, split="auto", shift_t=sd["T"],
                                     h0=e["h"][r0:].data_ptr(), ld_h0=256, acc=1 if k else 0))

        def flush(calls, name):
            # (an accumulating product must follow the product it adds to: keep them in different launches)
            first = [d for d in deferred if not d.get("acc")]
            second = [d for d in deferred if d.get("acc")]
            if first:
                self._gemm(calls, name, first)
            if second:
                self._gemm(calls, name + "2", second)
            del deferred[:]

        if self.aux_stream is not None:
            flush(ca, "wgrad_dec")
        # the encoders' chain on the main stream ...
        self.stream = main_stream
        e_pg["denc"], e_pg["dh"], e_qr["denc"], e_qr["dh"] = denc_pg, dh_pg, denc_qr, dh_qr
        self._encoders_bwd(self.bwd_b, "enc", [e_pg, e_qr])
        # ... their parameter gradients behind it
        if self.aux_stream is not None:
            self.stream = self.aux_stream.cuda_stream
            self._encoders_param_grads(self.aux_b, deferred, [e_pg, e_qr])
            flush(self.aux_b, "wgrad_enc")
            self.stream = main_stream
            self.events = [torch.cuda.Event() for _ in range(3)]
        else:
            self.bwd_b.extend(ca)
            self._encoders_param_grads(self.bwd_b, deferred, [e_pg, e_qr])
            flush(self.bwd_b, "wgrad")
        self.out = dict(z=z, raw=raw, loss_s=loss_s, loss_t=loss_t, loss_q=loss_q, loss_p=self._bufs.get("pr.loss"), ques=ques,
                        prog_sup=prog_sup)
        self._derived_sig = self._sig()

    def _sig(self):
        d = [self.pg.model._derived(), self.qr.model._derived()] + ([self.prior._derived()] if self.with_prior else [])
        return tuple(t.data_ptr() for dd in d for t in dd.values())

    def still_valid(self) -> bool:
        """Parameters, derived copies and the stream are where the plan's calls point (a ``.to()``, a re-pointed parameter
        or another current stream: rebuild).  Also refreshes the derived copies (one launch per model and optimiser step)."""
        if _hip.stream_ptr(self.dev) != self.stream:
            return False
        for mm in (self.pg, self.qr):
            if tuple(p.data_ptr() for p in mm.params) != mm.signature:
                return False
        if self.with_prior and tuple(p.data_ptr() for p in self.prior.parameters()) != self.prior_sig:
            return False
        return self._sig() == self._derived_sig

    # ---- one iteration -------------------------------------------------------------------------------------------------------
    def run_encoder(self, question: torch.Tensor, program: torch.Tensor, nosup_d: torch.Tensor, sup_d: torch.Tensor) -> None:
        """The row subsets [unsupervised ; supervised] of the batch's questions, the supervised rows' programs, and the
        generator's encoder over the questions."""
        lib, st = _hip.lib(), self.stream
        if question.dtype != torch.long or program.dtype != torch.long or question.stride(1) != 1 or program.stride(1) != 1:
            raise _hip.HipLibraryError("token matrices must be int64 with contiguous rows")
        segs = np.zeros(2, _hip.TOKEN_SEG)
        segs[0]["src"], segs[0]["index"], segs[0]["row_stride"], segs[0]["rows"], segs[0]["width"] = \
            question.data_ptr(), nosup_d.data_ptr(), question.stride(0), self.n, self.tq
        segs[1]["src"], segs[1]["index"], segs[1]["row_stride"], segs[1]["rows"], segs[1]["width"] = \
            question.data_ptr(), sup_d.data_ptr(), question.stride(0), self.m, self.tq
        _hip.check(lib.pnmn_token_rows(segs.ctypes.data, 2, self.out["ques"].data_ptr(), self.tq, 0, st), "token_rows")
        segs[0]["src"], segs[0]["index"], segs[0]["row_stride"], segs[0]["rows"], segs[0]["width"] = \
            program.data_ptr(), sup_d.data_ptr(), program.stride(0), self.m, self.tp
        _hip.check(lib.pnmn_token_rows(segs.ctypes.data, 1, self.out["prog_sup"].data_ptr(), self.tp, 0, st), "token_rows")
        # one seed per pass, drawn as Seq2SeqBase.decode_prepare draws them (sampling, teacher-forced, reconstructor): the
        # eager and the planned iteration sample the same programs from the same torch seed
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        torch.randint(0, 2 ** 62, (1,))
        torch.randint(0, 2 ** 62, (1,))
        self.pair_jobs[0]["seed"] = seed
        self.pair_jobs[0]["row_offset"] = self.pg.model.sample_row_offset
        self.fwd_pg_enc.run()

    def run_decoders(self) -> torch.Tensor:
        """The generator's sampling + teacher-forced decoder pair and the trim; returns the trimmed samples z [n, D] (a
        persistent buffer: overwritten by the next iteration)."""
        self.fwd_pg.run()
        return self.out["z"]

    def run_generator_finish(self) -> None:
        self.fwd_pg_finish.run()

    def run_reconstructor(self) -> None:
        self.fwd_qr.run()

    def run_prior(self) -> torch.Tensor:
        self.fwd_prior.run()
        return self.out["loss_p"]

    def losses(self):
        """(generator loss of the sampled rows [n], generator cross entropy of the supervised rows [m], reconstruction
        losses [n + m]) as outputs of the plan's single autograd node."""
        return _PlanNode.apply(self.anchor, self)

    def backward(self, d_s, d_t, d_q) -> None:
        for key, d in (("s", d_s), ("t", d_t), ("q", d_q)):
            if d is None:
                self.d_rows[key].zero_()
            else:
                self.d_rows[key].copy_(d)
        self.bwd_a.run()
        if self.aux_stream is not None:
            main = torch.cuda.current_stream(self.dev)
            e1, e2, e3 = self.events
            e1.record(main)
            self.aux_stream.wait_event(e1)
            self.aux_a.run()
            self.bwd_b.run()
            e2.record(main)
            self.aux_stream.wait_event(e2)
            self.aux_b.run()
            e3.record(self.aux_stream)
            main.wait_event(e3)  # (the optimiser and the gradient all-reduce read the gradients on this stream)
        else:
            self.bwd_b.run()
        self.pg.attach()
        self.qr.attach()


class _PlanNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, plan: Seq2SeqPlan):
        ctx.plan = plan
        o = plan.out
        return o["loss_s"].view_as(o["loss_s"]), o["loss_t"].view_as(o["loss_t"]), o["loss_q"].view_as(o["loss_q"])

    @staticmethod
    def backward(ctx, d_s, d_t, d_q):
        ctx.plan.backward(d_s, d_t, d_q)
        return None, None
