"""``Seq2SeqBase`` -- class surface of the reference's ``probnmn.modules.seq2seq_base`` (reference:
probnmn/modules/seq2seq_base.py:19-375), without AllenNLP.

The reference subclasses ``allennlp.models.encoder_decoders.SimpleSeq2Seq`` (AllenNLP 0.9.0, not
part of this build); the pieces it inherits are restated here from the reference's call sites
and SURVEY.md App. A: token embedder with a zero padding row, a 2-layer LSTM encoder over packed
sequences, dot-product attention with AllenNLP's ``masked_softmax``, an ``LSTMCell`` decoder fed
``cat(attended, embedded)``, a linear output projection.  Parameter names follow the reference's
``state_dict`` so released checkpoints load.

On the MI355X the recurrences are persistent hand-written kernels: one launch per LSTM layer over
the whole padded sequence (``pnmn_lstm_seq_{fwd,bwd}``) and one per decoding loop -- attention,
gates, cell and token choice of every step (``pnmn_attn_lstm_{fwd,bwd}[_multi]``), four or eight
workgroups sharing each 16-row tile (csrc/seq2seq.hip, decoder_multi.hip).  What can be batched over
time (input / output projections, losses, weight gradients) is library GEMMs over all steps; other
hidden sizes fall back to a GEMM per step plus the cell kernel (``pnmn_lstm_cell_{fwd,bwd}``,
``pnmn_sample_tokens``).  Everything that the reference does with per-row Python loops and ``.cpu()``
round trips (sentence boundaries, trimming at ``@end@``) is vectorised on the device: a forward pass
performs no host synchronisation.
"""
import os
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from probnmn import _hip
from probnmn.running_metrics import BLEU, Average


class _LSTMCellPointwise(torch.autograd.Function):
    """(gate pre-activations [B,4H], c_prev [B,H]) -> (h, c) on the gfx950 kernel."""

    @staticmethod
    def forward(ctx, gates, c_prev):
        if gates.device.type != "cuda":
            raise _hip.HipLibraryError("LSTM cell on %s: the HIP path needs a ROCm device (no CPU fallback)" % gates.device)
        gates, c_prev = gates.contiguous(), c_prev.contiguous()
        B, H4 = gates.shape
        Hd = H4 // 4
        h = torch.empty_like(c_prev)
        c = torch.empty_like(c_prev)
        act = torch.empty_like(gates)
        _hip.check(_hip.lib().pnmn_lstm_cell_fwd(gates.data_ptr(), c_prev.data_ptr(), h.data_ptr(), c.data_ptr(),
                                                 act.data_ptr(), B, Hd, _hip.stream_ptr(gates.device)), "lstm_cell_fwd")
        ctx.save_for_backward(act, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        act, c_prev, c = ctx.saved_tensors
        B, H4 = act.shape
        dgates = torch.empty_like(act)
        dc_prev = torch.empty_like(c)
        dh = dh.contiguous() if dh is not None else None  # (kept alive until after the launch)
        dc = dc.contiguous() if dc is not None else None
        dh_p = dh.data_ptr() if dh is not None else None
        dc_p = dc.data_ptr() if dc is not None else None
        _hip.check(_hip.lib().pnmn_lstm_cell_bwd(act.data_ptr(), c_prev.data_ptr(), c.data_ptr(), dh_p, dc_p,
                                                 dgates.data_ptr(), dc_prev.data_ptr(), B, H4 // 4,
                                                 _hip.stream_ptr(act.device)), "lstm_cell_bwd")
        return dgates, dc_prev


def pack_fragments(w: torch.Tensor) -> torch.Tensor:
    """[N][K] row-major -> MFMA-fragment order [N/16][K/16][64][4] (see include/probnmn_hip.h): one wave-wide
    operand load of the persistent LSTM / decoder kernels then reads 1 KiB contiguous."""
    n, k = w.shape
    return w.detach().reshape(n // 16, 16, k // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()


def lstm_cell_pointwise(gates, c_prev):
    return _LSTMCellPointwise.apply(gates, c_prev)


def wgrad_gemm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``a.t() @ b`` for tall operands (a [K, M], b [K, N], K = rows x time steps >> M, N).  A weight
    gradient has few output tiles (1024 x 256 -> 32 workgroups on 256 CUs) and a very long reduction;
    splitting K into chunks run as one batched GEMM fills the chip, the partial sums add up after."""
    K = a.size(0)
    if a.size(1) <= 128:
        # few output rows (an output projection's V = 44-100): more, shorter chunks -- the largest count up to 64 that divides
        # K and leaves >= 256 rows per chunk (no remainder GEMM: at 128 questions a call costs ~20 us of host time)
        chunks = max((c for c in range(1, 65) if K % c == 0 and K // c >= 256), default=1)
        if chunks < 4:
            chunks = min(64, K // 512)
    else:
        chunks = min(16, K // 2048)
        if chunks > 1 and K % chunks:  # (a count near it that divides K: no remainder GEMM + addition)
            chunks = max((c for c in range(chunks - 3, 17) if c > 1 and K % c == 0 and K // c >= 1024), default=chunks)
    if chunks <= 1 or not (a.is_contiguous() and b.is_contiguous()):
        return a.t() @ b
    kc = K // chunks
    main = kc * chunks
    out = torch.bmm(a[:main].view(chunks, kc, -1).transpose(1, 2), b[:main].view(chunks, kc, -1)).sum(0)
    if main < K:
        out = out + a[main:].t() @ b[main:]
    return out


class DerivedParams:
    """Per-model cache of what the recurrent kernels read instead of the raw parameters: MFMA-fragment copies
    of W_hh / W_c (and of their transposes, for the backward kernels) and the bias sums b_ih + b_hh.  All of
    them are produced by ONE ``pnmn_derive_params`` launch per model and optimiser step, where the straight-line
    code issued a permute + copy per weight and pass and an add per bias and pass (~25 launches per step).
    The cache is keyed on the parameters' storage, their version counters and ``probnmn.optim.parameter_epoch()``
    (the fused optimiser updates parameters through their pointers, which bumps no version counter).  Anything
    that changes a parameter behind autograd's back the same way -- writes through ``p.data`` or a raw pointer --
    must call ``probnmn.optim.parameters_changed()``; ``load_state_dict``, ``torch.optim`` steps, ``copy_`` /
    in-place ops under ``no_grad`` and ``.to(device)`` are all seen through the version counters / storage."""

    def __init__(self):
        self._key = None
        self._where = None
        self._out: Dict[str, torch.Tensor] = {}

    def get(self, params, make_specs) -> Dict[str, torch.Tensor]:
        """``params``: the parameters everything is derived from (the cache key is their storage, version
        counters and the fused optimiser's step count -- a dozen attribute reads per call);
        ``make_specs()``: list of (name, kind, a, b) -- kind "pack" (fragment order of the [N][K] matrix ``a``),
        "packT" (of its transpose), "sum" (a + b) -- built only when the cache misses."""
        from probnmn.optim import parameter_epoch

        flat = [parameter_epoch()]
        for p in params:
            flat.append(p._version)
            flat.append(p.data_ptr())
        key = tuple(flat)
        if key == self._key:
            return self._out
        where = key[2::2]
        if where == self._where:
            # same storage, new values (every optimiser step): the job list still describes the work -- run it
            # again into the same buffer (launches on this stream that read the old copies are ordered before it)
            _hip.check(_hip.lib().pnmn_derive_params(self._jobs.data_ptr(), self._n_jobs, self._max_quads,
                                                     _hip.stream_ptr(self._device)), "derive_params")
            self._key = key
            return self._out
        specs = make_specs()
        dev = specs[0][2].device
        import numpy as np

        total = sum(a.numel() for _, _, a, _ in specs)
        buf = torch.empty(total, dtype=torch.float32, device=dev)
        rec = np.zeros(len(specs), _hip.DERIVE_JOB)
        out, off, max_quads = {}, 0, 0
        for i, (name, kind, a, b) in enumerate(specs):
            dst = buf[off:off + a.numel()]
            rec[i]["src"], rec[i]["dst"] = a.data_ptr(), dst.data_ptr()
            if kind == "sum":
                if not (a.is_contiguous() and b.is_contiguous()):
                    raise _hip.HipLibraryError("bias vectors must be contiguous")
                rec[i]["src2"], rec[i]["n"], rec[i]["kind"] = b.data_ptr(), a.numel(), 2
                out[name] = dst
            else:
                n, k = a.shape
                if a.stride(1) != 1 or n % 16 or k % 16:
                    raise _hip.HipLibraryError("recurrent weight of shape %s / strides %s" % (tuple(a.shape), a.stride()))
                if kind == "pack":
                    rec[i]["n"], rec[i]["k"], rec[i]["kind"] = n, k, 0
                else:  # fragment order of a.t(): logical [k][n]
                    rec[i]["n"], rec[i]["k"], rec[i]["kind"] = k, n, 1
                rec[i]["ld"] = a.stride(0)
                out[name] = dst
            max_quads = max(max_quads, a.numel() // 4)
            off += a.numel()
        jobs = _hip.to_device(rec, dev)
        _hip.check(_hip.lib().pnmn_derive_params(jobs.data_ptr(), len(specs), max_quads, _hip.stream_ptr(dev)), "derive_params")
        self._key, self._out, self._jobs = key, out, jobs
        self._where, self._n_jobs, self._max_quads, self._device, self._buf = where, len(specs), max_quads, dev, buf
        return out


class _Alias(torch.autograd.Function):
    """``value`` (computed elsewhere from ``a`` and ``b`` as a + b) enters the graph as if it were ``a + b``."""

    @staticmethod
    def forward(ctx, a, b, value):
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        return g, g, None


class _SplitRows(torch.autograd.Function):
    """``(x[:n], x[n:])`` as views whose backward is ONE concatenation (a missing side: zeros).  For an encoder state that
    two decodes share (``Seq2SeqBase.split_rows``): gathering the two row sets with ``index_select`` costs a copy per tensor
    and set forward, and a zero-filled full-size gradient, an ``index_add_`` and an addition each backward."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.rest = n, x.size(0) - n
        ctx.tail, ctx.kind = tuple(x.shape[1:]), (x.dtype, x.device)
        return x.narrow(0, 0, n), x.narrow(0, n, x.size(0) - n)

    @staticmethod
    def backward(ctx, da, db):
        dtype, device = ctx.kind
        if da is None:
            da = torch.zeros((ctx.n,) + ctx.tail, dtype=dtype, device=device)
        if db is None:
            db = torch.zeros((ctx.rest,) + ctx.tail, dtype=dtype, device=device)
        return torch.cat((da, db), 0), None


class _SplitColumns(torch.autograd.Function):
    """``(w[:, :k], w[:, k:])`` whose backward is ONE concatenation: two slices of a parameter otherwise cost two
    zero-filled full-size gradients, two copies into them and an addition per use."""

    @staticmethod
    def forward(ctx, w, k):
        ctx.k, ctx.n = k, w.size(1)
        return w[:, :k], w[:, k:]

    @staticmethod
    def backward(ctx, da, db):
        if da is None and db is None:
            return None, None
        ref = da if da is not None else db
        if da is None:
            da = ref.new_zeros(ref.size(0), ctx.k)
        if db is None:
            db = ref.new_zeros(ref.size(0), ctx.n - ctx.k)
        return torch.cat((da, db), 1), None


class _TokenTable(torch.autograd.Function):
    """``F.linear(embedding.weight, weight, bias)`` ([V, 4H]) where the embedding's padding row receives no
    gradient (nn.Embedding(padding_idx=...) never updates it; its value is zero, so the table row is the bias).
    One launch each way for the vocabularies here (``pnmn_token_table_fwd`` / ``_bwd``); library GEMMs otherwise."""

    @staticmethod
    def _fused(emb, weight):
        return (emb.device.type == "cuda" and emb.size(0) <= 128 and emb.size(1) % 16 == 0 and weight.size(0) % 64 == 0
                and emb.is_contiguous() and weight.stride(1) == 1 and weight.stride(0) % 4 == 0)

    @staticmethod
    def forward(ctx, emb, weight, bias, padding_idx):
        ctx.save_for_backward(emb, weight)
        ctx.padding_idx = padding_idx
        emb_d, weight_d = emb.detach(), weight.detach()
        if not _TokenTable._fused(emb_d, weight_d) or not bias.is_contiguous():
            return torch.addmm(bias, emb_d, weight_d.t())
        V, K = emb_d.shape
        N = weight_d.size(0)
        table = torch.empty(V, N, dtype=emb_d.dtype, device=emb_d.device)
        _hip.check(_hip.lib().pnmn_token_table_fwd(emb_d.data_ptr(), weight_d.data_ptr(), weight_d.stride(0), bias.data_ptr(),
                                                   V, K, N, table.data_ptr(), _hip.stream_ptr(emb_d.device)), "token_table_fwd")
        return table

    @staticmethod
    def backward(ctx, dtable):
        emb, weight = ctx.saved_tensors
        need = ctx.needs_input_grad
        if _TokenTable._fused(emb, weight):
            dtable = dtable.contiguous()
            V, K = emb.shape
            N = weight.size(0)
            f = dict(dtype=emb.dtype, device=emb.device)
            demb = torch.empty(V, K, **f) if need[0] else None
            dweight = torch.empty(N, K, **f) if need[1] else None
            dbias = torch.empty(N, **f) if need[2] else None
            ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            _hip.check(_hip.lib().pnmn_token_table_bwd(dtable.data_ptr(), emb.data_ptr(), weight.data_ptr(), weight.stride(0),
                                                       V, K, N, -1 if ctx.padding_idx is None else ctx.padding_idx,
                                                       ptr(demb), ptr(dweight), 0, ptr(dbias), None, _hip.stream_ptr(emb.device)),
                       "token_table_bwd")
            return demb, dweight, dbias, None
        demb = dweight = dbias = None
        if need[0]:
            demb = dtable @ weight
            if ctx.padding_idx is not None:
                demb[ctx.padding_idx].zero_()
        if need[1]:
            dweight = dtable.t() @ emb
        if need[2]:
            dbias = dtable.sum(0)
        return demb, dweight, dbias, None


class _TokenPrep(object):
    """``pnmn_token_prep`` (see include/probnmn_hip.h): sentence boundaries, mask and last-token index of a
    right-padded token matrix in one launch."""

    @staticmethod
    def run(tokens: torch.Tensor, pad: int, bos: int, eos: int, drop_first: bool, want_mask: bool):
        if tokens.device.type != "cuda":
            raise _hip.HipLibraryError("token input on %s: the HIP path needs a ROCm device" % tokens.device)
        if tokens.dtype != torch.long:
            tokens = tokens.long()
        if tokens.stride(1) != 1:
            tokens = tokens.contiguous()
        B, T = tokens.shape
        W = T + 2 - int(drop_first)
        out = torch.empty(B, W, dtype=torch.long, device=tokens.device)
        fmask = torch.empty(B, W, dtype=torch.float32, device=tokens.device) if want_mask else None
        last = torch.empty(B, dtype=torch.int32, device=tokens.device) if want_mask else None
        _hip.check(_hip.lib().pnmn_token_prep(tokens.data_ptr(), tokens.stride(0), B, T, pad, bos, eos, int(drop_first),
                                              out.data_ptr(), fmask.data_ptr() if want_mask else None,
                                              last.data_ptr() if want_mask else None, _hip.stream_ptr(tokens.device)),
                   "token_prep")
        return out, fmask, last


class _MaskAndLast(torch.autograd.Function):
    """(hs [B,T,H], fmask [B,T], last [B]) -> (hs * fmask[..., None], its row ``last[b]`` per example): the
    zeroed padded steps of ``PytorchSeq2SeqWrapper`` and ``get_final_encoder_states`` in one launch each way."""

    @staticmethod
    def forward(ctx, hs, fmask, last):
        hs = hs.contiguous()
        B, T, H = hs.shape
        enc = torch.empty_like(hs)
        hlast = torch.empty(B, H, dtype=hs.dtype, device=hs.device)
        _hip.check(_hip.lib().pnmn_mask_last_fwd(hs.data_ptr(), fmask.data_ptr(), last.data_ptr(), B, T, H, enc.data_ptr(),
                                                 hlast.data_ptr(), _hip.stream_ptr(hs.device)), "mask_last_fwd")
        ctx.save_for_backward(fmask, last)
        ctx.shape = (B, T, H)
        return enc, hlast

    @staticmethod
    def backward(ctx, denc, dhlast):
        fmask, last = ctx.saved_tensors
        B, T, H = ctx.shape
        denc = denc.contiguous() if denc is not None else None
        dhlast = dhlast.contiguous() if dhlast is not None else None
        dhs = torch.empty(B, T, H, dtype=fmask.dtype, device=fmask.device)
        _hip.check(_hip.lib().pnmn_mask_last_bwd(denc.data_ptr() if denc is not None else None,
                                                 dhlast.data_ptr() if dhlast is not None else None, fmask.data_ptr(),
                                                 last.data_ptr(), B, T, H, dhs.data_ptr(), _hip.stream_ptr(fmask.device)),
                   "mask_last_bwd")
        return dhs, None, None


def embedding_grad(dy: torch.Tensor, tokens: torch.Tensor, vocab: int, shift: bool = False, start: int = 0,
                   skip: int = -1) -> torch.Tensor:
    """dW[v] = sum of the rows of ``dy`` ([B, T, C]) whose token is v (``pnmn_embedding_grad``)."""
    B, T = tokens.shape
    C = dy.size(-1)
    dy = dy.contiguous()
    if tokens.stride(1) != 1:
        tokens = tokens.contiguous()
    dw = torch.empty(vocab, C, dtype=dy.dtype, device=dy.device)
    ws = torch.empty(int(_hip.lib().pnmn_embedding_grad_workspace_bytes(B, T, vocab)), dtype=torch.uint8, device=dy.device)
    _hip.check(_hip.lib().pnmn_embedding_grad(dy.data_ptr(), tokens.data_ptr(), tokens.stride(0), B, T, C, vocab, int(shift),
                                              start, skip, 0, dw.data_ptr(), ws.data_ptr(), _hip.stream_ptr(dy.device)),
               "embedding_grad")
    return dw


class _LinearRows(torch.autograd.Function):
    """``F.linear`` over B x T rows whose weight gradient goes through ``wgrad_gemm`` (autograd's
    plain ``dy.t() @ x`` has 32 output tiles and a 47 000-long reduction at these shapes)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.size(-1))
        dx = (dy2 @ weight).view_as(x) if ctx.needs_input_grad[0] else None
        dw = wgrad_gemm(dy2.contiguous(), x.reshape(-1, x.size(-1)).contiguous()) if ctx.needs_input_grad[1] else None
        db = dy2.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    if x.device.type != "cuda" or not torch.is_grad_enabled():
        return F.linear(x, weight, bias)
    return _LinearRows.apply(x, weight, bias)


class _EmbeddingLookup(torch.autograd.Function):
    """``F.embedding`` whose weight gradient is one kernel: the vocabularies here have < 100 entries, so the
    scatter-add of B x T rows into them that torch's backward does (sort + segmented reduction, ~0.4 ms per
    call) is ``pnmn_embedding_grad`` (rows bucketed by token, summed chunk by chunk; larger vocabularies: a
    one-hot GEMM)."""

    @staticmethod
    def forward(ctx, weight, tokens, padding_idx):
        ctx.save_for_backward(tokens)
        ctx.vocab, ctx.padding_idx = weight.size(0), padding_idx
        return F.embedding(tokens, weight)

    @staticmethod
    def backward(ctx, dy):
        (tokens,) = ctx.saved_tensors
        return _table_grad(dy, tokens, ctx.vocab, ctx.padding_idx), None, None


def embedding_lookup(module: nn.Embedding, tokens: torch.Tensor) -> torch.Tensor:
    if tokens.device.type != "cuda" or not torch.is_grad_enabled() or not module.weight.requires_grad:
        return module(tokens)
    return _EmbeddingLookup.apply(module.weight, tokens, module.padding_idx)


def _lstm_workspace(batch: int, backward: bool, device) -> Optional[torch.Tensor]:
    """Scratch for the multi-CU LSTM kernels (step counters + the backward's exchange buffer); ``None``
    when the library keeps one workgroup per row tile (batch large enough to fill the chip, or
    PNMN_LSTM_CLUSTER=0).  A fresh allocation per launch: the caching allocator orders its reuse on
    the launch stream."""
    if os.environ.get("PNMN_LSTM_CLUSTER", "1") == "0":
        return None
    n = int(_hip.lib().pnmn_lstm_seq_workspace_bytes(batch, 1 if backward else 0))
    return torch.empty(n, dtype=torch.uint8, device=device) if n > 0 else None


def _decoder_workspace(batch: int, backward: bool, device) -> Optional[torch.Tensor]:
    """Scratch of the multi-CU decoder kernels; ``None`` = use the one-workgroup-per-tile kernels
    (PNMN_DECODER_CLUSTER=0, or a device too small for eight resident workgroups per tile)."""
    if os.environ.get("PNMN_DECODER_CLUSTER", "1") == "0":
        return None
    n = int(_hip.lib().pnmn_attn_lstm_multi_workspace_bytes(batch, 1 if backward else 0))
    return torch.empty(n, dtype=torch.uint8, device=device) if n > 0 else None


class _LSTMLayerSeq(torch.autograd.Function):
    """The recurrent half of one LSTM layer over a whole padded sequence, as ONE persistent kernel
    launch (``pnmn_lstm_seq_fwd`` / ``_bwd``): (xp [B,T,4H] = input projection + biases, W_hh) -> all
    hidden states [B,T,H].  The weight gradient of W_hh is one GEMM over the saved states.
    With ``tokens`` ([B,T] int64): ``xp`` is the [V,4H] per-token table of ``_TokenTable`` and the kernel reads
    row ``tokens[b,t]`` of it -- ``F.embedding(tokens, table)`` is never written out; its gradient is the
    per-token sum of ``pnmn_embedding_grad`` over the gate gradients."""

    @staticmethod
    def forward(ctx, xp, w_hh, wp=None, w_t=None, tokens=None):
        if xp.device.type != "cuda":
            raise _hip.HipLibraryError("LSTM layer on %s: the HIP path needs a ROCm device (no CPU fallback)" % xp.device)
        xp, w = xp.contiguous(), w_hh.detach()
        if tokens is not None:
            if tokens.dtype != torch.long or tokens.stride(1) != 1:
                tokens = tokens.long().contiguous()
            (B, T), H4 = tokens.shape, xp.size(1)
        else:
            B, T, H4 = xp.shape
        Hd = H4 // 4
        hs = torch.empty(B, T, Hd, dtype=xp.dtype, device=xp.device)
        cs = torch.empty_like(hs)
        act = torch.empty(B, T, H4, dtype=xp.dtype, device=xp.device)
        if wp is None:
            wp = pack_fragments(w)
        ctx.w_t = w_t  # (fragment order of W_hh^T from the model's DerivedParams, else packed in backward)
        ctx.tokens, ctx.vocab = tokens, xp.size(0)
        ws = _lstm_workspace(B, False, xp.device)
        _hip.check(_hip.lib().pnmn_lstm_seq_fwd(xp.data_ptr(), tokens.data_ptr() if tokens is not None else None,
                                                tokens.stride(0) if tokens is not None else 0, wp.data_ptr(),
                                                hs.data_ptr(), cs.data_ptr(), act.data_ptr(),
                                                B, T, Hd, ws.data_ptr() if ws is not None else None,
                                                _hip.stream_ptr(xp.device)), "lstm_seq_fwd")
        ctx.save_for_backward(hs, cs, act, w)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        hs, cs, act, w = ctx.saved_tensors
        B, T, Hd = hs.shape
        dhs = dhs.contiguous()
        w_t = ctx.w_t if ctx.w_t is not None else pack_fragments(w.t())  # W_hh^T [H][4H], fragment order
        dgates = torch.empty_like(act)
        ws = _lstm_workspace(B, True, hs.device)
        _hip.check(_hip.lib().pnmn_lstm_seq_bwd(dhs.data_ptr(), act.data_ptr(), cs.data_ptr(), w_t.data_ptr(),
                                                dgates.data_ptr(), B, T, Hd, ws.data_ptr() if ws is not None else None,
                                                _hip.stream_ptr(hs.device)), "lstm_seq_bwd")
        dw_hh = None
        if ctx.needs_input_grad[1]:
            hprev = torch.cat((hs.new_zeros(B, 1, Hd), hs[:, :-1]), 1).reshape(B * T, Hd)  # h_{t-1} per (row, step)
            dw_hh = wgrad_gemm(dgates.reshape(B * T, 4 * Hd), hprev)
        dxp = dgates
        if ctx.tokens is not None:
            dxp = _table_grad(dgates, ctx.tokens, ctx.vocab) if ctx.needs_input_grad[0] else None
        return dxp, dw_hh, None, None, None


def _table_grad(dy: torch.Tensor, tokens: torch.Tensor, vocab: int, padding_idx: Optional[int] = None) -> torch.Tensor:
    """Gradient of ``F.embedding(tokens, table)`` wrt the table, from the gradient ``dy`` [B,T,C] of its output."""
    if vocab <= 128 and dy.size(-1) % 64 == 0 and tokens.dim() == 2:
        return embedding_grad(dy, tokens, vocab, skip=padding_idx if padding_idx is not None else -1)
    flat = tokens.reshape(-1)
    onehot = torch.zeros(flat.numel(), vocab, dtype=dy.dtype, device=dy.device)
    onehot.scatter_(1, flat.unsqueeze(1), 1.0)
    dw = wgrad_gemm(onehot, dy.reshape(flat.numel(), -1).contiguous())
    if padding_idx is not None:
        dw[padding_idx].zero_()
    return dw


class _AttnLSTMDecoder(torch.autograd.Function):
    """The decoding loop as one persistent kernel launch (``pnmn_attn_lstm_fwd`` / ``_bwd``).

    inputs : xe [B,T,4H] (teacher forcing) or etable [V,4H] (free running), enc [B,S,H], mask [B,S] float,
             h0 [B,H], W_c [4H,H], W_hh [4H,H], W_p [V,H], b_p [V]
    outputs: hidden states [B,T,H], tokens [B,T] (free running only)
    Weight gradients are batched GEMMs over what the kernels saved."""

    @staticmethod
    def forward(ctx, xe, etable, enc, mask, h0, w_c, w_hh, w_p, b_p, mode, T, seed, row_offset, pad, unk, start,
                packs=None, in_tokens=None):
        dev = enc.device
        if dev.type != "cuda":
            raise _hip.HipLibraryError("decoder on %s: the HIP path needs a ROCm device (no CPU fallback)" % dev)
        enc, mask, h0 = enc.contiguous(), mask.contiguous(), h0.contiguous()
        w_c, w_hh = w_c.detach(), w_hh.detach()
        if packs is not None:  # (w_c, w_hh, w_c^T, w_hh^T in fragment order, from the model's DerivedParams)
            w_c_p, w_hh_p = packs[0], packs[1]
            ctx.packs_t = (packs[2], packs[3])
        else:
            w_c_p, w_hh_p = pack_fragments(w_c), pack_fragments(w_hh)
            ctx.packs_t = None
        B, S, Hd = enc.shape
        f = dict(dtype=torch.float32, device=dev)
        hs, cs, cx = torch.empty(B, T, Hd, **f), torch.empty(B, T, Hd, **f), torch.empty(B, T, Hd, **f)
        act = torch.empty(B, T, 4 * Hd, **f)
        probs = torch.empty(B, T, S, **f)
        tokens = None
        V = 0
        if mode != 0:
            etable, w_p, b_p = etable.contiguous(), w_p.detach().contiguous(), b_p.detach().contiguous()
            tokens = torch.empty(B, T, dtype=torch.long, device=dev)
            V = w_p.size(0)
        elif in_tokens is not None:  # teacher forcing from the [V,4H] table: step t's input is row in_tokens[b,t]
            etable = etable.contiguous()
            if in_tokens.dtype != torch.long or in_tokens.stride(1) != 1:
                in_tokens = in_tokens.long().contiguous()
            xe = None
        else:
            xe = xe.contiguous()
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        args = (ptr(xe if mode == 0 else None), ptr(etable if (mode != 0 or in_tokens is not None) else None),
                enc.data_ptr(), mask.data_ptr(),
                h0.data_ptr(), w_c_p.data_ptr(), w_hh_p.data_ptr(), ptr(w_p if mode != 0 else None),
                ptr(b_p if mode != 0 else None), hs.data_ptr(), cs.data_ptr(), act.data_ptr(), cx.data_ptr(),
                probs.data_ptr(), ptr(tokens), B, T, S, V, Hd, mode, pad, unk, start, seed, row_offset,
                ptr(in_tokens) if mode == 0 else None, in_tokens.stride(0) if (mode == 0 and in_tokens is not None) else 0)
        ws = _decoder_workspace(B, False, dev)
        if ws is not None:
            _hip.check(_hip.lib().pnmn_attn_lstm_fwd_multi(*args, ws.data_ptr(), _hip.stream_ptr(dev)), "attn_lstm_fwd_multi")
        else:
            _hip.check(_hip.lib().pnmn_attn_lstm_fwd(*args, _hip.stream_ptr(dev)), "attn_lstm_fwd")
        ctx.save_for_backward(hs, cs, act, cx, probs, enc, mask, h0, w_c, w_hh,
                              tokens if tokens is not None else torch.empty(0, device=dev))
        ctx.mode, ctx.start, ctx.vocab = mode, start, (etable.size(0) if etable is not None else 0)
        ctx.in_tokens = in_tokens if mode == 0 else None
        if tokens is not None:
            ctx.mark_non_differentiable(tokens)
            return hs, tokens
        return hs, torch.empty(0, dtype=torch.long, device=dev)

    @staticmethod
    def backward(ctx, dhs, _):
        hs, cs, act, cx, probs, enc, mask, h0, w_c, w_hh, tokens = ctx.saved_tensors
        B, T, Hd = hs.shape
        S = enc.size(1)
        dev = hs.device
        dgates = torch.empty_like(act)
        dh0 = torch.empty_like(h0)
        # (named temporaries: a tensor that dies right after .data_ptr() may be recycled by the next allocation)
        dhs_c = dhs.contiguous()
        w_c_t, w_hh_t = ctx.packs_t if ctx.packs_t is not None else (pack_fragments(w_c.t()), pack_fragments(w_hh.t()))
        hprev = torch.cat((h0.unsqueeze(1), hs[:, :-1]), 1)  # h_{t-1} of every (row, step)
        ws = _decoder_workspace(B, True, dev)
        if ws is not None:
            dctx, dscore, weights = torch.empty_like(hs), torch.empty_like(probs), torch.empty_like(probs)
            _hip.check(_hip.lib().pnmn_attn_lstm_bwd_multi(
                dhs_c.data_ptr(), act.data_ptr(), cs.data_ptr(), hs.data_ptr(), probs.data_ptr(), enc.data_ptr(),
                mask.data_ptr(), h0.data_ptr(), w_c_t.data_ptr(), w_hh_t.data_ptr(), dgates.data_ptr(), dctx.data_ptr(),
                dscore.data_ptr(), weights.data_ptr(), dh0.data_ptr(), B, T, S, Hd, ws.data_ptr(), _hip.stream_ptr(dev)),
                "attn_lstm_bwd_multi")
            # encoder-output gradient as two GEMMs per row over the T steps: enc_s enters step t through
            # the context (weight w_ts = masked, renormalised attention, written out by the kernel) and through the
            # score (gradient dscore_ts, times h_{t-1})
            if T <= 64:
                denc = torch.empty_like(enc)
                _hip.check(_hip.lib().pnmn_attn_denc(weights.data_ptr(), dscore.data_ptr(), dctx.data_ptr(), hs.data_ptr(),
                                                     h0.data_ptr(), denc.data_ptr(), B, T, S, Hd, _hip.stream_ptr(dev)),
                           "attn_denc")
            else:
                denc = torch.baddbmm(torch.bmm(weights.transpose(1, 2), dctx), dscore.transpose(1, 2), hprev)
        else:
            denc = torch.zeros_like(enc)
            _hip.check(_hip.lib().pnmn_attn_lstm_bwd(
                dhs_c.data_ptr(), act.data_ptr(), cs.data_ptr(), hs.data_ptr(), cx.data_ptr(), probs.data_ptr(),
                enc.data_ptr(), mask.data_ptr(), h0.data_ptr(), w_c_t.data_ptr(), w_hh_t.data_ptr(), dgates.data_ptr(),
                denc.data_ptr(), dh0.data_ptr(), B, T, S, Hd, _hip.stream_ptr(dev)), "attn_lstm_bwd")
        flat = dgates.reshape(B * T, 4 * Hd)
        dw_c = wgrad_gemm(flat, cx.reshape(B * T, Hd))
        dw_hh = wgrad_gemm(flat, hprev.reshape(B * T, Hd))
        dxe = detable = None
        if ctx.mode == 0 and ctx.in_tokens is not None:
            detable = _table_grad(dgates, ctx.in_tokens, ctx.vocab) if ctx.needs_input_grad[1] else None
        elif ctx.mode == 0:
            dxe = dgates
        else:
            if ctx.vocab <= 128:  # step t's input is the token chosen at step t - 1 (@start@ first)
                detable = embedding_grad(dgates, tokens, ctx.vocab, shift=True, start=ctx.start)
            else:
                tok_in = torch.cat((tokens.new_full((B, 1), ctx.start), tokens[:, :-1]), 1).reshape(-1)
                detable = torch.zeros(ctx.vocab, 4 * Hd, dtype=dgates.dtype, device=dev).index_add_(0, tok_in, flat)
        return (dxe, detable, denc, None, dh0, dw_c, dw_hh) + (None,) * 11


class _AttnLSTMDecoderPair(torch.autograd.Function):
    """TWO independent decoder passes in one launch each way (``pnmn_attn_lstm_fwd_multi_pair`` / ``_bwd_multi_pair``):
    the persistent decoder kernels are bound by their per-step hand-off latency, so side by side they take as long as
    the longer pass.  Either side may be teacher forced (mode 0: step inputs ``in_tokens`` [B,T] index the per-token
    table) or free running (mode 1 sampling / 2 greedy: the kernel picks each step's token; ``etable`` then is the
    projected embedding table).  Per side the tensor inputs are etable [V,4H], enc [B,S,H], mask [B,S], h0 [B,H], W_c,
    W_hh; ``meta`` carries the rest.  Same arithmetic as two ``_AttnLSTMDecoder`` calls (bit for bit: the kernels'
    bodies are shared; the library runs the passes one after the other when they do not fit the chip together)."""

    @staticmethod
    def forward(ctx, etable_a, enc_a, mask_a, h0_a, w_c_a, w_hh_a, etable_b, enc_b, mask_b, h0_b, w_c_b, w_hh_b, meta):
        dev = enc_a.device
        sides, jobs = [], np.zeros(2, _hip.DECODER_FWD_JOB)
        for k, (etable, enc, mask, h0, w_c, w_hh, m) in enumerate(((etable_a, enc_a, mask_a, h0_a, w_c_a, w_hh_a, meta[0]),
                                                                  (etable_b, enc_b, mask_b, h0_b, w_c_b, w_hh_b, meta[1]))):
            etable, enc, mask, h0 = etable.contiguous(), enc.contiguous(), mask.contiguous(), h0.contiguous()
            w_c, w_hh = w_c.detach(), w_hh.detach()
            packs = m["packs"] if m["packs"] is not None else (pack_fragments(w_c), pack_fragments(w_hh), None, None)
            mode, T = m["mode"], m["T"]
            B, S, Hd = enc.shape
            f = dict(dtype=torch.float32, device=dev)
            hs, cs, cx = torch.empty(B, T, Hd, **f), torch.empty(B, T, Hd, **f), torch.empty(B, T, Hd, **f)
            act, probs = torch.empty(B, T, 4 * Hd, **f), torch.empty(B, T, S, **f)
            j = jobs[k]
            j["etable"], j["enc"], j["mask"], j["h0"] = etable.data_ptr(), enc.data_ptr(), mask.data_ptr(), h0.data_ptr()
            j["w_c"], j["w_hh"] = packs[0].data_ptr(), packs[1].data_ptr()
            j["hs"], j["cs"], j["act"], j["ctx"], j["probs"] = hs.data_ptr(), cs.data_ptr(), act.data_ptr(), cx.data_ptr(), probs.data_ptr()
            j["B"], j["T"], j["S"], j["start_index"] = B, T, S, m["start"]
            in_tokens = tokens = None
            keep = [etable, packs]
            if mode == 0:
                in_tokens = m["in_tokens"]
                if in_tokens.dtype != torch.long or in_tokens.stride(1) != 1:
                    in_tokens = in_tokens.long().contiguous()
                j["in_tokens"], j["in_token_stride"] = in_tokens.data_ptr(), in_tokens.stride(0)
            else:
                w_p, b_p = m["w_p"].detach().contiguous(), m["b_p"].detach().contiguous()
                tokens = torch.empty(B, T, dtype=torch.long, device=dev)
                j["w_p"], j["b_p"], j["tokens"], j["V"], j["sample"] = w_p.data_ptr(), b_p.data_ptr(), tokens.data_ptr(), w_p.size(0), mode
                j["pad_index"], j["unk_index"], j["seed"], j["row_offset"] = m["pad"], m["unk"], m["seed"], m["row_offset"]
                keep += [w_p, b_p]
            sides.append(dict(hs=hs, cs=cs, act=act, cx=cx, probs=probs, enc=enc, mask=mask, h0=h0, w_c=w_c, w_hh=w_hh,
                              in_tokens=in_tokens, tokens=tokens, vocab=etable.size(0), mode=mode, start=m["start"],
                              packs_t=(packs[2], packs[3]) if packs[2] is not None else None, keep=keep))
        Ba, Bb = sides[0]["hs"].size(0), sides[1]["hs"].size(0)
        ws = torch.empty(int(_hip.lib().pnmn_attn_lstm_pair_workspace_bytes(Ba, Bb, 0)), dtype=torch.uint8, device=dev)
        _hip.check(_hip.lib().pnmn_attn_lstm_fwd_multi_pair(jobs[0:1].ctypes.data, jobs[1:2].ctypes.data, sides[0]["hs"].size(2),
                                                            ws.data_ptr(), _hip.stream_ptr(dev)), "attn_lstm_fwd_multi_pair")
        saved, outs = [], []
        empty = torch.empty(0, dtype=torch.long, device=dev)
        for sd in sides:
            step_tokens = sd["in_tokens"] if sd["mode"] == 0 else sd["tokens"]
            saved += [sd["hs"], sd["cs"], sd["act"], sd["cx"], sd["probs"], sd["enc"], sd["mask"], sd["h0"], sd["w_c"], sd["w_hh"],
                      step_tokens]
            toks = sd["tokens"] if sd["tokens"] is not None else empty
            outs += [sd["hs"], toks]
        ctx.save_for_backward(*saved)
        ctx.side_meta = [(sd["vocab"], sd["packs_t"], sd["mode"], sd["start"]) for sd in sides]
        ctx.mark_non_differentiable(outs[1], outs[3])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dhs_a, _ta, dhs_b, _tb):
        return (*_decoder_sides_backward(ctx.saved_tensors, ctx.side_meta, (dhs_a, dhs_b), ctx.needs_input_grad), None)


def _decoder_sides_backward(saved, side_meta, dhs_list, needs_input_grad):
    """Backward of 1-3 decoder passes whose forward saved (hs, cs, act, cx, probs, enc, mask, h0, w_c, w_hh, step tokens) each:
    ONE launch for all of them (``pnmn_attn_lstm_bwd_multi`` / ``_pair`` / ``_group3``), then per pass the encoder-output
    gradient, the two weight-gradient GEMMs and the table gradient.  Returns six gradients per pass, in the order of the
    forward's tensor inputs (etable, enc, mask, h0, w_c, w_hh)."""
    dev = saved[0].device
    n = len(dhs_list)
    jobs = np.zeros(n, _hip.DECODER_BWD_JOB)
    sides = []
    for k, dhs in enumerate(dhs_list):
        hs, cs, act, cx, probs, enc, mask, h0, w_c, w_hh, step_tokens = saved[11 * k: 11 * k + 11]
        vocab, packs_t, mode, start = side_meta[k]
        B, T, Hd = hs.shape
        S = enc.size(1)
        dhs_c = torch.zeros_like(hs) if dhs is None else dhs.contiguous()
        w_c_t, w_hh_t = packs_t if packs_t is not None else (pack_fragments(w_c.t()), pack_fragments(w_hh.t()))
        dgates, dh0 = torch.empty_like(act), torch.empty_like(h0)
        dctx, dscore, weights = torch.empty_like(hs), torch.empty_like(probs), torch.empty_like(probs)
        j = jobs[k]
        for name, t in (("dhs", dhs_c), ("act", act), ("cs", cs), ("hs", hs), ("probs", probs), ("enc", enc), ("mask", mask),
                        ("h0", h0), ("w_c_t", w_c_t), ("w_hh_t", w_hh_t), ("dgates", dgates), ("dctx", dctx),
                        ("dscore", dscore), ("weights", weights), ("dh0", dh0)):
            j[name] = t.data_ptr()
        j["B"], j["T"], j["S"] = B, T, S
        sides.append(dict(hs=hs, cx=cx, enc=enc, h0=h0, step_tokens=step_tokens, vocab=vocab, mode=mode, start=start,
                          dgates=dgates, dh0=dh0, dctx=dctx, dscore=dscore, weights=weights, keep=(dhs_c, w_c_t, w_hh_t),
                          B=B, T=T, S=S, Hd=Hd))
    lib, Hd = _hip.lib(), sides[0]["Hd"]
    rows = [sd["B"] for sd in sides]
    if n == 1:
        ws = torch.empty(int(lib.pnmn_attn_lstm_multi_workspace_bytes(rows[0], 1)), dtype=torch.uint8, device=dev)
        j = jobs[0]
        _hip.check(lib.pnmn_attn_lstm_bwd_multi(*(int(j[f]) for f in ("dhs", "act", "cs", "hs", "probs", "enc", "mask", "h0", "w_c_t",
                                                                      "w_hh_t", "dgates", "dctx", "dscore", "weights", "dh0")),
                                                rows[0], sides[0]["T"], sides[0]["S"], Hd, ws.data_ptr(), _hip.stream_ptr(dev)),
                   "attn_lstm_bwd_multi")
    elif n == 2:
        ws = torch.empty(int(lib.pnmn_attn_lstm_pair_workspace_bytes(rows[0], rows[1], 1)), dtype=torch.uint8, device=dev)
        _hip.check(lib.pnmn_attn_lstm_bwd_multi_pair(jobs[0:1].ctypes.data, jobs[1:2].ctypes.data, Hd, ws.data_ptr(),
                                                     _hip.stream_ptr(dev)), "attn_lstm_bwd_multi_pair")
    else:
        ws = torch.empty(int(lib.pnmn_attn_lstm_group3_workspace_bytes(rows[0], rows[1], rows[2], 1)), dtype=torch.uint8, device=dev)
        _hip.check(lib.pnmn_attn_lstm_bwd_multi_group3(jobs[0:1].ctypes.data, jobs[1:2].ctypes.data, jobs[2:3].ctypes.data, Hd,
                                                       ws.data_ptr(), _hip.stream_ptr(dev)), "attn_lstm_bwd_multi_group3")
    grads = []
    for k, sd in enumerate(sides):
        B, T, S, Hd = sd["B"], sd["T"], sd["S"], sd["Hd"]
        need = needs_input_grad[6 * k: 6 * k + 6]
        denc = None
        if need[1]:
            if T <= 64:
                denc = torch.empty_like(sd["enc"])
                _hip.check(_hip.lib().pnmn_attn_denc(sd["weights"].data_ptr(), sd["dscore"].data_ptr(), sd["dctx"].data_ptr(),
                                                     sd["hs"].data_ptr(), sd["h0"].data_ptr(), denc.data_ptr(), B, T, S, Hd,
                                                     _hip.stream_ptr(dev)), "attn_denc")
            else:
                hprev = torch.cat((sd["h0"].unsqueeze(1), sd["hs"][:, :-1]), 1)
                denc = torch.baddbmm(torch.bmm(sd["weights"].transpose(1, 2), sd["dctx"]), sd["dscore"].transpose(1, 2), hprev)
        flat = sd["dgates"].reshape(B * T, 4 * Hd)
        dw_c = wgrad_gemm(flat, sd["cx"].reshape(B * T, Hd)) if need[4] else None
        dw_hh = None
        if need[5]:
            hprev = torch.cat((sd["h0"].unsqueeze(1), sd["hs"][:, :-1]), 1)
            dw_hh = wgrad_gemm(flat, hprev.reshape(B * T, Hd))
        detable = None
        if need[0]:
            if sd["mode"] == 0:
                detable = _table_grad(sd["dgates"], sd["step_tokens"], sd["vocab"])
            elif sd["vocab"] <= 128:  # step t's input is the token chosen at step t - 1 (@start@ first)
                detable = embedding_grad(sd["dgates"], sd["step_tokens"], sd["vocab"], shift=True, start=sd["start"])
            else:
                tok_in = torch.cat((sd["step_tokens"].new_full((B, 1), sd["start"]), sd["step_tokens"][:, :-1]), 1).reshape(-1)
                detable = torch.zeros(sd["vocab"], 4 * Hd, dtype=flat.dtype, device=dev).index_add_(0, tok_in, flat)
        grads += [detable, denc, None, sd["dh0"], dw_c, dw_hh]
    return grads


class _Capture:
    """Stands in for an autograd context where a Function's ``forward`` is run for its launches only (the graph node that
    owns what it saved is created later: ``_AttnLSTMDecoderGroup``)."""

    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        self.saved = tensors

    def mark_non_differentiable(self, *tensors):
        pass


class _AttnLSTMDecoderGroup(torch.autograd.Function):
    """ONE graph node for up to three decoder passes whose forward launches happened at different times, so that their
    BACKWARD is one launch (``pnmn_attn_lstm_bwd_multi_group3``).  In a training iteration the generator's two decodes
    run first (their samples are the reconstructor's input), the reconstructor's decode later -- but backward the three are
    independent, and on one stream they add their step counts on the iteration's critical chain (at 128 questions per GPU
    the seq2seq backward IS that chain: 275 + 470 us of decoder kernels become 470).  Inputs: per pass etable, enc, mask,
    h0, W_c, W_hh (as ``_AttnLSTMDecoderPair``), then ``meta`` (one dict per pass; ``meta[k]["pre"]`` = index of the pass in
    ``pre``, the ``_Capture`` of an earlier ``decode_pair_launch``, or None: launched here) and ``pre``."""

    @staticmethod
    def forward(ctx, *args):
        meta, pre = args[-2], args[-1]
        tens = args[:-2]
        saved, side_meta, outs = [], [], []
        for k, m in enumerate(meta):
            etable, enc, mask, h0, w_c, w_hh = tens[6 * k: 6 * k + 6]
            if m.get("pre") is not None:
                j = m["pre"]
                saved += list(pre.saved[11 * j: 11 * j + 11])
                side_meta.append(pre.side_meta[j])
                outs += [pre.outs[2 * j], pre.outs[2 * j + 1]]
                continue
            cap = _Capture()
            if m["mode"] == 0:
                hs, tok = _AttnLSTMDecoder.forward(cap, None, etable, enc, mask, h0, w_c, w_hh, None, None, 0, m["T"], 0, 0, 0, 0,
                                                   m["start"], m["packs"], m["in_tokens"])
                step_tokens = cap.in_tokens
            else:
                hs, tok = _AttnLSTMDecoder.forward(cap, None, etable, enc, mask, h0, w_c, w_hh, m["w_p"], m["b_p"], m["mode"], m["T"],
                                                   m["seed"], m["row_offset"], m["pad"], m["unk"], m["start"], m["packs"])
                step_tokens = cap.saved[10]
            saved += list(cap.saved[:10]) + [step_tokens]
            side_meta.append((cap.vocab, cap.packs_t, cap.mode, cap.start))
            outs += [hs, tok]
        ctx.save_for_backward(*saved)
        ctx.side_meta = side_meta
        ctx.mark_non_differentiable(*outs[1::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        return (*_decoder_sides_backward(ctx.saved_tensors, ctx.side_meta, grads[0::2], ctx.needs_input_grad), None, None)


def choose_tokens(logits: torch.Tensor, greedy: bool, seed: int, row_offset: int, step: int,
                  pad: int, unk: int, start: int):
    """One decoding step's token choice on the device; returns (tokens int64 [B], logprob [B] no grad)."""
    logits = logits.detach().contiguous()
    B, V = logits.shape
    tokens = torch.empty(B, dtype=torch.long, device=logits.device)
    lp = torch.empty(B, dtype=torch.float32, device=logits.device)
    _hip.check(_hip.lib().pnmn_sample_tokens(logits.data_ptr(), tokens.data_ptr(), lp.data_ptr(), B, V, int(greedy),
                                             seed, row_offset, step, pad, unk, start,
                                             _hip.stream_ptr(logits.device)), "sample_tokens")
    return tokens, lp


def add_sentence_boundary_token_ids(tokens: torch.Tensor, pad: int, bos: int, eos: int) -> torch.Tensor:
    """(B,T) right-padded -> (B,T+2): @start@ first, @end@ right after the last real token
    (allennlp.nn.util.add_sentence_boundary_token_ids, vectorised)."""
    B, T = tokens.shape
    lengths = (tokens != pad).sum(1)
    out = tokens.new_zeros(B, T + 2)
    out[:, 1:-1] = tokens
    out[:, 0] = bos
    out.scatter_(1, (lengths + 1).unsqueeze(1), eos)
    return out


def masked_softmax(vector: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """allennlp.nn.util.masked_softmax (memory_efficient=False), including its 1e-13."""
    result = F.softmax(vector * mask, dim=-1) * mask
    return result / (result.sum(dim=-1, keepdim=True) + 1e-13)


def sequence_cross_entropy(logits, targets, weights, eps: float = 1e-13):
    """Per-sequence masked-mean cross entropy (allennlp sequence_cross_entropy_with_logits,
    average=None) with explicit weights -- plain torch ops (the kernels take a token mask, see
    ``sequence_nll``)."""
    weights = weights.float()
    nll = -F.log_softmax(logits, dim=-1).gather(2, targets.unsqueeze(-1)).squeeze(-1) * weights
    return nll.sum(1) / (weights.sum(1) + eps)


class _SeqNLL(torch.autograd.Function):
    """``pnmn_seq_nll_{fwd,bwd}``: one launch instead of log_softmax / gather / mask / sum / divide."""

    @staticmethod
    def forward(ctx, logits, tokens, mask_tokens, pad, eps):
        B, T, V = logits.shape
        if logits.stride(2) != 1 or logits.stride(1) != V:
            logits = logits.contiguous()
        if tokens.stride(1) != 1:
            tokens = tokens.contiguous()
        if mask_tokens.stride(1) != 1:
            mask_tokens = mask_tokens.contiguous()
        loss = torch.empty(B, dtype=torch.float32, device=logits.device)
        lse = torch.empty(B, T, dtype=torch.float32, device=logits.device)
        _hip.check(_hip.lib().pnmn_seq_nll_fwd(logits.data_ptr(), logits.stride(0), tokens.data_ptr(), tokens.stride(0),
                                               mask_tokens.data_ptr(), mask_tokens.stride(0), pad, loss.data_ptr(),
                                               lse.data_ptr(), B, T, V, eps, _hip.stream_ptr(logits.device)), "seq_nll_fwd")
        ctx.save_for_backward(logits, tokens, mask_tokens, lse)
        ctx.pad, ctx.eps = pad, eps
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, tokens, mask_tokens, lse = ctx.saved_tensors
        B, T, V = logits.shape
        dlogits = torch.empty(B, T, V, dtype=torch.float32, device=logits.device)
        dloss = dloss.contiguous()
        _hip.check(_hip.lib().pnmn_seq_nll_bwd(logits.data_ptr(), logits.stride(0), tokens.data_ptr(), tokens.stride(0),
                                               mask_tokens.data_ptr(), mask_tokens.stride(0), ctx.pad, lse.data_ptr(),
                                               dloss.data_ptr(), dlogits.data_ptr(), T * V, B, T, V, ctx.eps,
                                               _hip.stream_ptr(logits.device)), "seq_nll_bwd")
        return dlogits, None, None, None, None


def sequence_nll(logits: torch.Tensor, tokens: torch.Tensor, mask_tokens: torch.Tensor, pad: int, eps: float) -> torch.Tensor:
    """loss[b] = sum_t w_t (-log_softmax(logits[b,t])[tokens[b,t]]) / (sum_t w_t + eps), w = mask_tokens != pad.
    Covers both sequence losses of the reference: teacher-forced cross entropy (tokens = mask_tokens =
    targets, eps 1e-13: allennlp's sequence_cross_entropy_with_logits) and the negative mean log-probability
    of a sampled sequence (tokens = the raw samples, mask_tokens = the trimmed predictions, eps 1e-12:
    seq2seq_base.py:235-244)."""
    if logits.device.type != "cuda":
        raise _hip.HipLibraryError("sequence loss input on %s: the HIP path needs a ROCm device" % logits.device)
    return _SeqNLL.apply(logits, tokens, mask_tokens, pad, eps)


def lstm_derived_specs(lstm: nn.LSTM, prefix: str = "l"):
    """DerivedParams specs of an ``nn.LSTM``: per layer the fragment-order W_hh, its transpose and b_ih + b_hh."""
    specs = []
    for layer in range(lstm.num_layers):
        w_hh = getattr(lstm, "weight_hh_l%d" % layer)
        specs.append(("%s%d.hh" % (prefix, layer), "pack", w_hh, None))
        specs.append(("%s%d.hhT" % (prefix, layer), "packT", w_hh, None))
        specs.append(("%s%d.b" % (prefix, layer), "sum", getattr(lstm, "bias_ih_l%d" % layer), getattr(lstm, "bias_hh_l%d" % layer)))
        if layer > 0 and lstm.input_size == lstm.hidden_size == 256:
            # the wavefront launches (pnmn_lstm_stack_*) keep a slice of the upper layers' input weights in registers too
            w_ih = getattr(lstm, "weight_ih_l%d" % layer)
            specs.append(("%s%d.ih" % (prefix, layer), "pack", w_ih, None))
            specs.append(("%s%d.ihT" % (prefix, layer), "packT", w_ih, None))
    return specs


def lstm_derived_params(lstm: nn.LSTM):
    """The parameters ``lstm_derived_specs`` reads (cache key of ``DerivedParams.get``)."""
    return [getattr(lstm, "%s_l%d" % (n, layer)) for layer in range(lstm.num_layers)
            for n in ("weight_hh", "bias_ih", "bias_hh") + (("weight_ih",) if layer > 0 else ())]


_SLOW_PATHS_NOTED = set()


def _note_slow_path(what: str, why: str) -> None:
    """The step-by-step paths (a GEMM and a cell kernel per time step) serve shapes no reference config has; they are
    correct and ~10x slower than the persistent kernels, so they say so once instead of engaging silently."""
    if what not in _SLOW_PATHS_NOTED:
        _SLOW_PATHS_NOTED.add(what)
        import warnings

        warnings.warn("probnmn seq2seq %s runs step by step (one GEMM + one cell launch per time step): %s" % (what, why),
                      RuntimeWarning, stacklevel=3)


def lstm_bias(lstm: nn.LSTM, layer: int, derived: Optional[Dict[str, torch.Tensor]], prefix: str = "l") -> torch.Tensor:
    b_ih, b_hh = getattr(lstm, "bias_ih_l%d" % layer), getattr(lstm, "bias_hh_l%d" % layer)
    if derived is None:
        return b_ih + b_hh
    return _Alias.apply(b_ih, b_hh, derived["%s%d.b" % (prefix, layer)])


def masked_lstm(lstm: nn.LSTM, x: torch.Tensor, mask: torch.Tensor, first_projection: Optional[torch.Tensor] = None,
                derived: Optional[Dict[str, torch.Tensor]] = None, last: Optional[torch.Tensor] = None,
                first_tokens: Optional[torch.Tensor] = None):
    """``PytorchSeq2SeqWrapper(nn.LSTM)(x, mask)``: zero initial state, outputs zero past each row's
    length.  Rows are run over all T steps (a unidirectional state never sees later steps) with the
    input GEMM batched over time and the recurrence in one persistent HIP kernel per layer.
    ``first_projection``: the first layer's input projection when the caller already has it
    ; ``x`` is then unused -- with ``first_tokens`` it is the [V,4H] per-token TABLE and the
    layer kernel looks the rows up itself.  ``derived``: the model's ``DerivedParams`` output (packed
    weights, bias sums).  ``last`` ([B] int32, ``mask`` then being the float mask): also return each row's
    state at that step -- (outputs, last states) from one launch."""
    B, T = mask.shape
    inp = x
    for layer in range(lstm.num_layers):
        w_ih = getattr(lstm, "weight_ih_l%d" % layer)
        w_hh = getattr(lstm, "weight_hh_l%d" % layer)
        tokens = None
        if layer == 0 and first_projection is not None:
            xp = first_projection
            if first_tokens is not None:
                if lstm.hidden_size == 256:
                    tokens = first_tokens
                else:
                    xp = _EmbeddingLookup.apply(xp, first_tokens, None)
        else:
            xp = linear_rows(inp, w_ih, lstm_bias(lstm, layer, derived))  # (B,T,4H): one GEMM for all time steps
        if lstm.hidden_size == 256:
            # one persistent launch for all T steps
            if derived is not None:
                inp = _LSTMLayerSeq.apply(xp, w_hh, derived["l%d.hh" % layer], derived["l%d.hhT" % layer], tokens)
            else:
                inp = _LSTMLayerSeq.apply(xp, w_hh, None, None, tokens)
        else:  # other widths: step by step (GEMM per step + the cell kernel)
            _note_slow_path("LSTM layer", "hidden size %d (the persistent layer kernel is built for 256)" % lstm.hidden_size)
            h = xp.new_zeros(B, lstm.hidden_size)
            c = xp.new_zeros(B, lstm.hidden_size)
            w_hh_t = w_hh.t()
            outs = []
            for t in range(T):
                gates = torch.addmm(xp[:, t], h, w_hh_t)
                h, c = lstm_cell_pointwise(gates, c)
                outs.append(h)
            inp = torch.stack(outs, 1)
    if last is not None:
        return _MaskAndLast.apply(inp, mask, last)
    return inp * mask.unsqueeze(-1).to(inp.dtype)


class _TokenEmbedder(nn.Module):
    """``BasicTextFieldEmbedder({"tokens": Embedding})`` as far as parameter naming goes."""

    def __init__(self, key: str, num_embeddings: int, dim: int, padding_index: int):
        super().__init__()
        emb = nn.Embedding(num_embeddings, dim, padding_idx=padding_index)
        nn.init.xavier_uniform_(emb.weight)  # AllenNLP Embedding init, then zero padding row
        with torch.no_grad():
            emb.weight[padding_index].fill_(0)
        setattr(self, "token_embedder_" + key, emb)
        self._key = "token_embedder_" + key

    @property
    def embedding(self) -> nn.Embedding:
        return getattr(self, self._key)

    def forward(self, tokens):
        return embedding_lookup(self.embedding, tokens)


class _Encoder(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, dropout):
        super().__init__()
        self._module = nn.LSTM(input_size, hidden_size, num_layers, dropout=dropout, batch_first=True)

    def forward(self, x, mask):
        return masked_lstm(self._module, x, mask)

    def forward_tokens(self, embedding: nn.Embedding, tokens: torch.Tensor, mask: torch.Tensor,
                       derived: Optional[Dict[str, torch.Tensor]] = None, last: Optional[torch.Tensor] = None):
        """``forward(embedding(tokens), mask)`` with the first layer's input projection taken from a
        per-token table (``_TokenTable``: V < 100 projected rows instead of a GEMM over all B x T); with ``last`` also each row's state at that step."""
        lstm = self._module
        table = _TokenTable.apply(embedding.weight, lstm.weight_ih_l0, lstm_bias(lstm, 0, derived), embedding.padding_idx)
        return masked_lstm(lstm, None, mask, first_projection=table, derived=derived, last=last, first_tokens=tokens)


class Seq2SeqBase(nn.Module):
    def __init__(
        self,
        vocabulary,
        source_namespace: str,
        target_namespace: str,
        input_size: int = 256,
        hidden_size: int = 256,
        num_layers: int = 2,
        dropout: float = 0.0,
        max_decoding_steps: int = 30,
    ):
        super().__init__()
        self.vocabulary = vocabulary
        # @@PADDING@@, @@UNKNOWN@@, @start@, @end@ have the same indices in all namespaces
        self._pad_index = vocabulary.get_token_index("@@PADDING@@", namespace=source_namespace)
        self._unk_index = vocabulary.get_token_index("@@UNKNOWN@@", namespace=source_namespace)
        self._end_index = vocabulary.get_token_index("@end@", namespace=target_namespace)
        self._start_index = vocabulary.get_token_index("@start@", namespace=target_namespace)
        self._max_decoding_steps = max_decoding_steps
        self._scheduled_sampling_ratio = 0.0
        if dropout != 0.0:
            raise NotImplementedError("dropout != 0 is not used by any reference config and not built")

        v_src = vocabulary.get_vocab_size(namespace=source_namespace)
        v_tgt = vocabulary.get_vocab_size(namespace=target_namespace)
        self._source_embedder = _TokenEmbedder("tokens", v_src, input_size, self._pad_index)
        self._encoder = _Encoder(input_size, hidden_size, num_layers, dropout)
        # SimpleSeq2Seq: target embedding dim = source embedding dim; decoder dim = encoder dim
        self._target_embedder = nn.Embedding(v_tgt, input_size)
        nn.init.xavier_uniform_(self._target_embedder.weight)
        self._decoder_cell = nn.LSTMCell(hidden_size + input_size, hidden_size)
        self._output_projection_layer = nn.Linear(hidden_size, v_tgt)

        self._log2_perplexity = Average()
        self._sequence_accuracy = Average()
        self._unigram_recall = Average()
        # SimpleSeq2Seq(use_bleu=True): BLEU over evaluation predictions, special indices excluded
        self._bleu = BLEU(exclude_indices={self._pad_index, self._end_index, self._start_index})
        # row offset of this rank's shard in the global batch (keeps the sample stream shard-invariant)
        self.sample_row_offset = 0
        self.__dict__["_derived_cache"] = DerivedParams()  # (not a submodule, not part of the state_dict)

    def _derived(self) -> Optional[Dict[str, torch.Tensor]]:
        """Packed recurrent weights and bias sums of this model (``DerivedParams``), None for shapes the
        persistent kernels are not built for."""
        lstm, cell = self._encoder._module, self._decoder_cell
        Hd = cell.hidden_size
        if lstm.hidden_size != 256 or Hd != 256 or lstm.weight_hh_l0.device.type != "cuda":
            return None
        def specs():
            w_c = cell.weight_ih[:, :Hd]
            return lstm_derived_specs(lstm) + [
                ("d.c", "pack", w_c, None), ("d.hh", "pack", cell.weight_hh, None),
                ("d.cT", "packT", w_c, None), ("d.hhT", "packT", cell.weight_hh, None),
                ("d.b", "sum", cell.bias_ih, cell.bias_hh)]

        return self._derived_cache.get(lstm_derived_params(lstm) + [cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh],
                                       specs)

    # ---------------------------------------------------------------------------------------------
    def forward(
        self,
        source_tokens: torch.LongTensor,
        target_tokens: Optional[torch.LongTensor] = None,
        decoding_strategy: str = "sampling",
        need_predictions: bool = True,
    ) -> Dict[str, torch.Tensor]:
        return self.decode(self.encode(source_tokens), target_tokens, decoding_strategy, need_predictions)

    def encode(self, source_tokens: torch.LongTensor) -> Dict[str, torch.Tensor]:
        """Encoder half of ``forward`` (reference seq2seq_base.py ``_encode`` + ``_init_decoder_state``).
        Rows are independent, so a trainer may encode one batch once and ``decode`` row subsets of
        the state in different modes (``select_rows``)."""
        if source_tokens.device.type != "cuda":
            raise _hip.HipLibraryError("seq2seq input on %s: the HIP path needs a ROCm device" % source_tokens.device)
        pad, bos, eos = self._pad_index, self._start_index, self._end_index
        # boundaries (@start@ is not encoded), mask and index of the last real token: one launch
        src, fmask, last = _TokenPrep.run(source_tokens, pad, bos, eos, drop_first=True, want_mask=True)
        enc, h = self._encoder.forward_tokens(self._source_embedder.embedding, src, fmask, derived=self._derived(), last=last)
        return {"enc": enc, "h": h, "fmask": fmask}

    @staticmethod
    def select_rows(state: Dict[str, torch.Tensor], rows: torch.LongTensor) -> Dict[str, torch.Tensor]:
        return {k: v.index_select(0, rows) for k, v in state.items()}

    @staticmethod
    def split_rows(state: Dict[str, torch.Tensor], n: int):
        """The state of rows ``[0, n)`` and of rows ``[n, B)`` -- views, no copies (a trainer that wants two row SETS
        decoded differently encodes the batch in that order: rows are independent)."""
        first, rest = {}, {}
        for k, v in state.items():
            if v.requires_grad:
                first[k], rest[k] = _SplitRows.apply(v, n)
            else:
                first[k], rest[k] = v[:n], v[n:]
        return first, rest

    def decode(
        self,
        state: Dict[str, torch.Tensor],
        target_tokens: Optional[torch.LongTensor] = None,
        decoding_strategy: str = "sampling",
        need_predictions: bool = True,
        seed: Optional[int] = None,
    ) -> Dict[str, torch.Tensor]:
        """``need_predictions=False`` (teacher forcing only): skip drawing the per-step predictions from the
        teacher-forced distributions (reference :196-220) -- training iterations never read them.  ``seed``: the sampler
        seed a ``decode_prepare`` of this pass already drew (its pairing fell through): one draw per pass either way, so
        paired and unpaired schedules sample the same programs from the same torch seed."""
        if decoding_strategy not in ("sampling", "greedy"):
            raise ValueError("decoding_strategy must be 'sampling' or 'greedy'")
        pad, bos, eos = self._pad_index, self._start_index, self._end_index
        enc, h, fmask = state["enc"], state["h"], state["fmask"]
        tgt = None
        if target_tokens is not None:
            tgt = _TokenPrep.run(target_tokens, pad, bos, eos, drop_first=False, want_mask=False)[0]
        B = enc.size(0)

        steps = tgt.size(1) - 1 if tgt is not None else self._max_decoding_steps
        greedy = decoding_strategy == "greedy"
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator: no device sync
        Hd = h.size(1)
        w_ih = self._decoder_cell.weight_ih
        # the cell's input is cat(attended, embedding)
        w_c, w_e = _SplitColumns.apply(w_ih, Hd) if w_ih.requires_grad and torch.is_grad_enabled() else (w_ih[:, :Hd], w_ih[:, Hd:])
        w_p, b_p = self._output_projection_layer.weight, self._output_projection_layer.bias
        fused = Hd == 256 and enc.size(1) <= 64 and w_p.size(0) <= 128
        derived = self._derived() if fused else None
        if derived is not None:
            bias = _Alias.apply(self._decoder_cell.bias_ih, self._decoder_cell.bias_hh, derived["d.b"])
            packs = (derived["d.c"], derived["d.hh"], derived["d.cT"], derived["d.hhT"])
        else:
            bias = self._decoder_cell.bias_ih + self._decoder_cell.bias_hh
            packs = None
        if fused:
            args = (pad, self._unk_index, bos, packs)
            if tgt is not None:  # teacher forcing: every step's input embedding is known up front
                emb = self._target_embedder
                etable = _TokenTable.apply(emb.weight, w_e, bias, emb.padding_idx)
                hs, _ = _AttnLSTMDecoder.apply(None, etable, enc, fmask, h, w_c, self._decoder_cell.weight_hh, w_p, b_p,
                                               0, steps, seed, self.sample_row_offset, *args, tgt[:, :steps])
            else:  # free running: the kernel also picks each step's token
                etable = F.linear(self._target_embedder.weight, w_e, bias)
                hs, raw = _AttnLSTMDecoder.apply(None, etable, enc, fmask, h, w_c, self._decoder_cell.weight_hh, w_p,
                                                 b_p, 2 if greedy else 1, steps, seed, self.sample_row_offset, *args)
            # one GEMM for all steps; its weight gradient [V, B*T] x [B*T, H] -- 44-100 output rows over a 13 000-47 000 long
            # reduction -- through the K-split of wgrad_gemm (as a plain mm: 80-144 us at 3-17 TFLOP/s, scripts/r05_gemm_sites.py)
            logits_all = linear_rows(hs, w_p, b_p)
            if tgt is not None:
                output_dict = {"loss": sequence_nll(logits_all, tgt[:, 1:], tgt[:, 1:], pad, 1e-13)}
                if need_predictions or not self.training:
                    # predictions are drawn / arg-maxed from the teacher-forced distributions (reference :196-220)
                    raw, _ = choose_tokens(logits_all.reshape(B * steps, -1), greedy, seed, self.sample_row_offset * steps,
                                           0, pad, self._unk_index, bos)
                    output_dict["predictions"] = self._trim_predictions(raw.view(B, steps))
            else:
                predictions = self._trim_predictions(raw)
                output_dict = {"predictions": predictions, "loss": sequence_nll(logits_all, raw, predictions, pad, 1e-12)}
            ce = output_dict["loss"]
            predictions = output_dict.get("predictions")
        else:
            _note_slow_path("decoder", "hidden size %d, %d source positions, %d target tokens (the persistent decoder kernel is "
                            "built for 256 / <= 64 / <= 128)" % (Hd, enc.size(1), w_p.size(0)))
            raw, logits_all, logprobs = self._decode_stepwise(enc, fmask, h, torch.zeros_like(h), tgt, steps, greedy, seed)
            predictions = self._trim_predictions(raw)
            pmask = (predictions != pad).float()
            sequence_logprobs = (logprobs * pmask).sum(-1) / (pmask.sum(-1) + 1e-12)
            output_dict = {"predictions": predictions, "loss": -sequence_logprobs}
            if tgt is not None:
                tmask = tgt != pad
                ce = sequence_cross_entropy(logits_all, tgt[:, 1:], tmask[:, 1:])
                output_dict["loss"] = ce
        if tgt is not None:
            if not self.training:
                self._record_metrics(predictions, tgt[:, 1:], ce)
                self._bleu(predictions, tgt)  # (reference :260: against the targets WITH their @start@, as allennlp)
        return output_dict

    # ---- two teacher-forced decodes of a training iteration side by side -------------------------------------------
    def decode_prepare(self, state: Dict[str, torch.Tensor], target_tokens: Optional[torch.LongTensor] = None,
                       decoding_strategy: str = "sampling"):
        """First half of a training ``decode`` whose persistent-kernel launch is to be shared with another pass
        (``decode_pair``): everything up to the launch.  ``target_tokens`` given: teacher forced; ``None``: free running
        (the kernel samples / arg-maxes each step's token).  Returns ``None`` when this pass cannot take part (shapes
        outside the fused kernels, evaluation: metrics want predictions from the teacher-forced distributions)."""
        enc, h, fmask = state["enc"], state["h"], state["fmask"]
        w_p, b_p = self._output_projection_layer.weight, self._output_projection_layer.bias
        Hd = h.size(1)
        if not (self.training and torch.is_grad_enabled() and enc.is_cuda and Hd == 256 and enc.size(1) <= 64 and w_p.size(0) <= 128):
            return None
        # the pair kernels are multi-CU only: PNMN_DECODER_CLUSTER=0, or a device that cannot host eight resident
        # workgroups per tile (the library then reports no workspace), leave both passes to decode()
        if os.environ.get("PNMN_DECODER_CLUSTER", "1") == "0" or \
                int(_hip.lib().pnmn_attn_lstm_multi_workspace_bytes(enc.size(0), 0)) <= 0:
            return None
        derived = self._derived()
        if derived is None:
            return None
        pad, bos, eos = self._pad_index, self._start_index, self._end_index
        w_ih = self._decoder_cell.weight_ih
        w_c, w_e = _SplitColumns.apply(w_ih, Hd) if w_ih.requires_grad else (w_ih[:, :Hd], w_ih[:, Hd:])
        bias = _Alias.apply(self._decoder_cell.bias_ih, self._decoder_cell.bias_hh, derived["d.b"])
        emb = self._target_embedder
        meta = {"packs": (derived["d.c"], derived["d.hh"], derived["d.cT"], derived["d.hhT"]), "start": bos}
        tgt = None
        if target_tokens is not None:
            tgt = _TokenPrep.run(target_tokens, pad, bos, eos, drop_first=False, want_mask=False)[0]
            steps = tgt.size(1) - 1
            etable = _TokenTable.apply(emb.weight, w_e, bias, emb.padding_idx)
            torch.randint(0, 2 ** 62, (1,))  # (decode() draws one seed per pass, used or not: keep the generator in step)
            meta.update(mode=0, in_tokens=tgt[:, :steps], T=steps)
        else:
            steps = self._max_decoding_steps
            etable = F.linear(emb.weight, w_e, bias)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator: no device sync
            meta.update(mode=2 if decoding_strategy == "greedy" else 1, T=steps, pad=pad, unk=self._unk_index, seed=seed,
                        row_offset=self.sample_row_offset, w_p=w_p, b_p=b_p)
        return {"model": self, "tgt": tgt, "steps": steps, "etable": etable, "enc": enc, "fmask": fmask, "h": h, "w_c": w_c,
                "w_hh": self._decoder_cell.weight_hh, "meta": meta}

    def decode_finish(self, prep, hs: torch.Tensor, raw: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Second half: the output projection over all steps and the per-row loss -- the cross entropy of the targets
        (reference :235-254) or, free running, the length-normalised negative log-probability of the trimmed samples
        (reference :222-233)."""
        proj = self._output_projection_layer
        logits_all = linear_rows(hs, proj.weight, proj.bias)  # (weight gradient through wgrad_gemm's K-split, see decode)
        tgt = prep["tgt"]
        if tgt is not None:
            return {"loss": sequence_nll(logits_all, tgt[:, 1:], tgt[:, 1:], self._pad_index, 1e-13)}
        predictions = self._trim_predictions(raw)
        return {"predictions": predictions, "loss": sequence_nll(logits_all, raw, predictions, self._pad_index, 1e-12)}

    def _decode_stepwise(self, enc, fmask, h, c, tgt, steps, greedy, seed):
        """Step-by-step decoding for shapes the persistent kernel is not built for (hidden != 256,
        more than 64 source positions or 128 target tokens): a GEMM per step + the cell kernel."""
        B = enc.size(0)
        pad, bos = self._pad_index, self._start_index
        last = fmask.new_full((B,), bos, dtype=torch.long)
        w_ih_t = self._decoder_cell.weight_ih.t()
        w_hh_t = self._decoder_cell.weight_hh.t()
        bias = self._decoder_cell.bias_ih + self._decoder_cell.bias_hh
        step_logits, step_logprobs, step_predictions = [], [], []
        for t in range(steps):
            inputs = tgt[:, t] if tgt is not None else last
            e = self._target_embedder(inputs)
            scores = torch.bmm(enc, h.unsqueeze(-1)).squeeze(-1)
            weights = masked_softmax(scores, fmask)
            attended = torch.bmm(weights.unsqueeze(1), enc).squeeze(1)
            x = torch.cat((attended, e), -1)
            gates = torch.addmm(torch.addmm(bias, x, w_ih_t), h, w_hh_t)
            h, c = lstm_cell_pointwise(gates, c)
            logits = self._output_projection_layer(h)
            last, _ = choose_tokens(logits, greedy, seed, self.sample_row_offset, t, pad, self._unk_index, bos)
            step_predictions.append(last.unsqueeze(1))
            step_logits.append(logits.unsqueeze(1))
            step_logprobs.append(F.log_softmax(logits, dim=-1).gather(1, last.unsqueeze(1)))
        return torch.cat(step_predictions, 1), torch.cat(step_logits, 1), torch.cat(step_logprobs, 1)

    def _trim_predictions(self, predictions: torch.LongTensor) -> torch.LongTensor:
        """Keep each row up to and including its first @end@; a row starting with @end@ becomes all
        padding, a row without @end@ is kept whole (reference :278-293), without leaving the device."""
        if predictions.device.type == "cuda" and predictions.dtype == torch.long and predictions.dim() == 2:
            predictions = predictions.contiguous()
            out = torch.empty_like(predictions)
            _hip.check(_hip.lib().pnmn_trim_predictions(predictions.data_ptr(), predictions.size(0), predictions.size(1),
                                                        self._end_index, out.data_ptr(), _hip.stream_ptr(predictions.device)),
                       "trim_predictions")
            return out
        steps = predictions.size(1)
        is_end = predictions == self._end_index
        has_end = is_end.any(1, keepdim=True)
        first = is_end.float().argmax(1, keepdim=True)  # first occurrence
        pos = torch.arange(steps, device=predictions.device).unsqueeze(0)
        keep = torch.where(has_end, (pos <= first) & (first > 0), torch.ones_like(is_end))
        return predictions * keep

    @torch.no_grad()
    def _record_metrics(self, predictions, relevant_targets, ce) -> None:
        n = relevant_targets.size(1)
        pred = predictions[:, :n]
        mask = relevant_targets != self._pad_index
        correct = ((pred == relevant_targets) | ~mask).all(1).float().mean()
        # unigram recall: fraction of gold tokens that appear anywhere in the prediction
        hit = (relevant_targets.unsqueeze(2) == pred.unsqueeze(1)).any(2) & mask
        recall = (hit.sum(1).float() / mask.sum(1).clamp(min=1).float()).mean()
        self._log2_perplexity(ce.mean())
        self._sequence_accuracy(correct)
        self._unigram_recall(recall)

    def get_metrics(self, reset: bool = True) -> Dict[str, float]:
        if self.training:
            return {}
        return {
            **self._bleu.get_metric(reset=True),  # (the reference resets BLEU unconditionally, :367)
            "perplexity": 2 ** self._log2_perplexity.get_metric(reset=reset),
            "sequence_accuracy": self._sequence_accuracy.get_metric(reset=reset),
            "word_error_rate": 1 - self._unigram_recall.get_metric(reset=reset),
        }


def decode_pair(prep_a, prep_b):
    """The launches of two prepared decodes (``Seq2SeqBase.decode_prepare``) as one, then each model's second half.
    Returns the two output dicts ({"loss": per-row loss[, "predictions"]})."""
    hs_a, tok_a, hs_b, tok_b = _AttnLSTMDecoderPair.apply(
        prep_a["etable"], prep_a["enc"], prep_a["fmask"], prep_a["h"], prep_a["w_c"], prep_a["w_hh"],
        prep_b["etable"], prep_b["enc"], prep_b["fmask"], prep_b["h"], prep_b["w_c"], prep_b["w_hh"],
        (prep_a["meta"], prep_b["meta"]))
    return prep_a["model"].decode_finish(prep_a, hs_a, tok_a), prep_b["model"].decode_finish(prep_b, hs_b, tok_b)


def decode_pair_launch(prep_a, prep_b):
    """The forward launch of two prepared decodes NOW, without a graph node: returns what ``decode_group`` needs to create
    the node later (``.outs`` = (hidden states a, tokens a, hidden states b, tokens b))."""
    cap = _Capture()
    with torch.no_grad():
        cap.outs = _AttnLSTMDecoderPair.forward(
            cap, prep_a["etable"], prep_a["enc"], prep_a["fmask"], prep_a["h"], prep_a["w_c"], prep_a["w_hh"],
            prep_b["etable"], prep_b["enc"], prep_b["fmask"], prep_b["h"], prep_b["w_c"], prep_b["w_hh"],
            (prep_a["meta"], prep_b["meta"]))
    return cap


def decode_group(preps, pre, pre_index):
    """One graph node for the prepared decodes ``preps`` (``_AttnLSTMDecoderGroup``): ``pre_index[k]`` = position of pass k in
    the earlier ``decode_pair_launch`` ``pre``, or None (launched now); then each model's second half.  Returns the
    passes' output dicts."""
    flat, metas = [], []
    for p, j in zip(preps, pre_index):
        flat += [p["etable"], p["enc"], p["fmask"], p["h"], p["w_c"], p["w_hh"]]
        m = dict(p["meta"])
        m["pre"] = j
        metas.append(m)
    outs = _AttnLSTMDecoderGroup.apply(*flat, metas, pre)
    return [p["model"].decode_finish(p, outs[2 * k], outs[2 * k + 1]) for k, p in enumerate(preps)]
