"""CPU restatement of the seq2seq models (test infrastructure -- see oracle/__init__).

**Parity unpinned by the reference**: ProgramGenerator / QuestionReconstructor subclass
``allennlp.models.encoder_decoders.SimpleSeq2Seq`` and ProgramPrior uses AllenNLP's embedder /
wrapper / loss (``allennlp==0.9.0``, requirements.txt:1), which is neither vendored in
/root/reference nor installable here, and the reference has no tests or golden outputs for this
path.  What follows restates (a) the reference's own overrides -- probnmn/modules/seq2seq_base.py
:101-155 (forward), :157-276 (_forward_loop), :278-293 (_trim_predictions), :295-341 (_get_loss),
probnmn/models/program_prior.py:80-155 -- and (b) the published AllenNLP 0.9.0 semantics of the
pieces they call, on top of torch primitives that exist on both boxes
(``nn.LSTM`` with packed sequences, ``nn.LSTMCell``, ``F.embedding``, ``F.softmax``).  It is pinned
by tests/test_seq2seq_oracle.py against those primitives and hand-derived known answers.

Functional over a ``state_dict`` with the reference's key names:
  _source_embedder.token_embedder_tokens.weight, _encoder._module.{weight,bias}_{ih,hh}_l{0,1},
  _target_embedder.weight, _decoder_cell.{weight,bias}_{ih,hh}, _output_projection_layer.{weight,bias}
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

PAD, UNK, START, END = 0, 1, 2, 3


def seq2seq_param_shapes(v_src: int, v_tgt: int, input_size=256, hidden=256, layers=2) -> Dict[str, Tuple[int, ...]]:
    s = {"_source_embedder.token_embedder_tokens.weight": (v_src, input_size)}
    for layer in range(layers):
        cin = input_size if layer == 0 else hidden
        s["_encoder._module.weight_ih_l%d" % layer] = (4 * hidden, cin)
        s["_encoder._module.weight_hh_l%d" % layer] = (4 * hidden, hidden)
        s["_encoder._module.bias_ih_l%d" % layer] = (4 * hidden,)
        s["_encoder._module.bias_hh_l%d" % layer] = (4 * hidden,)
    s["_target_embedder.weight"] = (v_tgt, input_size)  # SimpleSeq2Seq: target_embedding_dim = source dim
    s["_decoder_cell.weight_ih"] = (4 * hidden, hidden + input_size)
    s["_decoder_cell.weight_hh"] = (4 * hidden, hidden)
    s["_decoder_cell.bias_ih"] = (4 * hidden,)
    s["_decoder_cell.bias_hh"] = (4 * hidden,)
    s["_output_projection_layer.weight"] = (v_tgt, hidden)
    s["_output_projection_layer.bias"] = (v_tgt,)
    return s


def prior_param_shapes(vocab: int, input_size=256, hidden=256, layers=2) -> Dict[str, Tuple[int, ...]]:
    s = {"_embedder.token_embedder_programs.weight": (vocab, input_size)}
    for layer in range(layers):
        cin = input_size if layer == 0 else hidden
        s["_encoder._module.weight_ih_l%d" % layer] = (4 * hidden, cin)
        s["_encoder._module.weight_hh_l%d" % layer] = (4 * hidden, hidden)
        s["_encoder._module.bias_ih_l%d" % layer] = (4 * hidden,)
        s["_encoder._module.bias_hh_l%d" % layer] = (4 * hidden,)
    s["_projection_layer.weight"] = (input_size, hidden)
    # _output_layer.weight is tied to the embedding (program_prior.py:60-62)
    return s


# ---- AllenNLP 0.9.0 pieces -----------------------------------------------------------------------
def add_sentence_boundary_token_ids(tokens: torch.Tensor, mask: torch.Tensor, bos: int, eos: int):
    """allennlp.nn.util.add_sentence_boundary_token_ids, 2-D case: (B,T) -> (B,T+2); assumes
    right padding; @end@ goes right after the last real token."""
    lengths = mask.sum(dim=1).long()
    B, T = tokens.shape
    out = tokens.new_zeros(B, T + 2)
    out[:, 1:-1] = tokens
    out[:, 0] = bos
    for i in range(B):
        out[i, int(lengths[i]) + 1] = eos
    return out, (out != 0).long()


def masked_softmax(vector: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """allennlp.nn.util.masked_softmax (memory_efficient=False)."""
    mask = mask.float()
    result = F.softmax(vector * mask, dim=-1)
    result = result * mask
    return result / (result.sum(dim=-1, keepdim=True) + 1e-13)


def sequence_cross_entropy_with_logits(logits, targets, weights, eps=1e-13):
    """allennlp.nn.util.sequence_cross_entropy_with_logits(average=None): per-sequence mean."""
    weights = weights.float()
    logp = F.log_softmax(logits, dim=-1)
    nll = -torch.gather(logp, 2, targets.unsqueeze(-1).long()).squeeze(-1)
    nll = nll * weights
    return nll.sum(1) / (weights.sum(1) + eps)


def packed_lstm(sd, prefix: str, x: torch.Tensor, mask: torch.Tensor, layers: int = 2) -> torch.Tensor:
    """PytorchSeq2SeqWrapper(nn.LSTM(batch_first=True))(x, mask): pack by length, zero initial
    state, run, pad the output back to x.size(1) with zeros.  A unidirectional LSTM's state at a
    valid step never depends on later steps, so this equals running every row over all T steps
    and zeroing the outputs past its length; written with ``lstm_cell`` so gradients reach ``sd``.
    tests/test_seq2seq_oracle.py checks it (values and gradients) against the real
    ``nn.LSTM`` + ``pack_padded_sequence`` / ``pad_packed_sequence``."""
    B, T, _ = x.shape
    valid = mask.bool()
    inp = x
    for layer in range(layers):
        w_ih, w_hh = sd[prefix + "weight_ih_l%d" % layer], sd[prefix + "weight_hh_l%d" % layer]
        b_ih, b_hh = sd[prefix + "bias_ih_l%d" % layer], sd[prefix + "bias_hh_l%d" % layer]
        h = x.new_zeros(B, w_hh.shape[1])
        c = x.new_zeros(B, w_hh.shape[1])
        outs = []
        for t in range(T):
            h, c = lstm_cell(inp[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            outs.append(h)
        inp = torch.stack(outs, 1)
    return inp * valid.unsqueeze(-1).to(inp.dtype)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell arithmetic (gate order i, f, g, o)."""
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def trim_predictions(predictions: torch.Tensor, end_index: int = END) -> torch.Tensor:
    """seq2seq_base.py:278-293: keep up to and including the first @end@; a row whose first token
    is @end@ becomes all zeros; a row without @end@ is kept whole."""
    out = torch.zeros_like(predictions)
    for i, row in enumerate(predictions):
        idx = row.tolist()
        if end_index in idx:
            e = idx.index(end_index)
            if e > 0:
                out[i, : e + 1] = row[: e + 1]
        else:
            out[i] = row
    return out


# ---- Seq2SeqBase.forward ---------------------------------------------------------------------------
def seq2seq_forward(
    sd: Dict[str, torch.Tensor],
    source_tokens: torch.Tensor,
    target_tokens: Optional[torch.Tensor] = None,
    decoding_strategy: str = "sampling",
    max_decoding_steps: int = 26,
    forced_predictions: Optional[torch.Tensor] = None,
    generator: Optional[torch.Generator] = None,
):
    """Returns {"predictions", "loss", "logits", "step_logprobs"}.  ``forced_predictions`` (B, steps)
    replaces the multinomial draw (test hook: the sampler's stream differs between devices)."""
    src, _ = add_sentence_boundary_token_ids(source_tokens, source_tokens != PAD, START, END)
    tgt = None
    if target_tokens is not None:
        tgt, _ = add_sentence_boundary_token_ids(target_tokens, target_tokens != PAD, START, END)
    src = src[:, 1:]  # the @start@ of the source is not encoded (seq2seq_base.py:139)

    # _encode
    emb = F.embedding(src, sd["_source_embedder.token_embedder_tokens.weight"], padding_idx=PAD)
    src_mask = (src != PAD).long()
    enc = packed_lstm(sd, "_encoder._module.", emb, src_mask)
    # _init_decoder_state: last valid encoder output, zero context
    B = src.size(0)
    last = src_mask.sum(1) - 1
    h = enc[torch.arange(B), last]
    c = torch.zeros_like(h)

    steps = tgt.size(1) - 1 if tgt is not None else max_decoding_steps
    last_predictions = src.new_full((B,), START)
    step_logits, step_logprobs, step_predictions = [], [], []
    fmask = src_mask.float()
    for t in range(steps):
        inputs = tgt[:, t] if tgt is not None else last_predictions
        e = F.embedding(inputs, sd["_target_embedder.weight"])
        # _prepare_attended_input: dot-product attention over the encoder outputs
        scores = torch.bmm(enc, h.unsqueeze(-1)).squeeze(-1)
        weights = masked_softmax(scores, fmask)
        attended = torch.bmm(weights.unsqueeze(1), enc).squeeze(1)
        x = torch.cat((attended, e), -1)
        h, c = lstm_cell(x, h, c, sd["_decoder_cell.weight_ih"], sd["_decoder_cell.weight_hh"],
                         sd["_decoder_cell.bias_ih"], sd["_decoder_cell.bias_hh"])
        logits = F.linear(h, sd["_output_projection_layer.weight"], sd["_output_projection_layer.bias"])
        probs = F.softmax(logits, dim=-1)
        logprobs = F.log_softmax(logits, dim=-1)
        if forced_predictions is not None:
            predicted = forced_predictions[:, t]
        elif decoding_strategy == "greedy":
            predicted = torch.max(probs, 1)[1]
        else:
            p = probs.detach().clone()
            p[:, PAD] = 0
            p[:, UNK] = 0
            p[:, START] = 0
            predicted = torch.multinomial(p, 1, generator=generator).squeeze(1)
        last_predictions = predicted
        step_predictions.append(predicted.unsqueeze(1))
        step_logits.append(logits.unsqueeze(1))
        step_logprobs.append(logprobs[torch.arange(B), predicted].unsqueeze(1))

    raw = torch.cat(step_predictions, 1)
    predictions = trim_predictions(raw)
    logprobs = torch.cat(step_logprobs, 1)
    pmask = (predictions != PAD).float()
    seq_logprob = (logprobs * pmask).sum(-1) / (pmask.sum(-1) + 1e-12)
    out = {"predictions": predictions, "loss": -seq_logprob, "step_logprobs": logprobs,
           "logits": torch.cat(step_logits, 1), "raw_predictions": raw}
    if tgt is not None:
        tmask = tgt != PAD
        out["loss"] = sequence_cross_entropy_with_logits(out["logits"], tgt[:, 1:], tmask[:, 1:])
    return out


# ---- ProgramPrior.forward (loss only; the sampled "predictions" are unused by the trainers) -----------
def program_prior_loss(sd: Dict[str, torch.Tensor], program_tokens: torch.Tensor) -> torch.Tensor:
    toks, _ = add_sentence_boundary_token_ids(program_tokens, program_tokens != PAD, START, END)
    mask = (toks != PAD).long()
    w = sd["_embedder.token_embedder_programs.weight"]
    emb = F.embedding(toks, w, padding_idx=PAD)
    enc = packed_lstm(sd, "_encoder._module.", emb, mask)
    proj = F.linear(enc, sd["_projection_layer.weight"])
    logits = F.linear(proj, w)  # tied output layer, no bias
    return sequence_cross_entropy_with_logits(logits[:, :-1], toks[:, 1:], mask[:, 1:])


def program_prior_sample(sd: Dict[str, torch.Tensor], forced: torch.Tensor, max_sequence_length: int):
    """ProgramPrior.sample (program_prior.py:174-301) with the multinomial draws replaced by ``forced``
    (num_samples, max_sequence_length - 1): stepwise LSTM from @start@, per-step log-probability gathered
    from log_softmax of the PROJECTION (the reference's own quirk, :243-244), trim at the first @end@,
    masked mean, most likely first."""
    n = forced.size(0)
    w = sd["_embedder.token_embedder_programs.weight"]
    layers = 2
    hidden = sd["_encoder._module.weight_hh_l0"].shape[1]
    h = [torch.zeros(n, hidden) for _ in range(layers)]
    c = [torch.zeros(n, hidden) for _ in range(layers)]
    last = torch.full((n,), START, dtype=torch.long)
    lps, preds = [], []
    for t in range(max_sequence_length - 1):
        x = F.embedding(last, w, padding_idx=PAD)
        for layer in range(layers):
            h[layer], c[layer] = lstm_cell(x, h[layer], c[layer], sd["_encoder._module.weight_ih_l%d" % layer],
                                           sd["_encoder._module.weight_hh_l%d" % layer],
                                           sd["_encoder._module.bias_ih_l%d" % layer], sd["_encoder._module.bias_hh_l%d" % layer])
            x = h[layer]
        proj = F.linear(x, sd["_projection_layer.weight"])
        last = forced[:, t]
        preds.append(last.unsqueeze(1))
        lps.append(F.log_softmax(proj, dim=-1).gather(1, last.unsqueeze(1)))
    predictions = trim_predictions(torch.cat(preds, 1))
    mask = (predictions != PAD).float()
    seq = (torch.cat(lps, 1) * mask).sum(-1) / (mask.sum(-1) + 1e-12)
    order = (-seq).sort()[1]
    return {"predictions": predictions[order], "loss": -seq[order]}
