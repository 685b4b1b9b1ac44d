"""SECOND, independent CPU restatement of the seq2seq models (test infrastructure only).

oracle/seq2seq_oracle.py is functional code over a state_dict, with its own LSTM cell arithmetic.  The
path it restates is *parity unpinned* by the reference (allennlp 0.9.0 is absent), so a single
author's reading is all that stands behind it.  This file is a second reading that shares NO code with
the first: it is built from torch.nn MODULES the way allennlp 0.9.0 builds SimpleSeq2Seq --
``nn.Embedding`` (padding_idx 0), ``nn.LSTM`` run over real ``pack_padded_sequence`` /
``pad_packed_sequence`` (what PytorchSeq2SeqWrapper does), ``nn.LSTMCell(512, 256)``, ``nn.Linear`` --
and writes the decoding loop, the attention and the losses from the reference's overrides
(probnmn/modules/seq2seq_base.py:101-341, probnmn/models/program_prior.py:80-155) and SURVEY App. A
directly.  tests/test_seq2seq_oracle.py compares the two whole-model (losses, predictions, every
gradient).  Agreement cannot make parity "pinned"; it removes single-author risk.
"""
import torch
from torch import nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

PAD, UNK, BOS, EOS = 0, 1, 2, 3


def _with_boundaries(tokens: torch.Tensor) -> torch.Tensor:
    """@start@ in front, @end@ right behind the last non-padding token of every row."""
    rows = []
    width = tokens.size(1) + 2
    for row in tokens.tolist():
        body = [t for t in row if t != PAD]
        assert row[: len(body)] == body, "right padding expected"
        full = [BOS] + body + [EOS]
        rows.append(full + [PAD] * (width - len(full)))
    return torch.tensor(rows, dtype=torch.long)


class _PackedLSTM(nn.Module):
    """PytorchSeq2SeqWrapper(nn.LSTM(batch_first=True)): sort / pack / run / unpack / pad back."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self._module = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True)

    def forward(self, x, lengths):
        packed = pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = self._module(packed)
        out, _ = pad_packed_sequence(out, batch_first=True, total_length=x.size(1))
        return out


class ModuleSeq2Seq(nn.Module):
    """Parameter names are the reference's, so a state_dict moves over unchanged."""

    def __init__(self, v_src, v_tgt, input_size=256, hidden_size=256, num_layers=2, max_decoding_steps=26):
        super().__init__()
        self._source_embedder = nn.ModuleDict()  # allennlp: BasicTextFieldEmbedder({"tokens": Embedding})
        self._source_embedder["token_embedder_tokens"] = nn.Embedding(v_src, input_size, padding_idx=PAD)
        self._encoder = _PackedLSTM(input_size, hidden_size, num_layers)
        self._target_embedder = nn.Embedding(v_tgt, input_size)
        self._decoder_cell = nn.LSTMCell(hidden_size + input_size, hidden_size)
        self._output_projection_layer = nn.Linear(hidden_size, v_tgt)
        self.max_decoding_steps = max_decoding_steps

    def forward(self, source_tokens, target_tokens=None, strategy="greedy", forced=None):
        source = _with_boundaries(source_tokens)[:, 1:]
        lengths = (source != PAD).sum(1)
        keep = (torch.arange(source.size(1))[None, :] < lengths[:, None]).to(torch.float32)
        memory = self._encoder(self._source_embedder["token_embedder_tokens"](source), lengths)
        hidden = torch.stack([memory[b, n - 1] for b, n in enumerate(lengths.tolist())])
        cell = torch.zeros_like(hidden)
        targets = _with_boundaries(target_tokens) if target_tokens is not None else None
        steps = targets.size(1) - 1 if targets is not None else self.max_decoding_steps
        previous = torch.full((source.size(0),), BOS, dtype=torch.long)
        all_logits, picked, picked_logprob = [], [], []
        for t in range(steps):
            fed = targets[:, t] if targets is not None else previous
            # dot-product attention, allennlp's masked softmax: softmax of the masked scores, re-masked, renormalised
            similarity = torch.einsum("bsh,bh->bs", memory, hidden)
            attention = torch.softmax(similarity * keep, dim=1) * keep
            attention = attention / (attention.sum(1, keepdim=True) + 1e-13)
            context = torch.einsum("bs,bsh->bh", attention, memory)
            hidden, cell = self._decoder_cell(torch.cat([context, self._target_embedder(fed)], dim=1), (hidden, cell))
            logits = self._output_projection_layer(hidden)
            log_distribution = logits - torch.logsumexp(logits, dim=1, keepdim=True)
            if forced is not None:
                choice = forced[:, t]
            elif strategy == "greedy":
                choice = logits.argmax(1)
            else:
                weights = log_distribution.detach().exp()
                weights[:, [PAD, UNK, BOS]] = 0.0
                choice = torch.multinomial(weights, 1).squeeze(1)
            previous = choice
            all_logits.append(logits)
            picked.append(choice)
            picked_logprob.append(log_distribution.gather(1, choice[:, None]).squeeze(1))
        raw = torch.stack(picked, 1)
        predictions = raw.clone()
        for b, row in enumerate(raw.tolist()):  # keep up to and including the first @end@
            if EOS in row:
                e = row.index(EOS)
                predictions[b, e + 1:] = PAD
                if e == 0:
                    predictions[b] = PAD
        logprob = torch.stack(picked_logprob, 1)
        live = (predictions != PAD).to(logprob.dtype)
        loss = -(logprob * live).sum(1) / (live.sum(1) + 1e-12)
        if targets is not None:
            gold = targets[:, 1:]
            weight = (gold != PAD).to(logprob.dtype)
            logits = torch.stack(all_logits, 1)
            token_nll = nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), gold.reshape(-1), reduction="none")
            loss = (token_nll.view_as(gold) * weight).sum(1) / (weight.sum(1) + 1e-13)
        return {"predictions": predictions, "loss": loss}


class ModulePrior(nn.Module):
    def __init__(self, vocab, input_size=256, hidden_size=256, num_layers=2):
        super().__init__()
        self._embedder = nn.ModuleDict()
        self._embedder["token_embedder_programs"] = nn.Embedding(vocab, input_size, padding_idx=PAD)
        self._encoder = _PackedLSTM(input_size, hidden_size, num_layers)
        self._projection_layer = nn.Linear(hidden_size, input_size, bias=False)

    def forward(self, program_tokens):
        tokens = _with_boundaries(program_tokens)
        lengths = (tokens != PAD).sum(1)
        embedding = self._embedder["token_embedder_programs"]
        states = self._encoder(embedding(tokens), lengths)
        logits = self._projection_layer(states) @ embedding.weight.t()  # output layer tied to the embedding
        gold, weight = tokens[:, 1:], (tokens[:, 1:] != PAD).to(logits.dtype)
        nll = nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.size(-1)), gold.reshape(-1), reduction="none")
        return (nll.view_as(gold) * weight).sum(1) / (weight.sum(1) + 1e-13)
