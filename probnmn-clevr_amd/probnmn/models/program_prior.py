"""``ProgramPrior``: LSTM language model over programs, p(z) (reference:
probnmn/models/program_prior.py:15-155).  Input and output embeddings are tied.  ``forward`` gives
the per-sequence cross entropy used as -log p(z) in the REINFORCE reward, plus per-position
samples (unused by the trainers).  ``sample`` (reference :174-301) is outside the hot path."""
from typing import Dict

import torch
from torch import nn
from torch.nn import functional as F

from probnmn import _hip
from probnmn.modules.seq2seq_base import (DerivedParams, _Encoder, _TokenEmbedder, _TokenPrep, lstm_derived_params,
                                          lstm_derived_specs,
                                          sequence_nll)
from probnmn.running_metrics import Average


class ProgramPrior(nn.Module):
    def __init__(self, vocabulary, input_size: int = 256, hidden_size: int = 128, num_layers: int = 2,
                 dropout: float = 0.0):
        super().__init__()
        self.vocabulary = vocabulary
        self._start_index = vocabulary.get_token_index("@start@", namespace="programs")
        self._end_index = vocabulary.get_token_index("@end@", namespace="programs")
        self._pad_index = vocabulary.get_token_index("@@PADDING@@", namespace="programs")
        self._unk_index = vocabulary.get_token_index("@@UNKNOWN@@", namespace="programs")
        vocab_size = vocabulary.get_vocab_size(namespace="programs")
        self._embedder = _TokenEmbedder("programs", vocab_size, input_size, self._pad_index)
        self._encoder = _Encoder(input_size, hidden_size, num_layers, dropout)
        self._projection_layer = nn.Linear(hidden_size, input_size, bias=False)
        self._output_layer = nn.Linear(input_size, vocab_size, bias=False)
        self._output_layer.weight = self._embedder.embedding.weight  # tied
        self._log2_perplexity = Average()
        self.__dict__["_derived_cache"] = DerivedParams()  # (packed recurrent weights; not part of the state_dict)

    def _derived(self):
        lstm = self._encoder._module
        if lstm.hidden_size != 256 or lstm.weight_hh_l0.device.type != "cuda":
            return None
        return self._derived_cache.get(lstm_derived_params(lstm), lambda: lstm_derived_specs(lstm))

    @classmethod
    def from_config(cls, config):
        from probnmn.vocabulary import Vocabulary

        _C = config
        return cls(vocabulary=Vocabulary.from_files(_C.DATA.VOCABULARY), input_size=_C.PROGRAM_PRIOR.INPUT_SIZE,
                   hidden_size=_C.PROGRAM_PRIOR.HIDDEN_SIZE, num_layers=_C.PROGRAM_PRIOR.NUM_LAYERS,
                   dropout=_C.PROGRAM_PRIOR.DROPOUT)

    def forward(self, program_tokens: torch.Tensor, need_predictions: bool = True) -> Dict[str, torch.Tensor]:
        # ``need_predictions=False``: skip the per-position samples (reference :119-143), which no trainer reads
        if program_tokens.device.type != "cuda":
            raise _hip.HipLibraryError("program prior input on %s: the HIP path needs a ROCm device" % program_tokens.device)
        toks, fmask, _ = _TokenPrep.run(program_tokens, self._pad_index, self._start_index, self._end_index,
                                        drop_first=False, want_mask=True)
        encoded = self._encoder.forward_tokens(self._embedder.embedding, toks, fmask, derived=self._derived())
        logits = self._output_layer(self._projection_layer(encoded))
        loss = sequence_nll(logits[:, :-1], toks[:, 1:], toks[:, 1:], self._pad_index, 1e-13)
        if not need_predictions:  # (the trainers' reward path: no samples, no validation metric)
            return {"loss": loss}
        if not self.training:
            self._log2_perplexity(loss.mean())
        with torch.no_grad():
            probs = F.softmax(logits, dim=-1).clone()
            forbidden = self.__dict__.get("_forbidden")
            if forbidden is None or forbidden.device != probs.device:
                forbidden = torch.tensor([self._start_index, self._pad_index, self._unk_index]).to(probs.device)
                self.__dict__["_forbidden"] = forbidden  # cached: building it costs a host -> device copy
            probs.index_fill_(2, forbidden, 0.0)
            B, T, V = probs.shape
            predictions = torch.multinomial(probs.view(B * T, V), 1).view(B, T)
            predictions = predictions[:, :-1] * (toks[:, 1:] != self._pad_index)
        return {"predictions": predictions, "loss": loss}

    @torch.no_grad()
    def sample(self, num_samples: int = 1, max_sequence_length: int = 28, _forced=None) -> Dict[str, torch.Tensor]:
        """Free-running categorical samples from the prior, most likely first (reference :174-301; inspection
        only, no trainer calls it -- written with torch ops, one step at a time, on the model's device).
        Reproduced as the reference has it, including that the per-step log-probability is gathered from
        ``log_softmax`` of the 256-wide PROJECTION, not of the vocabulary logits (:243-244,257-259).
        ``_forced`` (steps = max_sequence_length - 1 columns) replaces the draws in tests."""
        device = self._output_layer.weight.device
        lstm = self._encoder._module
        last = torch.full((num_samples, 1), self._start_index, dtype=torch.long, device=device)
        h = torch.zeros(lstm.num_layers, num_samples, lstm.hidden_size, device=device)
        c = torch.zeros_like(h)
        step_logprobs, step_predictions = [], []
        for t in range(max_sequence_length - 1):
            encoded, (h, c) = lstm(self._embedder.embedding(last), (h, c))
            projection = self._projection_layer(encoded)
            probabilities = F.softmax(self._output_layer(projection), dim=-1)
            logprobs = F.log_softmax(projection, dim=-1)
            probabilities[:, :, [self._start_index, self._pad_index, self._unk_index]] = 0
            last = torch.multinomial(probabilities.squeeze(1), 1) if _forced is None else _forced[:, t:t + 1].to(device)
            step_predictions.append(last)
            step_logprobs.append(torch.gather(logprobs, 2, last.unsqueeze(1)).squeeze(-1))
        raw = torch.cat(step_predictions, 1)
        # keep up to and including the first @end@; a row starting with @end@ becomes padding (:270-280)
        steps = raw.size(1)
        is_end = raw == self._end_index
        first = is_end.float().argmax(1, keepdim=True)
        pos = torch.arange(steps, device=device).unsqueeze(0)
        keep = torch.where(is_end.any(1, keepdim=True), (pos <= first) & (first > 0), torch.ones_like(is_end))
        predictions = raw * keep
        mask = (predictions != self._pad_index).float()
        sequence_logprobs = (torch.cat(step_logprobs, 1) * mask).sum(-1) / (mask.sum(-1) + 1e-12)
        order = (-sequence_logprobs).sort()[1]
        return {"predictions": predictions[order], "loss": -sequence_logprobs[order]}

    def get_metrics(self, reset: bool = True) -> Dict[str, float]:
        return {"perplexity": 2 ** self._log2_perplexity.get_metric(reset=reset)}
