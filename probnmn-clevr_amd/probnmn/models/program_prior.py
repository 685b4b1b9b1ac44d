"""``ProgramPrior``: LSTM language model over programs, p(z) (reference:
probnmn/models/program_prior.py:15-155).  Input and output embeddings are tied.  ``forward`` gives
the per-sequence cross entropy used as -log p(z) in the REINFORCE reward, plus per-position
samples (unused by the trainers).  ``sample`` (reference :174-301) is outside the hot path."""
from typing import Dict

import torch
from torch import nn
from torch.nn import functional as F

from probnmn import _hip
from probnmn.modules.seq2seq_base import _Encoder, _TokenEmbedder, add_sentence_boundary_token_ids, sequence_nll
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
        toks = add_sentence_boundary_token_ids(program_tokens, self._pad_index, self._start_index, self._end_index)
        mask = toks != self._pad_index
        encoded = self._encoder.forward_tokens(self._embedder.embedding, toks, mask)
        logits = self._output_layer(self._projection_layer(encoded))
        loss = sequence_nll(logits[:, :-1], toks[:, 1:], toks[:, 1:], self._pad_index, 1e-13)
        if not self.training:
            self._log2_perplexity(loss.mean())
        if not need_predictions:
            return {"loss": loss}
        with torch.no_grad():
            probs = F.softmax(logits, dim=-1).clone()
            forbidden = self.__dict__.get("_forbidden")
            if forbidden is None or forbidden.device != probs.device:
                forbidden = torch.tensor([self._start_index, self._pad_index, self._unk_index]).to(probs.device)
                self.__dict__["_forbidden"] = forbidden  # cached: building it costs a host -> device copy
            probs.index_fill_(2, forbidden, 0.0)
            B, T, V = probs.shape
            predictions = torch.multinomial(probs.view(B * T, V), 1).view(B, T)
            predictions = predictions[:, :-1] * mask[:, 1:]
        return {"predictions": predictions, "loss": loss}

    def get_metrics(self, reset: bool = True) -> Dict[str, float]:
        return {"perplexity": 2 ** self._log2_perplexity.get_metric(reset=reset)}
