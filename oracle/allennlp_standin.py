"""A minimal stand-in for the parts of ``allennlp==0.9.0`` the reference's seq2seq code imports (test
infrastructure; build container only -- used by oracle/make_seq2seq_overrides.py and nothing else).

WHY: the reference's ``Seq2SeqBase`` / ``ProgramPrior`` (probnmn/modules/seq2seq_base.py:3-16,
probnmn/models/program_prior.py:3-8) import AllenNLP 0.9.0, which is neither vendored under /root/reference nor
installable here, so those files could only ever be *restated* (oracle/seq2seq_oracle.py, oracle/seq2seq_modules.py),
never *executed*.  With this stand-in in ``sys.modules`` the reference's OWN lines run unmodified from where they
lie: ``forward`` (:101-155: boundary tokens, the dropped source @start@), ``_forward_loop`` (:157-276: loop order,
teacher forcing vs. fed-back samples, the zeroed pad/unk/start probabilities before the draw, log-prob bookkeeping,
length normalisation), ``_trim_predictions`` (:278-293), ``_get_loss`` (:295-341: target alignment) and
``ProgramPrior.forward`` (program_prior.py:80-155).

WHAT IT IS NOT: AllenNLP.  ``SimpleSeq2Seq._encode / _init_decoder_state / _prepare_output_projections``,
``DotProductAttention``, ``PytorchSeq2SeqWrapper``, ``Embedding``, ``BasicTextFieldEmbedder`` and ``nn.util`` below
are written from AllenNLP 0.9.0's published behaviour on top of torch modules, in the module style of
oracle/seq2seq_modules.py.  Vectors generated through it pin the reference-held lines; the AllenNLP-held pieces stay
restated, and the seq2seq parity stays "unpinned" in that part (oracle/__init__).
Metrics (BLEU, SequenceAccuracy, UnigramRecall) are inert recorders: no arithmetic under test depends on them.
"""
import sys
import types
from typing import Dict

import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

START_SYMBOL, END_SYMBOL = "@start@", "@end@"


# ---- allennlp.nn.util ------------------------------------------------------------------------------
def add_sentence_boundary_token_ids(tensor, mask, sentence_begin_token, sentence_end_token):
    lengths = mask.sum(dim=1).detach().cpu().numpy()
    shape = list(tensor.shape)
    shape[1] += 2
    out = tensor.new_zeros(*shape)
    out[:, 1:-1] = tensor
    out[:, 0] = sentence_begin_token
    for i, j in enumerate(lengths):
        out[i, j + 1] = sentence_end_token
    return out, (out != 0).long()


def masked_softmax(vector, mask, dim: int = -1):
    mask = mask.float()
    while mask.dim() < vector.dim():
        mask = mask.unsqueeze(1)
    result = F.softmax(vector * mask, dim=dim)
    result = result * mask
    return result / (result.sum(dim=dim, keepdim=True) + 1e-13)


def weighted_sum(matrix, attention):
    return attention.unsqueeze(1).bmm(matrix).squeeze(1)


def get_text_field_mask(text_field_tensors: Dict[str, torch.Tensor]):
    (tensor,) = text_field_tensors.values()
    return (tensor != 0).long()


def get_final_encoder_states(encoder_outputs, mask, bidirectional=False):
    last = mask.sum(1).long() - 1
    b, _, d = encoder_outputs.size()
    idx = last.view(-1, 1, 1).expand(b, 1, d)
    return encoder_outputs.gather(1, idx).squeeze(1)


def sequence_cross_entropy_with_logits(logits, targets, weights, average="batch"):
    weights = weights.float()
    non_batch_dims = tuple(range(1, weights.dim()))
    weights_batch_sum = weights.sum(dim=non_batch_dims)
    log_probs_flat = F.log_softmax(logits.view(-1, logits.size(-1)), dim=-1)
    nll = -torch.gather(log_probs_flat, dim=1, index=targets.view(-1, 1).long()).view(*targets.size())
    nll = nll * weights
    if average is not None:
        raise NotImplementedError("the reference only calls average=None")
    return nll.sum(non_batch_dims) / (weights_batch_sum + 1e-13)


# ---- modules ---------------------------------------------------------------------------------------
class Embedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, padding_index=None):
        super().__init__()
        self.padding_index = padding_index
        self.output_dim = embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.xavier_uniform_(self.weight)
        if padding_index is not None:
            self.weight.data[padding_index].fill_(0)

    def get_output_dim(self):
        return self.output_dim

    def forward(self, inputs):
        return F.embedding(inputs, self.weight, padding_idx=self.padding_index)


class BasicTextFieldEmbedder(nn.Module):
    def __init__(self, token_embedders: Dict[str, nn.Module]):
        super().__init__()
        self._keys = sorted(token_embedders)
        for k in self._keys:
            self.add_module("token_embedder_%s" % k, token_embedders[k])

    def get_output_dim(self):
        return sum(getattr(self, "token_embedder_%s" % k).get_output_dim() for k in self._keys)

    def forward(self, text_field_input):
        return torch.cat([getattr(self, "token_embedder_%s" % k)(text_field_input[k]) for k in self._keys], dim=-1)


class PytorchSeq2SeqWrapper(nn.Module):
    def __init__(self, module):
        super().__init__()
        self._module = module

    def get_output_dim(self):
        return self._module.hidden_size

    def is_bidirectional(self):
        return self._module.bidirectional

    def forward(self, inputs, mask):
        lengths = mask.long().sum(1)
        packed = pack_padded_sequence(inputs, lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = self._module(packed)
        out, _ = pad_packed_sequence(out, batch_first=True, total_length=inputs.size(1))
        return out


class DotProductAttention(nn.Module):
    def forward(self, vector, matrix, matrix_mask=None):
        return masked_softmax(matrix.bmm(vector.unsqueeze(-1)).squeeze(-1), matrix_mask)


class _Recorder:
    """Inert metric: remembers that it was called (no arithmetic under test reads it)."""

    def __init__(self, *a, **k):
        self.calls = 0

    def __call__(self, *a, **k):
        self.calls += 1

    def get_metric(self, reset=False):
        return {"BLEU": 0.0} if type(self).__name__ == "BLEU" else 0.0


class BLEU(_Recorder):
    pass


class SimpleSeq2Seq(nn.Module):
    """The constructor arguments the reference passes (seq2seq_base.py:86-94) and the three methods it leaves
    untouched (:143-148, :201)."""

    def __init__(self, vocab, source_embedder, encoder, max_decoding_steps, attention=None, target_namespace="tokens",
                 use_bleu=True):
        super().__init__()
        self.vocab = vocab
        self._target_namespace = target_namespace
        self._scheduled_sampling_ratio = 0.0
        self._start_index = vocab.get_token_index(START_SYMBOL, target_namespace)
        self._end_index = vocab.get_token_index(END_SYMBOL, target_namespace)
        self._bleu = BLEU() if use_bleu else None
        self._max_decoding_steps = max_decoding_steps
        self._source_embedder = source_embedder
        self._encoder = encoder
        num_classes = vocab.get_vocab_size(target_namespace)
        self._attention = attention
        target_embedding_dim = source_embedder.get_output_dim()
        self._target_embedder = Embedding(num_classes, target_embedding_dim)
        self._encoder_output_dim = encoder.get_output_dim()
        self._decoder_output_dim = self._encoder_output_dim
        self._decoder_input_dim = self._decoder_output_dim + target_embedding_dim
        self._decoder_cell = nn.LSTMCell(self._decoder_input_dim, self._decoder_output_dim)
        self._output_projection_layer = nn.Linear(self._decoder_output_dim, num_classes)

    def _encode(self, source_tokens):
        embedded = self._source_embedder(source_tokens)
        source_mask = get_text_field_mask(source_tokens)
        return {"source_mask": source_mask, "encoder_outputs": self._encoder(embedded, source_mask)}

    def _init_decoder_state(self, state):
        batch = state["source_mask"].size(0)
        state["decoder_hidden"] = get_final_encoder_states(state["encoder_outputs"], state["source_mask"],
                                                           self._encoder.is_bidirectional())
        state["decoder_context"] = state["encoder_outputs"].new_zeros(batch, self._decoder_output_dim)
        return state

    def _prepare_output_projections(self, last_predictions, state):
        embedded = self._target_embedder(last_predictions)
        weights = self._attention(state["decoder_hidden"], state["encoder_outputs"], state["source_mask"].float())
        attended = weighted_sum(state["encoder_outputs"], weights)
        decoder_input = torch.cat((attended, embedded), -1)
        h, c = self._decoder_cell(decoder_input, (state["decoder_hidden"], state["decoder_context"]))
        state["decoder_hidden"], state["decoder_context"] = h, c
        return self._output_projection_layer(h), state


def install(vocabulary_class) -> None:
    """Register the stand-in under the import names the reference uses."""

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("allennlp")
    mod("allennlp.data", Vocabulary=vocabulary_class)
    mod("allennlp.models")
    mod("allennlp.models.encoder_decoders", SimpleSeq2Seq=SimpleSeq2Seq)
    mod("allennlp.modules")
    mod("allennlp.modules.attention", DotProductAttention=DotProductAttention)
    mod("allennlp.modules.seq2seq_encoders", PytorchSeq2SeqWrapper=PytorchSeq2SeqWrapper)
    mod("allennlp.modules.text_field_embedders", BasicTextFieldEmbedder=BasicTextFieldEmbedder)
    mod("allennlp.modules.token_embedders", Embedding=Embedding)
    mod("allennlp.nn")
    mod("allennlp.nn.util", add_sentence_boundary_token_ids=add_sentence_boundary_token_ids,
        sequence_cross_entropy_with_logits=sequence_cross_entropy_with_logits)
    mod("allennlp.training")
    mod("allennlp.training.metrics", Average=type("Average", (_Recorder,), {}),
        SequenceAccuracy=type("SequenceAccuracy", (_Recorder,), {}), UnigramRecall=type("UnigramRecall", (_Recorder,), {}),
        BooleanAccuracy=type("BooleanAccuracy", (_Recorder,), {}))
