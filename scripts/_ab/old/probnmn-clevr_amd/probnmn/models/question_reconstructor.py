"""``QuestionReconstructor``: programs -> questions (reference:
probnmn/models/question_reconstructor.py:10-61).  The generative model p(x|z): a
:class:`Seq2SeqBase` from the "programs" to the "questions" namespace, 45 decoding steps."""
from probnmn.modules.seq2seq_base import Seq2SeqBase


class QuestionReconstructor(Seq2SeqBase):
    def __init__(self, vocabulary, input_size: int = 256, hidden_size: int = 256, num_layers: int = 2,
                 dropout: float = 0.0, max_decoding_steps: int = 45):
        super().__init__(vocabulary, source_namespace="programs", target_namespace="questions",
                         input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                         dropout=dropout, max_decoding_steps=max_decoding_steps)

    @classmethod
    def from_config(cls, config):
        from probnmn.vocabulary import Vocabulary

        _C = config
        return cls(vocabulary=Vocabulary.from_files(_C.DATA.VOCABULARY),
                   input_size=_C.QUESTION_RECONSTRUCTOR.INPUT_SIZE,
                   hidden_size=_C.QUESTION_RECONSTRUCTOR.HIDDEN_SIZE,
                   num_layers=_C.QUESTION_RECONSTRUCTOR.NUM_LAYERS, dropout=_C.QUESTION_RECONSTRUCTOR.DROPOUT)
