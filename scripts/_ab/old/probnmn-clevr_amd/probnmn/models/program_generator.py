"""``ProgramGenerator``: questions -> programs (reference: probnmn/models/program_generator.py:9-59).
The inference network q(z|x): a :class:`Seq2SeqBase` from the "questions" to the "programs"
namespace.  The reference hard-codes 26 decoding steps; it is a keyword here (BASELINE config 5
decodes up to 40)."""
from probnmn.modules.seq2seq_base import Seq2SeqBase


class ProgramGenerator(Seq2SeqBase):
    def __init__(self, vocabulary, input_size: int = 256, hidden_size: int = 256, num_layers: int = 2,
                 dropout: float = 0.0, max_decoding_steps: int = 26):
        super().__init__(vocabulary, source_namespace="questions", target_namespace="programs",
                         input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                         dropout=dropout, max_decoding_steps=max_decoding_steps)

    @classmethod
    def from_config(cls, config):
        from probnmn.vocabulary import Vocabulary

        _C = config
        return cls(vocabulary=Vocabulary.from_files(_C.DATA.VOCABULARY),
                   input_size=_C.PROGRAM_GENERATOR.INPUT_SIZE, hidden_size=_C.PROGRAM_GENERATOR.HIDDEN_SIZE,
                   num_layers=_C.PROGRAM_GENERATOR.NUM_LAYERS, dropout=_C.PROGRAM_GENERATOR.DROPOUT)
