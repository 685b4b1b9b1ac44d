from .nmn import NeuralModuleNetwork
from .program_generator import ProgramGenerator
from .program_prior import ProgramPrior
from .question_reconstructor import QuestionReconstructor

__all__ = ["ProgramPrior", "ProgramGenerator", "QuestionReconstructor", "NeuralModuleNetwork"]
