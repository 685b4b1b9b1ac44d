from .nmn import NeuralModuleNetwork

__all__ = ["NeuralModuleNetwork"]
