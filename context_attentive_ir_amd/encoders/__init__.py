from .rnn_encoder import RNNEncoder

__all__ = ["RNNEncoder"]
