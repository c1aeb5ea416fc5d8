"""Thin wrappers that give CARS the reference's attribute / state-dict nesting
(mirror of /root/reference/neuroir/multitask/layers.py:10-51: `embedder.word_embeddings...`,
`<x>_encoder.encoder.rnns.0...`)."""
import torch.nn as nn

from ..constants import PAD
from ..encoders import RNNEncoder
from ..modules import Embeddings


class Embedder(nn.Module):
    def __init__(self, emsize, src_vocab_size, dropout_emb):
        super().__init__()
        self.word_embeddings = Embeddings(emsize, src_vocab_size, PAD)
        self.output_size = emsize
        self.dropout = nn.Dropout(dropout_emb)

    def forward(self, sequence):
        """Fused into the consuming kernels like Embeddings.forward (layers.py:23-27 of the reference)."""
        raise NotImplementedError("Embedder.forward is fused into the consuming HIP kernels (pass ids + word_embeddings.table)")


class Encoder(nn.Module):
    def __init__(self, rnn_type, input_size, bidirection, nlayers, nhid, dropout_rnn):
        super().__init__()
        self.encoder = RNNEncoder(rnn_type, input_size, bidirection, nlayers, nhid, dropout_rnn)

    def forward(self, input, input_len, init_states=None):
        return self.encoder(input, input_len, init_states)
