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
        """layers.py:23-27: [B, P] ids -> dropout(word embeddings) [B, P, d].  Not on the hot path (its kernels gather from
        word_embeddings.table themselves); HIP gather + HIP dropout operators."""
        from .. import autograd as A
        return A.dropout(self.word_embeddings(sequence.unsqueeze(2)), self.dropout.p, self.training)


class Encoder(nn.Module):
    def __init__(self, rnn_type, input_size, bidirection, nlayers, nhid, dropout_rnn):
        super().__init__()
        self.encoder = RNNEncoder(rnn_type, input_size, bidirection, nlayers, nhid, dropout_rnn)

    def forward(self, input, input_len, init_states=None):
        return self.encoder(input, input_len, init_states)
