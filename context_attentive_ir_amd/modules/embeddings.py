"""Word-embedding table with the reference's state-dict layout.

Mirror of neuroir.modules.embeddings.Embeddings (/root/reference/neuroir/modules/embeddings.py:88-252) for the
word-only case the hot path uses: one nn.Embedding(padding_idx=PAD) stored under
`make_embedding.emb_luts.0.weight` so reference checkpoints load unchanged.
On the HIP path the table is never "looked up" into a [.., L, E] tensor: kernels take `table` and gather
rows inside their operand loads (train mode: autograd.embed).  `forward` is the same HIP gather for modules outside the hot path.
"""
import torch
import torch.nn as nn


class Embeddings(nn.Module):
    def __init__(self, word_vec_size, word_vocab_size, word_padding_idx):
        super().__init__()
        self.word_vec_size = word_vec_size
        self.word_padding_idx = word_padding_idx
        self.embedding_size = word_vec_size
        self.make_embedding = nn.Sequential()
        self.make_embedding.add_module(
            "emb_luts", nn.ModuleList([nn.Embedding(word_vocab_size, word_vec_size, padding_idx=word_padding_idx)]))

    @property
    def word_lut(self):
        return self.make_embedding[0][0]

    @property
    def table(self):
        """[V, E] fp32 table handed to the kernels."""
        return self.word_lut.weight

    def init_word_vectors(self, vocabulary, embeddings_index, fixed):
        """Same contract as embeddings.py:213-226: rows for known tokens, zeros elsewhere."""
        pretrained = torch.zeros(len(vocabulary), self.word_vec_size)
        for i in range(len(vocabulary)):
            tok = vocabulary.ind2tok[i]
            if tok in embeddings_index:
                pretrained[i] = embeddings_index[tok]
        with torch.no_grad():
            self.word_lut.weight.copy_(pretrained)      # in-place on the parameter itself: bumps _version (lib.PackCache)
        if fixed:
            self.word_lut.weight.requires_grad = False

    def load_pretrained_vectors(self, emb_file, fixed):
        if emb_file:
            with torch.no_grad():
                self.word_lut.weight.copy_(torch.load(emb_file))
            if fixed:
                self.word_lut.weight.requires_grad = False

    def forward(self, source):
        """embeddings.py:243-252: source `[.., .., 1]` (one word feature per position; the reference documents `[len x batch x nfeat]`, its
        callers pass `[batch x len x 1]`) -> `[.., .., embedding_size]`.  The hot-path networks never call this -- their kernels gather rows
        of `self.table` inside their operand loads -- but any other module that shares the embedder gets the lookup here, on the HIP gather
        operator (nir_embed_f32; differentiable: autograd.embed; an id outside the table raises the device flag lib.IdCheck polls)."""
        from .. import autograd as A
        if source.dim() < 1 or (source.dim() >= 3 and source.shape[-1] != 1):
            raise NotImplementedError("Embeddings.forward: only the word feature is implemented (nfeat = 1; the reference's char / feature merges are out of scope)")
        ids = source[..., 0] if source.dim() >= 3 else source
        return A.embed(ids, self.table, self.word_padding_idx)
