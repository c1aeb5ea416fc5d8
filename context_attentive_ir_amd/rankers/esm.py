"""ESM -- embedding space model (drop-in for neuroir.rankers.esm.ESM, /root/reference/neuroir/rankers/esm.py:9-45).

scores[b,n] = cos(mean_L emb(q_b), mean_L emb(d_bn)); means divide by the PADDED length.  One fused
gather-reduce kernel (nir_esm_score): the embedding rows are read once, nothing is materialised.
"""
import torch
import torch.nn as nn

from .. import lib
from ..constants import PAD
from ..modules import Embeddings


class ESM(nn.Module, lib.IdCheck):
    def __init__(self, args):
        super().__init__()
        self.word_embeddings = Embeddings(args.emsize, args.src_vocab_size, PAD)

    def forward(self, batch_queries, query_len, batch_docs, doc_len):
        assert batch_queries.shape[0] == batch_docs.shape[0]
        lib.require_device(batch_queries, batch_docs, self.word_embeddings.table)
        q, d = self._clean_ids(batch_queries, batch_docs, self.word_embeddings.table.shape[0])
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        table = self.word_embeddings.table
        scores = torch.empty(B, N, device=q.device, dtype=torch.float32)
        if B == 0:   # empty batch: nothing to enqueue (zero-size tensors have no device pointer)
            return scores
        lib.check(lib.load().nir_esm_score(lib.ptr(q), lib.ptr(d), B, N, QL, DL, lib.ptr(table), table.shape[0],
                                           table.shape[1], lib.ptr(scores), lib.stream()), "nir_esm_score")
        return scores
