"""Input contract of the hot path: vectorised examples -> the padded batch tensors the
networks consume.  Same dict keys, shapes, dtypes and padding rules as the reference's collate functions
(/root/reference/neuroir/inputters/ranker/vector.py:39-90 and neuroir/inputters/multitask/vector.py:82-149):
zero (= PAD) padding up to the batch maximum of the per-example maxima, int64 ids / lengths / ranker labels,
float32 CARS click labels.

Built for the device hand-off instead of per-row `copy_` calls: every tensor of the batch is a typed view of ONE
contiguous (optionally pinned) byte buffer, 16-byte aligned per field, filled by flat numpy scatters -- so the batch
reaches the GPU with a single H2D copy (the layout graph_runner.GraphedPredictor replays from).  The whole buffer is
returned under the extra key '_buffer'.
"""
import numpy as np
import torch

_ALIGN = 16


def _alloc(fields, pin):
    """fields: [(name, shape, torch dtype)] -> dict of zeroed tensor views over one byte buffer (+ '_buffer')."""
    offs, total = [], 0
    for _, shape, dt in fields:
        nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty(0, dtype=dt).element_size()
        offs.append((total, nbytes))
        total = (total + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    buf = torch.zeros(max(total, _ALIGN), dtype=torch.uint8)
    if pin and torch.cuda.is_available():
        buf = buf.pin_memory()
    out = {"_buffer": buf}
    for (name, shape, dt), (o, n) in zip(fields, offs):
        out[name] = buf[o:o + n].view(dt).view(*shape)
    return out


def _scatter_rows(dst2d, rows):
    """dst2d [R, L] (zeroed) <- ragged rows: one flat index scatter instead of R copy_ calls."""
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    if lens.sum() == 0:
        return lens
    flat = np.concatenate([np.asarray(r, dtype=np.int64).reshape(-1) for r in rows])
    row_of = np.repeat(np.arange(len(rows), dtype=np.int64), lens)
    col_of = np.arange(flat.size, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    dst2d.numpy()[row_of, col_of] = flat
    return lens


def ranker_batchify(batch, pin=False):
    """batch: list of vectorised examples with the reference's keys ('id', 'query_words', 'doc_words' (list of N id
    vectors), 'label', 'num_candidates', 'max_doc_len', 'max_query_len') -> {'batch_size', 'ids', 'doc_rep' [B,N,DL],
    'doc_len' [B,N], 'que_rep' [B,QL], 'que_len' [B], 'label' [B,N]} (all int64)."""
    B = len(batch)
    N = int(batch[0]["num_candidates"])
    DL = max(int(b["max_doc_len"]) for b in batch)
    QL = max(int(b["max_query_len"]) for b in batch)
    for b in batch:
        if len(b["doc_words"]) != N:
            raise RuntimeError("example %r has %d candidates, the batch expects %d" % (b.get("id"), len(b["doc_words"]), N))
    t = _alloc([("doc_rep", (B, N, DL), torch.int64), ("doc_len", (B, N), torch.int64), ("que_rep", (B, QL), torch.int64),
                ("que_len", (B,), torch.int64), ("label", (B, N), torch.int64)], pin)
    dlen = _scatter_rows(t["doc_rep"].view(B * N, DL), [d for b in batch for d in b["doc_words"]])
    t["doc_len"].view(-1).numpy()[:] = dlen
    t["que_len"].numpy()[:] = _scatter_rows(t["que_rep"], [b["query_words"] for b in batch])
    t["label"].numpy()[:] = np.stack([np.asarray(b["label"], dtype=np.int64).reshape(N) for b in batch])
    t.update(batch_size=B, ids=[b["id"] for b in batch])
    return t


def session_batchify(batch, pin=False):
    """batch: list of vectorised sessions with the reference's keys ('id', 'session_len', 'num_candidates',
    'source_words' [S,QLi], 'source_lens' [S], 'document_words' [S,N,DLi], 'document_lens' [S,N], 'document_labels'
    [S,N], 'target_words'/'target_seq' [S-1,TLi], 'target_lens' [S-1], 'max_source_len', 'max_target_len',
    'max_document_len') -> the CARS batch: 'source_words' [B,S,QL], 'source_lens' [B,S], 'document_words'
    [B,S,N,DL], 'document_lens' [B,S,N], 'document_labels' [B,S,N] float32, 'target_words'/'target_seq' [B,S-1,TL],
    'target_lens' [B,S-1].  All sessions of a batch must have the same length (the reference's sampler buckets by it)."""
    B = len(batch)
    S, N = int(batch[0]["session_len"]), int(batch[0]["num_candidates"])
    if any(int(b["session_len"]) != S for b in batch):
        raise AssertionError("all the sessions of a batch must have the same length")
    QL = max(int(b["max_source_len"]) for b in batch)
    TL = max(int(b["max_target_len"]) for b in batch)
    DL = max(int(b["max_document_len"]) for b in batch)
    t = _alloc([("document_words", (B, S, N, DL), torch.int64), ("document_lens", (B, S, N), torch.int64),
                ("document_labels", (B, S, N), torch.float32), ("source_words", (B, S, QL), torch.int64),
                ("source_lens", (B, S), torch.int64), ("target_words", (B, S - 1, TL), torch.int64),
                ("target_seq", (B, S - 1, TL), torch.int64), ("target_lens", (B, S - 1), torch.int64)], pin)
    for i, b in enumerate(batch):       # per-session blocks are dense already: one slice assignment each
        t["source_lens"][i] = torch.as_tensor(b["source_lens"])
        t["source_words"][i, :, :int(b["max_source_len"])] = torch.as_tensor(b["source_words"])
        t["document_lens"][i] = torch.as_tensor(b["document_lens"])
        t["document_labels"][i] = torch.as_tensor(b["document_labels"]).float()
        t["document_words"][i, :, :, :int(b["max_document_len"])] = torch.as_tensor(b["document_words"])
        t["target_lens"][i] = torch.as_tensor(b["target_lens"])
        t["target_words"][i, :, :int(b["max_target_len"])] = torch.as_tensor(b["target_words"])
        t["target_seq"][i, :, :int(b["max_target_len"])] = torch.as_tensor(b["target_seq"])
    t.update(batch_size=B, ids=[b["id"] for b in batch], session_len=S,
             source_tokens=[b.get("source_tokens") for b in batch], target_tokens=[b.get("target_tokens") for b in batch])
    return t


def flat_examples(que_ids, doc_ids, labels, num_candidates=None, force_pad=None):
    """Convenience for tests / synthetic streams: ragged id lists -> vectorised ranker examples.
    que_ids: list of B id lists; doc_ids: list of B lists of N id lists; labels: [B][N];
    force_pad=(max_query_len, max_doc_len) mirrors args.force_pad (vector.py:18-21)."""
    out = []
    for i, (q, ds, lab) in enumerate(zip(que_ids, doc_ids, labels)):
        n = num_candidates if num_candidates is not None else len(ds)
        out.append({"id": i, "query_words": torch.as_tensor(np.asarray(q, dtype=np.int64)),
                    "doc_words": [torch.as_tensor(np.asarray(d, dtype=np.int64)) for d in ds],
                    "label": torch.as_tensor(np.asarray(lab, dtype=np.int64)), "num_candidates": n,
                    "max_doc_len": force_pad[1] if force_pad else max(len(d) for d in ds),
                    "max_query_len": force_pad[0] if force_pad else len(q)})
    return out
