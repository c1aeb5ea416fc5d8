"""Session stream for the CARS hot path: sampler -> collate straight into pinned staging -> int32 wire format.

The reference feeds CARS from a DataLoader over vectorised sessions, batched by the session-length-bucketing sampler
(/root/reference/neuroir/inputters/multitask/data.py:42-72) and collated to int64 tensors
(neuroir/inputters/multitask/vector.py:82-149).  At GPU speeds that host path, not the kernels, bounds a 224k-session
stream, so this module keeps the reference's BATCH COMPOSITION (samplers.session_length_batches) and changes
the transport:

* WIRE FORMAT: one contiguous block per batch -- every integer field (document_words, document_lens, source_words, source_lens)
  as int32, back to back, then the float32 click labels.  Half the PCIe bytes of the reference's LongTensors; the device widens the
  integer block with ONE kernel (nir_widen_ids_i32) into the int64 tensors the entry points read.  `WireLayout` is the single
  description of that block shared by the writer (host) and graph_runner.StreamingSessionPredictor (device views).
* COLLATE IN PLACE: a batch is gathered from the corpus with one numpy `take(..., out=view of the pinned slot)` per field -- no
  intermediate batch tensors, no second host copy, and numpy releases the GIL during the copy so a producer thread overlaps the
  consumer's enqueue calls.

`SyntheticSessionCorpus` is the MSMARCO-shaped synthetic stream of SURVEY.md section 8(d): n sessions with
S ~ clip(Poisson(4.84) + 2, 2, 16) queries, N candidates per query, Zipf token ids, one click per query.  Token bodies come from a pool of
`pool` distinct sessions per length (a 224k-session corpus of fresh tokens would be 20 GB of host memory and minutes of generation);
the session ORDER, LENGTHS and batch composition are per-session.
"""
import numpy as np
import torch

from .. import synth
from . import samplers

INT_FIELDS = ("document_words", "document_lens", "source_words", "source_lens")


class WireLayout(object):
    """Byte layout of one batch on the wire (and in the device staging buffer): int32 fields back to back, then float32 labels.
    Field starts are multiples of 4 elements (16 bytes)."""

    def __init__(self, B, S, N, QL, DL, groups=0):
        """groups > 0: the block also carries `click_max` int32 [groups] -- the click mask's batch-wide count m (cars.py:285-289) of each of
        the `groups` sampler batches the B sessions were cut from (sharding.StreamShardPlan mode "pair": a rank holds B/G sessions of a batch and
        cannot derive the batch's m from its own labels; the collator knows every label and ships the integer)."""
        self.B, self.S, self.N, self.QL, self.DL, self.groups = int(B), int(S), int(N), int(QL), int(DL), int(groups)
        self.shapes = {"document_words": (B, S, N, DL), "document_lens": (B, S, N), "source_words": (B, S, QL), "source_lens": (B, S),
                       "document_labels": (B, S, N)}
        off = 0
        self.offset = {}
        for k in INT_FIELDS:
            self.offset[k] = off
            off += (int(np.prod(self.shapes[k])) + 3) // 4 * 4
        if self.groups:
            self.shapes["click_max"] = (self.groups,)
            self.offset["click_max"] = off
            off += (self.groups + 3) // 4 * 4
        self.n_int = off                                   # int32 elements (padded), = the count handed to nir_widen_ids_i32
        self.offset["document_labels"] = off               # float32 elements from here on (same 4-byte units)
        self.n_words = off + (int(np.prod(self.shapes["document_labels"])) + 3) // 4 * 4
        self.nbytes = 4 * self.n_words
        self.pairs = B * S * N

    def views(self, buf_u8):
        """typed views of a byte buffer (host numpy array or torch tensor, >= nbytes) -> {field: array/tensor of the field's shape}."""
        out = {}
        if isinstance(buf_u8, np.ndarray):
            w = buf_u8[:self.nbytes].view(np.int32)
            for k in INT_FIELDS:
                n = int(np.prod(self.shapes[k]))
                out[k] = w[self.offset[k]:self.offset[k] + n].reshape(self.shapes[k])
            o = self.offset["document_labels"]
            out["document_labels"] = buf_u8[:self.nbytes].view(np.float32)[o:o + self.pairs].reshape(self.shapes["document_labels"])
            if self.groups:
                out["click_max"] = w[self.offset["click_max"]:self.offset["click_max"] + self.groups]
            return out
        w = buf_u8[:self.nbytes].view(torch.int32)
        for k in INT_FIELDS:
            n = int(np.prod(self.shapes[k]))
            out[k] = w[self.offset[k]:self.offset[k] + n].view(self.shapes[k])
        o = self.offset["document_labels"]
        out["document_labels"] = buf_u8[:self.nbytes].view(torch.float32)[o:o + self.pairs].view(self.shapes["document_labels"])
        if self.groups:
            out["click_max"] = w[self.offset["click_max"]:self.offset["click_max"] + self.groups]
        return out

    def wide_views(self, buf_i64):
        """views of the widened int64 buffer [n_int] -> the int64 tensors the networks consume."""
        out = {}
        for k in INT_FIELDS:
            n = int(np.prod(self.shapes[k]))
            out[k] = buf_i64[self.offset[k]:self.offset[k] + n].view(self.shapes[k])
        return out


class SyntheticSessionCorpus(object):
    def __init__(self, n_sessions=223876, n_cands=50, qlen=4, dlen=64, vocab=100000, seed=1013, pool=128, full_length=True,
                 fixed_len=None, s_min=2, s_max=16, multi_click=False):
        """fixed_len: every session has this many queries (the fixed-shape configs); else S ~ clip(Poisson(4.84) + 2, s_min, s_max).
        multi_click: 1-6 clicked candidates per query instead of one (the click mask's batch-wide count then differs between batches)."""
        rng = np.random.default_rng(seed)
        self.N, self.QL, self.DL, self.V = int(n_cands), int(qlen), int(dlen), int(vocab)
        if fixed_len is not None:
            self.lengths = np.full(n_sessions, int(fixed_len), np.int64)
        else:
            self.lengths = np.clip(rng.poisson(4.84, size=n_sessions) + 2, s_min, s_max).astype(np.int64)
        self.pool = {}
        self.slot = np.empty(n_sessions, np.int64)          # which pool body session i uses
        for S in np.unique(self.lengths):
            S = int(S)
            P = min(int(pool), int((self.lengths == S).sum()))
            b = synth.session_batch(P, S, self.N, self.QL, self.DL, self.V, seed=seed + 7919 * S, full_length=full_length, multi_click=multi_click)
            self.pool[S] = {k: np.ascontiguousarray(v.numpy().astype(np.float32 if k == "document_labels" else np.int32)) for k, v in b.items()}
            idx = np.flatnonzero(self.lengths == S)
            self.slot[idx] = rng.integers(0, P, size=len(idx))
            # clicked candidates of a session's most-clicked query: the per-session term of the click mask's batch-wide count m
            self.pool[S]["_clicks"] = (self.pool[S]["document_labels"] != 0).sum(-1).max(-1).astype(np.int32)

    def __len__(self):
        return len(self.lengths)

    def batches(self, batch_size, shuffle=True, seed=1013):
        """the reference sampler's batch composition (inputters/multitask/data.py:42-72): equal-length sessions per batch, full batches
        only, batches shuffled -> list of index lists."""
        rng = np.random.RandomState(seed)
        return samplers.session_length_batches(self.lengths, batch_size, shuffle=shuffle, rng=rng)

    def layout(self, S, batch_size, groups=0):
        return WireLayout(batch_size, S, self.N, self.QL, self.DL, groups)

    def click_max(self, idx, batch_size=None):
        """m of cars.py:285-289 for the sampler batch(es) in `idx` (batch_size: idx holds len(idx) / batch_size batches back to back):
        the largest number of clicked candidates of any query of the batch -> int (one batch) or int32 array [groups]."""
        S = int(self.lengths[idx[0]])
        c = self.pool[S]["_clicks"][self.slot[np.asarray(idx, dtype=np.int64)]]
        if batch_size is None or len(idx) == batch_size:
            return int(c.max())
        return c.reshape(-1, int(batch_size)).max(1).astype(np.int32)

    def collate_into(self, idx, host_u8, whole=None, batch_size=None):
        """write the batch `idx` (sessions of one length) into the (pinned) byte buffer in wire format; returns its WireLayout.
        whole (optional): idx is this rank's share of the sampler batch(es) `whole` (batch_size sessions each): the block then also
        carries their click counts (`click_max`)."""
        S = int(self.lengths[idx[0]])
        groups = 0 if whole is None else len(whole) // int(batch_size)
        lay = self.layout(S, len(idx), groups)
        v = lay.views(host_u8)
        rows = self.slot[np.asarray(idx, dtype=np.int64)]
        p = self.pool[S]
        for k in INT_FIELDS + ("document_labels",):
            np.take(p[k], rows, axis=0, out=v[k], mode="clip")     # (mode='raise' would buffer `out`; rows are valid by construction)
        if groups:
            v["click_max"][:] = self.click_max(whole, batch_size)
        return lay

    def batch_tensors(self, idx):
        """the same batch as the reference's int64 / float32 tensors (oracle side of the parity tests)."""
        S = int(self.lengths[idx[0]])
        rows = self.slot[np.asarray(idx, dtype=np.int64)]
        p = self.pool[S]
        return {k: torch.from_numpy(p[k][rows].astype(np.float32 if k == "document_labels" else np.int64)) for k in p if not k.startswith("_")}
