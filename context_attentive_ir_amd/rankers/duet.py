"""DUET (drop-in for neuroir.rankers.duet.DUET, /root/reference/neuroir/rankers/duet.py:9-208).

local model       : exact-match matrix x Conv1d(k=1, channels = doc positions) -> tanh -> fc1..fc3
distributed model : Conv1d(E->300,k=3)+tanh on q and d, pools, 1x1 conv, Hadamard with the query vector,
                    Linear over positions, two more Linear+tanh.
One C-ABI call (nir_duet_score).  The document branch of the distributed model (conv_d1 -> tanh -> max-pool -> conv_d2 -> tanh
-> Hadamard . fc2) is ONE fused kernel per tile of 64 conv positions (csrc/duet_fused.hip: embedding rows are the only HBM
input, conv weights arrive as fp16 term planes in MFMA-fragment order built once per weight version below); the query side and
the general fallback run as GEMMs with the embedding gather fused into the A operand; the exact-match "convolution" is
evaluated sparsely (only matching (doc,query) positions add a weight row).
Like the reference (hyparam.py:34-46 `force_pad`), inputs must be padded to max_query_len / max_doc_len.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as A
from .. import lib
from ..constants import PAD
from ..modules import Embeddings


class LocalModel(nn.Module):
    """Parameters of duet.py:62-121."""

    def __init__(self, args):
        super().__init__()
        self.conv1d = nn.Conv1d(args.max_doc_len, args.nfilters, args.local_filter_size)
        self.drop = nn.Dropout(args.dropout)
        self.fc1 = nn.Linear(args.max_query_len, 1)
        self.fc2 = nn.Linear(args.nfilters, args.nfilters)
        self.fc3 = nn.Linear(args.nfilters, 1)


class DistributedModel(nn.Module):
    """Parameters of duet.py:124-208."""

    def __init__(self, args):
        super().__init__()
        self.conv_q = nn.Conv1d(args.emsize, args.nfilters, args.dist_filter_size)
        self.conv_d1 = nn.Conv1d(args.emsize, args.nfilters, args.dist_filter_size)
        self.conv_d2 = nn.Conv1d(args.nfilters, args.nfilters, 1)
        self.pool_size = args.pool_size
        self.dropout = nn.Dropout(args.dropout)
        self.fc1 = nn.Linear(args.nfilters, args.nfilters)
        self.fc2 = nn.Linear(args.max_doc_len - args.pool_size - 1, 1)
        self.fc3 = nn.Linear(args.nfilters, args.nfilters)
        self.fc4 = nn.Linear(args.nfilters, 1)


class DUET(nn.Module, lib.IdCheck):
    def __init__(self, args):
        super().__init__()
        self.use_word = args.use_word
        if not self.use_word:
            raise TypeError("Non-word inputs are not supported!")
        if args.local_filter_size != 1 or args.dist_filter_size != 3:
            raise NotImplementedError("HIP DUET supports local_filter_size=1, dist_filter_size=3 (hyparam.py:34-39)")
        self.word_embeddings = Embeddings(args.emsize, args.src_vocab_size, PAD)
        self.emb_drop = nn.Dropout(p=args.dropout_emb)
        self.local_model = LocalModel(args)
        self.distributed_model = DistributedModel(args)
        self.max_doc_len, self.max_query_len = args.max_doc_len, args.max_query_len
        self._dims = dict(NF=args.nfilters, pool=args.pool_size)
        self._pack = lib.PackCache()
        # fp16 term planes of table / conv weights for the pre-split GEMM (csrc/gemm.hip: gemm_h2p_kernel).  Off by default: at the C4
        # shape the pre-split kernel measured 2.9 ms for conv_d1 against 2.56 ms for the in-kernel split (its BK = 32 tile leaves
        # 2 workgroups per CU and both kernels are bound by the load -> LDS -> barrier chain, not by the split VALU work)
        self.presplit_operands = False
        # conv_d1 -> pool -> conv_d2 -> Hadamard . fc2 as one kernel per document tile (needs table / conv weights below 2^15)
        self.fuse_document_branch = True
        # ... with the embedding table as fp16 term planes [V][2][E rounded up to 64] (128 MB at V = 100 000, E = 300; once per weight
        # version): the fused kernel then brings a tile's token rows into LDS once, by LDS-direct loads, and splits nothing
        self.table_planes = True

    def _weights(self):
        def build():
            lm, dm = self.local_model, self.distributed_model
            t = dict(l_conv_w=lm.conv1d.weight.squeeze(2).t(),  # [DL][NF]: a matching position adds one contiguous row
                     l_conv_b=lm.conv1d.bias, l_fc1_w=lm.fc1.weight, l_fc1_b=lm.fc1.bias, l_fc2_w=lm.fc2.weight,
                     l_fc2_b=lm.fc2.bias, l_fc3_w=lm.fc3.weight, l_fc3_b=lm.fc3.bias,
                     convq_w=dm.conv_q.weight.permute(0, 2, 1), convq_b=dm.conv_q.bias,      # [NF][3][E]
                     convd1_w=dm.conv_d1.weight.permute(0, 2, 1), convd1_b=dm.conv_d1.bias,  # [NF][3][E]
                     convd2_w=dm.conv_d2.weight.squeeze(2), convd2_b=dm.conv_d2.bias,
                     fc1_w=dm.fc1.weight, fc1_b=dm.fc1.bias, fc2_w=dm.fc2.weight, fc2_b=dm.fc2.bias,
                     fc3_w=dm.fc3.weight, fc3_b=dm.fc3.bias, fc4_w=dm.fc4.weight, fc4_b=dm.fc4.bias)
            # host-side bound check, once per weight version: table and conv weights below 2^15 in magnitude -> the two big
            # convolution GEMMs may use the fp16 two-term split (their other operand is a tanh output)
            mx = max(float(self.word_embeddings.table.detach().abs().max()), float(dm.conv_d1.weight.detach().abs().max()),
                     float(dm.conv_d2.weight.detach().abs().max()), float(dm.conv_q.weight.detach().abs().max()))
            pk = lib.Packed(lib.DuetWeights, t, dict(self._dims, bounded=int(mx < 32768.0)))
            E, NF = self.word_embeddings.table.shape[1], self._dims["NF"]
            if mx < 32768.0 and self.fuse_document_branch and NF <= 320 and NF % 4 == 0 and E % 4 == 0 and self._dims["pool"] <= 5:
                # operands of the fused document-branch kernel (csrc/duet_fused.hip): conv_d1 / conv_d2 zero-padded to 320 filter rows and
                # a multiple of 32 in k, split into two fp16 terms and re-ordered into MFMA fragments, once per weight version
                def fragments(w2d, kp=None):
                    rows, k = w2d.shape
                    kp = (k + 31) // 32 * 32 if kp is None else kp
                    pad = torch.zeros(320, kp, device=w2d.device, dtype=torch.float32)
                    pad[:rows, :k] = w2d
                    planes = torch.stack(lib.split_f16x2(pad, kp))                      # [2 terms, 320, kp] int16
                    return planes.view(2, 20, 16, kp // 32, 4, 8).permute(3, 1, 0, 4, 2, 5).contiguous(), kp
                pk.keep["fw1"], k1p = fragments(pk.keep["convd1_w"].reshape(NF, 3 * E))
                pk.keep["fw2"], _ = fragments(pk.keep["convd2_w"].reshape(NF, NF), 320)   # GEMM 2 always runs its 10 k-steps
                pk.struct.fw1, pk.struct.fw2, pk.struct.K1P = pk.keep["fw1"].data_ptr(), pk.keep["fw2"].data_ptr(), k1p
                if self.table_planes:
                    EPT = (E + 63) // 64 * 64            # an even number of 32-element column chunks
                    wc = torch.zeros(NF, 3, EPT, device=pk.keep["convd1_w"].device, dtype=torch.float32)
                    wc[:, :, :E] = pk.keep["convd1_w"].reshape(NF, 3, E)
                    wc = wc.view(NF, 3, EPT // 32, 32).permute(0, 2, 1, 3).reshape(NF, 3 * EPT)       # k = (chunk, tap, element)
                    pk.keep["fw1c"], _ = fragments(wc, 3 * EPT)
                    pk.keep["ftable"] = torch.stack(lib.split_f16x2(self.word_embeddings.table, EPT), 1).contiguous()   # [V, 2, EPT]
                    pk.struct.fw1c, pk.struct.ftable, pk.struct.EPT = pk.keep["fw1c"].data_ptr(), pk.keep["ftable"].data_ptr(), EPT
            EP = (max(E, NF) + 7) // 8 * 8
            if mx < 32768.0 and self.presplit_operands and EP <= NF + 8 and EP - E < 8:
                # fp16 term planes of the table and of the two big conv weights, split once per weight version (122 MB at V = 100 000)
                planes = dict(zip(("table_h1", "table_h2"), lib.split_f16x2(self.word_embeddings.table, EP)))
                planes.update(zip(("convd1_h1", "convd1_h2"), lib.split_f16x2(pk.keep["convd1_w"].reshape(NF * 3, E), EP)))
                planes.update(zip(("convd2_h1", "convd2_h2"), lib.split_f16x2(pk.keep["convd2_w"], EP)))
                for k, v in planes.items():
                    pk.keep[k] = v
                    setattr(pk.struct, k, v.data_ptr())
                pk.struct.EP = EP
            return pk
        params = [p for n, p in self.named_parameters() if not n.startswith("word_embeddings")] + [self.word_embeddings.table]
        return self._pack.get(params, build)

    def _forward_train(self, q, d):
        """Train-mode forward (duet.py:28-59, 73-121, 147-208) on the autograd operators of autograd.py: every Linear / Conv1d runs as
        the HIP GEMM (the k=3 convolutions over three shifted row views, the k=1 ones directly), dropout and the embedding lookup on
        their HIP kernels; compare / max / Hadamard / reshapes are tensor glue."""
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        M = B * N
        lm, dm = self.local_model, self.distributed_model
        NF = self._dims["NF"]
        d2 = d.reshape(M, DL)
        # local model: exact-match matrix (PAD == PAD counts, as in the reference), Conv1d over doc positions as channels
        em = (d2.unsqueeze(1) == q.unsqueeze(1).expand(B, N, QL).reshape(M, QL).unsqueeze(2)).float()            # [M,QL,DL]
        cu = A.linear(em.reshape(M * QL, DL), lm.conv1d.weight.squeeze(2), lm.conv1d.bias, act="tanh")         # [M*QL,NF]
        cu = cu.view(M, QL, NF).transpose(1, 2).reshape(M * NF, QL)
        f1 = A.linear(cu, lm.fc1.weight, lm.fc1.bias, act="tanh").view(M, NF)
        f2 = A.dropout(A.linear(f1, lm.fc2.weight, lm.fc2.bias, act="tanh"), lm.drop.p, True)
        local = A.linear(f2, lm.fc3.weight, lm.fc3.bias, act="tanh").view(B, N)
        # distributed model
        table = self.word_embeddings.table
        eq = A.dropout(A.embed(q, table), self.emb_drop.p, True)
        ed = A.dropout(A.embed(d2, table), self.emb_drop.p, True)

        def conv3(x, conv):
            L = x.shape[1] - 2
            rows = torch.cat([x[:, k:k + L] for k in range(3)], 2).reshape(x.shape[0] * L, -1)                 # [rows, 3E] tap-major
            w = conv.weight.permute(0, 2, 1).reshape(conv.out_channels, -1)
            return A.linear(rows, w, conv.bias, act="tanh").view(x.shape[0], L, conv.out_channels)
        cq, cp = conv3(eq, dm.conv_q), conv3(ed, dm.conv_d1)
        mq = cq.max(1)[0]                                                                                       # [B,NF]
        mp = F.max_pool1d(cp.transpose(1, 2), dm.pool_size, 1).transpose(1, 2)                                  # [M,P,NF]
        P = mp.shape[1]
        qr = A.linear(mq, dm.fc1.weight, dm.fc1.bias, act="tanh")
        dr = A.linear(mp.reshape(M * P, NF), dm.conv_d2.weight.squeeze(2), dm.conv_d2.bias, act="tanh").view(M, P, NF)
        had = qr.unsqueeze(1).expand(B, N, NF).reshape(M, 1, NF) * dr
        m1 = A.linear(had.transpose(1, 2).reshape(M * NF, P), dm.fc2.weight, dm.fc2.bias, act="tanh").view(M, NF)
        m2 = A.dropout(A.linear(m1, dm.fc3.weight, dm.fc3.bias, act="tanh"), dm.dropout.p, True)
        dist = A.linear(m2, dm.fc4.weight, dm.fc4.bias, act="tanh").view(B, N)
        return local + dist

    def forward(self, batch_queries, query_len, batch_docs, doc_len, return_parts=False):
        assert batch_queries.shape[0] == batch_docs.shape[0]
        table = self.word_embeddings.table
        lib.require_device(batch_queries, batch_docs, table)
        q, d = self._clean_ids(batch_queries, batch_docs, table.shape[0])
        B, QL = q.shape
        N, DL = d.shape[1], d.shape[2]
        if QL != self.max_query_len or DL != self.max_doc_len:
            raise RuntimeError("DUET needs inputs padded to max_query_len=%d / max_doc_len=%d (force_pad), got %d / %d"
                               % (self.max_query_len, self.max_doc_len, QL, DL))
        if self.training and not return_parts:
            return self._forward_train(q, d)
        L = lib.load()
        w = self._weights()
        dev = q.device
        ws = lib.workspace(L.nir_duet_workspace_bytes(B, N, QL, DL, table.shape[1], w.ref()), dev)
        scores = torch.empty(B, N, device=dev, dtype=torch.float32)
        if B == 0 and not return_parts:
            return scores
        loc = torch.empty_like(scores) if return_parts else None
        dist = torch.empty_like(scores) if return_parts else None
        lib.check(L.nir_duet_score(lib.ptr(q), lib.ptr(d), B, N, QL, DL, lib.ptr(table), table.shape[0], table.shape[1],
                                   w.ref(), lib.ptr(ws), ws.numel(), lib.ptr(scores), lib.ptr(loc), lib.ptr(dist),
                                   lib.stream()), "nir_duet_score")
        return (scores, loc, dist) if return_parts else scores
