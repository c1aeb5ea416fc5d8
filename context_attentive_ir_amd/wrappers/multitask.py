"""Multitask -- wrapper with the call shapes of neuroir.models.multitask.Multitask
(/root/reference/neuroir/models/multitask.py:24-407) for CARS, M_MATCH_TENSOR and MNSRF, ranking side only:
predict(ex) -> {'click_scores': softmax over candidates [B,S,N]} (multitask.py:262-279)."""
import torch

from .. import lib
from ..multitask import CARS, M_MATCH_TENSOR, MNSRF
from .common import WrapperBase

NETWORKS = {"CARS": CARS, "M_MATCH_TENSOR": M_MATCH_TENSOR, "MNSRF": MNSRF}


class Multitask(WrapperBase):
    def __init__(self, args, src_dict=None, tgt_dict=None, state_dict=None):
        self.args = args
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        if src_dict is not None:
            self.args.src_vocab_size = len(src_dict)
        if tgt_dict is not None:
            self.args.tgt_vocab_size = len(tgt_dict)
        self.type = args.model_type.upper()
        if self.type not in NETWORKS:
            raise RuntimeError("Unsupported model: %s (hot-path multitask models: %s)" % (args.model_type, sorted(NETWORKS)))
        self.network = NETWORKS[self.type](args)
        if state_dict:
            self.network.load_state_dict(state_dict)
        self.updates, self.use_cuda, self.parallel = 0, False, False
        self.group = None

    def parallelize(self, group=None):
        """Candidate-axis sharding of the document encoder over `group` (replaces nn.DataParallel,
        models/multitask.py:402-407)."""
        self.parallel = True
        self.group = group

    def _dev(self, t):
        return t.cuda(non_blocking=True) if self.use_cuda else t

    @torch.no_grad()
    def _rank(self, ex, want_states):
        self.network.eval()
        if self.type == "M_MATCH_TENSOR" and not want_states:
            # ranking only: the interaction head re-derives the query side from the ids inside its fused call (bitwise the values encode()
            # hands over, mmtensor.py:127-189); encode()'s other products -- the session LSTM over the max-pooled queries, the decoder's
            # initial states -- feed the suggestion side alone, so the ranking path skips it (a query BiLSTM, two GEMMs and S session steps)
            s = self.network.rank_document(self._dev(ex["source_words"]), None, None, self._dev(ex["document_words"]),
                                           self._dev(ex["document_lens"]), source_len=self._dev(ex["source_lens"]))
            return s, None, None, (None, None)
        if self.type != "CARS":     # models/multitask.py:271-278: encode -> rank_document(source, memory, session, docs, lens)
            src = self._dev(ex["source_words"])
            memory_bank, session_bank, states = self.network.encode(src, self._dev(ex["source_lens"]))
            s = self.network.rank_document(src, memory_bank, session_bank, self._dev(ex["document_words"]), self._dev(ex["document_lens"]))
            return s, states, None, (None, None)
        src_lens = self._dev(ex["source_lens"])
        join = qside = docs = None
        if self.use_cuda and not self.parallel and torch.cuda.is_current_stream_capturing() and lib.batches_in_flight() <= 1:
            # one batch in flight inside a capture (PredictGraphCache, GraphedPredictor): the query encoder (7 of 256 CUs for ~25 us at a C3
            # batch) runs on a side stream next to the document encoder and joins in front of the session tail.  (Eagerly the host is the
            # bound and the extra event calls would cost more than the overlap buys; with several batches in flight the chip is full anyway.)
            cur = torch.cuda.current_stream()
            if getattr(self, "_fork_stream", None) is None or self._fork_stream.device != cur.device:
                self._fork_stream = torch.cuda.Stream(device=cur.device)
            side = self._fork_stream
            side.wait_stream(cur)
            # the document encoder is issued FIRST: a replayed graph dispatches its nodes in creation order, and every node of the side branch in
            # front of the document recurrence delayed it by ~5 us (tools/iter_timeline.py: 22 us from the gather's end to the recurrence's start
            # with four side nodes in front, 12 with two)
            if not (self.network.no_ranker and self.network.no_document_session_encoding):
                docs = self.network.encode_document(self._dev(ex["document_words"]), self._dev(ex["document_lens"]))
            with torch.cuda.stream(side):
                pooled, encoded, _ = self.network.encode(self._dev(ex["source_words"]), src_lens)
                qside = self.network.session_query_side(pooled)      # the tail's two query-only GEMMs ride on the same branch (no extra edge)
            join = lambda: cur.wait_stream(side)             # noqa: E731
        else:
            pooled, encoded, _ = self.network.encode(self._dev(ex["source_words"]), src_lens)
        s, states, attns = self.network.rank_document(pooled, self._dev(ex["document_words"]), self._dev(ex["document_lens"]),
                                                      self._dev(ex["document_labels"]), group=self.group, shard=self.parallel,
                                                      want_states=want_states, after_documents=join, query_side=qside, encoded_docs=docs)
        return s, states, attns, (encoded, src_lens)

    # ---- the candidate-sharded ranking step in two capturable halves (bench.py / a serving loop replays each as a hipGraph and issues
    # the one all-gather between them eagerly: a collective cannot sit inside the graph, and the eager step is host-bound) ----------
    @torch.no_grad()
    def shard_stage_a(self, ex, doc_shard, len_shard):
        """queries + this rank's candidate shard -> (pooled queries [B,S,D], pooled shard [B,S,per,D])."""
        self.network.eval()
        pooled, _, _ = self.network.encode(self._dev(ex["source_words"]), self._dev(ex["source_lens"]))
        return pooled, self.network.encode_document(doc_shard, len_shard)

    @torch.no_grad()
    def shard_stage_b(self, pooled, gathered, labels, n_candidates, own=None):
        """gathered = all_gather_into_tensor of every rank's pooled shard, [world*B*S, per*D].
        own = None: softmaxed click scores [B,S,N] (every rank scores every candidate).
        own = this rank's pooled shard [B,S,per,D] (the second output of shard_stage_a): raw click scores of that slice, [B*S, per] --
        the caller all-gathers them and finishes with shard_stage_c (the ranker MLP is then sharded as well)."""
        B, S, D = pooled.shape
        world = gathered.shape[0] // (B * S)
        per = gathered.shape[1] // D
        docs = gathered.view(world, B, S, per, D).permute(1, 2, 0, 3, 4).reshape(B, S, world * per, D)[:, :, :n_candidates].contiguous()
        s = self.network._rank_session(pooled, docs, labels, want_states=False, rank_docs=own)[0].contiguous()
        if own is not None:
            return s.view(B * S, per)
        probs = torch.empty_like(s)
        lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(probs), s.shape[0] * s.shape[1], s.shape[2], lib.stream()), "nir_softmax_rows")
        return probs

    @torch.no_grad()
    def shard_stage_c(self, gathered_scores, probs, n_candidates):
        """gathered_scores = all_gather_into_tensor of every rank's [B*S, per] score slice, [world*B*S, per]; writes the softmax over
        the n_candidates of each (session, step) into probs [B*S, N] (nir_softmax_gathered) and returns it."""
        rows, n = probs.shape
        world = gathered_scores.shape[0] // rows
        lib.check(lib.load().nir_softmax_gathered(lib.ptr(gathered_scores), lib.ptr(probs), None, world, rows, gathered_scores.shape[1],
                                                  n_candidates, lib.stream()), "nir_softmax_gathered")
        return probs

    # ---- round 3: candidate-sharded encode -> all-to-all -> SESSION-sharded tail (sharding.SessionShardPlan).  Two capturable halves; the
    # caller issues plan.exchange (all_to_all_single) between them and plan.gather (all_gather_into_tensor) after, eagerly or inside the
    # same hipGraph capture.  Nothing but the KB-sized probability gather is replicated across ranks. -------------------------------------
    @torch.no_grad()
    def shard_encode(self, q_own, ql_own, doc_shard, len_shard):
        """queries of this rank's sessions [bper,S,QL] + this rank's candidate slice of every session [G*bper,S,per,DL]
        -> (pooled queries [bper,S,D], pooled candidate slice [G*bper,S,per,D])."""
        self.network.eval()
        pooled, _, _ = self.network.encode(q_own, ql_own)
        return pooled, self.network.encode_document(doc_shard, len_shard)

    @torch.no_grad()
    def tail_probs(self, pooled_q, docs, labels_own, labels_all, probs=None, labels_groups=None, click_max=None):
        """pooled queries [b,S,D] + ALL N pooled documents [b,S,N,D] of a block of sessions -> click probabilities [b,S,N] (clicks, session
        LSTMs, ranknet, softmax); the click mask's batch-wide count comes from labels_all [B,S,N] (None: the block itself), or -- blocks of
        several batches merged into one call -- per block from labels_groups [G,B,S,N]."""
        s = self.network._rank_session(pooled_q, docs, labels_own, labels_all=labels_all, labels_groups=labels_groups, click_max=click_max)[0].contiguous()
        if probs is None:
            probs = torch.empty_like(s)
        lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(probs), s.shape[0] * s.shape[1], s.shape[2], lib.stream()), "nir_softmax_rows")
        return probs

    @torch.no_grad()
    def shard_tail(self, pooled_q, recv, labels_own, labels_all, n_candidates, probs=None):
        """recv [G,bper,S,per,D] = the exchanged candidate slices of this rank's sessions (SessionShardPlan.exchange) -> click
        probabilities [bper,S,N] of those sessions."""
        G, bper, S, per, D = recv.shape
        docs = recv.permute(1, 2, 0, 3, 4).reshape(bper, S, G * per, D)[:, :, :n_candidates].contiguous()
        return self.tail_probs(pooled_q, docs, labels_own, labels_all, probs)

    @torch.no_grad()
    def predict_many(self, exs, out=None, suggest=False):
        """Ranking path of several equal-shape batches as ONE macro-batch -> click probabilities [k,B,S,N] (CARS); suggest=True: the full
        predict -> {'click_scores': [k,B,S,N], 'predictions': [k,B,S-1,max_query_len]} (greedy decode over the k*B*(S-1) rows at once).

        The batches are concatenated along the session axis: one encode launch over k x the sequences (a C3 batch alone fills 140 of 256 CUs
        with recurrence workgroups), ONE pass over the session LSTM / ranknet weights for all of them (their 76 MB of L2 traffic per tail do
        not depend on the number of sessions).  Every batch keeps the click mask's batch-wide count m of ITS OWN batch
        (cars.py:285-289 via nir_cars_click_max / labels_groups), so the result equals k separate predict() calls -- it is how a serving loop
        should feed this model when several batches are waiting.  out (optional): [k*B,S,N] result buffer."""
        if self.type in ("M_MATCH_TENSOR", "MNSRF") and not suggest:
            # every (query, candidate) row of the interaction head is independent of the rest of its batch (mmtensor.py:127-189), and so is
            # every session of MNSRF (its session LSTM runs per session, mnsrf.py:62-162; no batch-wide quantity): plain concatenation
            # along the session axis, one launch sequence over k x the rows
            self.network.eval()
            cat = lambda key: torch.cat([self._dev(e[key]) for e in exs]) if len(exs) > 1 else self._dev(exs[0][key])     # noqa: E731
            probs = self.predict_groups({k_: cat(k_) for k_ in self._FIELDS[:4]}, len(exs), out=out)
            return probs.view(len(exs), -1, *probs.shape[1:])
        if self.type != "CARS":
            raise NotImplementedError("predict_many is built for CARS (and the ranking path of M_MATCH_TENSOR / MNSRF)")
        self.network.eval()
        cat = lambda key: torch.cat([self._dev(e[key]) for e in exs]) if len(exs) > 1 else self._dev(exs[0][key])     # noqa: E731
        labels = [self._dev(e["document_labels"]) for e in exs]
        if suggest and not self.network.no_recommender:
            src_lens = cat("source_lens")
            pooled, encoded, _ = self.network.encode(cat("source_words"), src_lens)
            lab = torch.cat(labels) if len(labels) > 1 else labels[0]
            s, states, attns = self.network.rank_document(pooled, cat("document_words"), cat("document_lens"), lab, want_states=True,
                                                          labels_groups=torch.stack(labels))
            s = s.contiguous()
            probs = out if out is not None else torch.empty_like(s)
            lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(probs), s.shape[0] * s.shape[1], s.shape[2], lib.stream()), "nir_softmax_rows")
            KB, S = lab.shape[0], lab.shape[1]
            # The reference concatenates the per-step decoder states along the batch axis in STEP-major order and the decoder then reads
            # row j as (session j // (S-1), step j % (S-1)) (cars.py:431-445, 716-760): which state meets which query depends on the
            # batch size.  To give every batch of the macro-batch exactly the pairing it has on its own, the rows are regrouped from
            # (step, batch, session) to (batch, step, session) order.
            k, B0, SD = len(exs), labels[0].shape[0], S - 1
            dv = states[0].device
            idx = (torch.arange(SD, device=dv).view(1, SD, 1) * KB + torch.arange(k, device=dv).view(k, 1, 1) * B0
                   + torch.arange(B0, device=dv).view(1, 1, B0)).reshape(-1)
            states = tuple(st.index_select(1, idx) for st in states)
            dec = self.network.decode(states=states, max_len=self.args.max_query_len, src_dict=self.src_dict, tgt_dict=self.tgt_dict,
                                      batch_size=KB, session_len=S - 1, use_cuda=self.use_cuda, encoded_source=encoded, source_len=src_lens,
                                      session_attns=attns)
            return {"click_scores": probs.view(len(exs), *labels[0].shape),
                    "predictions": dec["predictions"].view(len(exs), labels[0].shape[0], S - 1, -1)}
        pooled, _, _ = self.network.encode(cat("source_words"), cat("source_lens"))
        docs = self.network.encode_document(cat("document_words"), cat("document_lens"))
        probs = self.tail_probs(pooled, docs, torch.cat(labels) if len(labels) > 1 else labels[0], None, probs=out, labels_groups=torch.stack(labels))
        return probs.view(len(exs), *labels[0].shape)

    @torch.no_grad()
    def predict_groups(self, ex, groups, out=None, click_max=None):
        """predict_many for batches that are ALREADY concatenated: ex holds groups x B sessions (block g = batch g, whole) -> click
        probabilities [groups*B,S,N]; block g uses the click count of its own B sessions (graph_runner.StreamingSessionPredictor collates
        `groups` batches into one wire block).  click_max int32 [groups] (device, optional): block g is a SLICE of its batch and takes
        the batch's click count from here (the sharded stream, sharding.StreamShardPlan mode "pair")."""
        if self.type in ("M_MATCH_TENSOR", "MNSRF") and click_max is None:
            self.network.eval()
            s = self._rank(ex, False)[0].contiguous()                  # [groups*B,S,N]: the rows do not see each other
            if out is None:
                out = torch.empty_like(s)
            lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(out), s.shape[0] * s.shape[1], s.shape[2], lib.stream()), "nir_softmax_rows")
            return out.view_as(s)
        if self.type != "CARS":
            raise NotImplementedError("predict_groups is built for CARS (and the ranking path of M_MATCH_TENSOR / MNSRF)")
        self.network.eval()
        pooled, _, _ = self.network.encode(self._dev(ex["source_words"]), self._dev(ex["source_lens"]))
        docs = self.network.encode_document(self._dev(ex["document_words"]), self._dev(ex["document_lens"]))
        labels = self._dev(ex["document_labels"])
        if labels.shape[0] % groups:
            raise RuntimeError("predict_groups: %d sessions are not %d equal batches" % (labels.shape[0], groups))
        if click_max is not None:
            return self.tail_probs(pooled, docs, labels, None, probs=out, click_max=click_max)
        lg = labels.view(groups, labels.shape[0] // groups, *labels.shape[1:])
        return self.tail_probs(pooled, docs, labels, None, probs=out, labels_groups=lg if groups > 1 else None)

    @torch.no_grad()
    def predict_sharded(self, ex, plan, group=None):
        """Eager form of the whole sharded ranking step -> click probabilities [B,S,N] on every rank."""
        from .. import sharding
        self.network.eval()
        dex = {k: self._dev(v) for k, v in ex.items() if torch.is_tensor(v)}
        return sharding.session_sharded_click_probs(plan, lambda q, l: self.network.encode(q, l)[0], self.network.encode_document,
                                                    self.tail_probs, dex, group)

    @torch.no_grad()
    def scores(self, ex):
        """raw click scores [B,S,N] (ranking path only)."""
        return self._rank(ex, False)[0]

    _FIELDS = ("source_words", "source_lens", "document_words", "document_lens", "document_labels")

    @torch.no_grad()
    def predict(self, ex, suggest=True):
        """models/multitask.py:229-317: {'click_scores': softmax over candidates [B,S,N], 'predictions': LongTensor
        [B,S-1,max_query_len] (suggest=True; None for CARS with the recommender off)}.  For a batch in the reference's collate layout (`ids`,
        `source_tokens`, `target_tokens`: what its DataLoader yields) the dict is the reference's: `predictions` = the decoded strings, plus `ex_ids`,
        `targets`, `src_sequences` (and the ids under `prediction_ids`): `validate_official` of main/multitask.py runs unchanged.
        suggest=False = the ranking path only (what bench.py times as a step and GraphedPredictor captures).
        From the second call of a batch shape on, the call replays a captured hipGraph (WrapperBase._graph_entry; decode included); an
        out-of-vocabulary id raises IndexError from the scores' `.cpu()` or from the next call (WrapperBase.id_check)."""
        self._poll_ids()
        do_decode = bool(suggest) and not (self.type == "CARS" and self.network.no_recommender)
        fields = self._FIELDS if self.type == "CARS" else self._FIELDS[:4]
        if do_decode or self.type == "CARS" and self.network.no_ranker:
            out = self._graphed(ex, fields, do_decode, lambda e: self._predict_body(e, do_decode))
            if out is not None and self.id_check == "blocking":
                self._maybe_check_ids()
        else:
            # ranking only: the captured part ends at the raw scores; the softmax (+ the publication of the error word) runs eagerly into a
            # fresh tensor -- one launch instead of an in-graph softmax and a copy of the static output
            out = self._graphed(ex, fields, do_decode, lambda e: self._rank(e, False)[0].contiguous(), self._finish_scores)
        if out is None:
            out = self._predict_body(ex, do_decode)
        out["click_scores"] = self._checked(out["click_scores"])
        if do_decode and out.get("predictions") is not None and "ids" in ex and ex.get("source_tokens") is not None:
            out.update(self._suggestion_text(ex, out["predictions"]))
        return out

    def _suggestion_text(self, ex, pred_ids):
        """The host-side tail of the reference's predict (models/multitask.py:294-316) for a batch in the reference's collate layout (`ids`,
        `source_tokens`, `target_tokens`, `session_len`, `batch_size`: inputters/multitask/vector.py:60-149): decoded suggestions as strings
        (`tens2sen`, utils/misc.py:36-62: BOS skipped, cut at EOS, words through `tgt_dict`, an empty sentence becomes str(PAD)), step-major like the
        reference's `extend` loop, next to `ex_ids`, `targets` and `src_sequences` -- what `validate_official` (main/multitask.py:288-300) reads.
        The token ids stay available as `prediction_ids` [B, S-1, max_query_len].  This is the one place where the full predict synchronises
        (the reference's `wt.item()` per token does too)."""
        from ..constants import BOS, EOS, PAD
        S = int(ex["session_len"]) if "session_len" in ex else int(pred_ids.shape[1]) + 1
        B = int(ex.get("batch_size", pred_ids.shape[0]))
        host = pred_ids.cpu().tolist()                                      # [B][S-1][max_len]
        self._poll_ids()                                                    # (synchronised: the pinned error word of this call is final)
        words, nw = self.tgt_dict, (len(self.tgt_dict) if self.tgt_dict is not None else 0)
        preds, targets, srcs = [], [], []
        for sidx in range(S - 1):
            for bidx in range(len(host)):
                sent = []
                for wd in host[bidx][sidx]:
                    if wd == BOS:
                        continue
                    if wd == EOS:
                        break
                    sent.append(words[wd] if (words is not None and wd < nw) else str(wd))
                preds.append(" ".join(sent) if sent else str(PAD))
            for bidx in range(B):
                tokens = ex["target_tokens"][bidx][sidx]
                targets.append([" ".join(tokens[1:-1])])
                srcs.append(" ".join(" ".join(q[1:-1]) for q in ex["source_tokens"][bidx][0:sidx + 1]))
        return {"prediction_ids": pred_ids, "predictions": preds, "targets": targets, "src_sequences": srcs,
                "ex_ids": [_id + str(i) for i in range(S) for _id in ex["ids"]]}

    def _finish_scores(self, s):
        probs, published = self._softmax_rows(s)
        self._maybe_check_ids(published)
        return {"click_scores": probs, "predictions": None}

    def _predict_body(self, ex, do_decode):
        s, states, attns, enc = self._rank(ex, do_decode)
        out = {"click_scores": None, "predictions": None}
        published = False
        if torch.is_tensor(s):
            s = s.contiguous()
            if do_decode and states is not None:             # (the decoder runs behind the softmax: the error word is published after it)
                probs = torch.empty_like(s)
                lib.check(lib.load().nir_softmax_rows(lib.ptr(s), lib.ptr(probs), s.shape[0] * s.shape[1], s.shape[2], lib.stream()),
                          "nir_softmax_rows")
            else:
                probs, published = self._softmax_rows(s)
            out["click_scores"] = probs
        if do_decode and states is not None:
            B, S = ex["source_words"].shape[0], ex["source_words"].shape[1]
            dec = self.network.decode(states=states, max_len=self.args.max_query_len, src_dict=self.src_dict, tgt_dict=self.tgt_dict,
                                      batch_size=B, session_len=S - 1, use_cuda=self.use_cuda, encoded_source=enc[0], source_len=enc[1],
                                      session_attns=attns)
            out["predictions"] = dec["predictions"]
        self._maybe_check_ids(published)
        return out

    def update(self, ex):
        """models/multitask.py:161-223: train-mode forward -> (1 - alpha) ranking + alpha suggestion (+ regularisation) -> backward
        -> clip_grad_norm(grad_clipping) -> optimizer step (CARS, M_MATCH_TENSOR, MNSRF)."""
        if self.optimizer is None:
            raise RuntimeError("No optimizer set.")
        self._poll_ids()
        self.optimizer.zero_grad()
        loss = self._update_body(ex)
        self.updates += 1
        self._maybe_check_ids()
        return loss

    def _update_body(self, ex):
        """forward -> losses -> backward -> (gradient averaging) -> clipping -> optimizer step; no host synchronisation, so that
        common.GraphedUpdate can capture it into one hipGraph (gradients must be cleared by the caller)."""
        from .. import autograd as A
        self.network.train()
        A.STEP.begin()                         # in-place accumulation of the parameter gradients of A.linear, transposes once per step
        try:
            g = lambda k: self._dev(ex[k])        # noqa: E731
            loss = self.network(source_rep=g("source_words"), source_len=g("source_lens"), target_rep=g("target_words"),
                                target_len=g("target_lens"), target_seq=g("target_seq"), document_rep=g("document_words"),
                                document_len=g("document_lens"), document_label=g("document_labels"))
            total = (1 - self.args.alpha) * loss["ranking_loss"] + self.args.alpha * loss["suggestion_loss"]
            if loss.get("regularization") is not None:
                total = total + loss["regularization"]
            loss["total_loss"] = total
            total.backward()
        except BaseException:
            A.STEP.abort()
            raise
        A.STEP.end()
        self.sync_gradients()                 # multi-rank: average the gradients of all ranks (WrapperBase.sync_gradients)
        torch.nn.utils.clip_grad_norm_(self.network.parameters(), self.args.grad_clipping)
        self.optimizer.step()
        return loss

    @staticmethod
    def load(filename, new_args=None):
        saved = torch.load(filename, map_location="cpu", weights_only=False)
        args = saved["args"]
        if new_args is not None:
            from ..config import override_model_args
            args = override_model_args(args, new_args)
        return Multitask(args, saved.get("src_dict"), saved.get("tgt_dict"), saved["state_dict"])

    @staticmethod
    def load_checkpoint(filename, use_gpu=True):
        saved = torch.load(filename, map_location="cpu", weights_only=False)
        model = Multitask(saved["args"], saved.get("src_dict"), saved.get("tgt_dict"), saved["state_dict"])
        if use_gpu:
            model.cuda()
        model.init_optimizer(saved["optimizer"], use_gpu)
        return model, saved["epoch"]
