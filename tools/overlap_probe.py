"""Bisect harness for concurrent-vs-serial differences of captured predict graphs (X3_mnsrf 0.2068).

For every stage of a session ranker (full predict / encode only / rank_document only) capture one hipGraph per batch on `lanes`
streams, replay them (a) one at a time with a device synchronise in between, (b) all lanes in flight for `rounds` rounds, and print
max |a - b| per stage.  Usage: python tools/overlap_probe.py [MNSRF|M_MATCH_TENSOR|CARS] [--hint N] [--lanes 4] [--batches 8]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_attentive_ir_amd import lib  # noqa: E402
from context_attentive_ir_amd.config import default_args  # noqa: E402
from context_attentive_ir_amd.detinit import fill_module_  # noqa: E402
from context_attentive_ir_amd.wrappers import Multitask  # noqa: E402


def batches(n, B, S, N, QL, DL, V, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        slen = rng.integers(1, QL + 1, size=(B, S)); dlen = rng.integers(1, DL + 1, size=(B, S, N))
        srcw = rng.integers(4, V, size=(B, S, QL)); srcw[np.arange(QL)[None, None] >= slen[..., None]] = 0
        docw = rng.integers(4, V, size=(B, S, N, DL)); docw[np.arange(DL)[None, None, None] >= dlen[..., None]] = 0
        lab = np.zeros((B, S, N), np.float32); lab[..., 0] = 1
        out.append({"source_words": torch.from_numpy(srcw).cuda(), "source_lens": torch.from_numpy(slen).cuda(),
                    "document_words": torch.from_numpy(docw).cuda(), "document_lens": torch.from_numpy(dlen).cuda(),
                    "document_labels": torch.from_numpy(lab).cuda()})
    return out


def probe(name, fn, exs, lanes, rounds):
    nl = len(lanes)
    for i, ex in enumerate(exs):                       # warm: packs, workspaces
        with torch.cuda.stream(lanes[i % nl]):
            fn(ex)
    torch.cuda.synchronize()
    graphs = []
    for i, ex in enumerate(exs):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=lanes[i % nl]):
            out = fn(ex)
        graphs.append((g, out))
    torch.cuda.synchronize()
    serial = []
    for g, out in graphs:
        g.replay()
        torch.cuda.synchronize()
        serial.append(out.clone())
    worst = 0.0
    for _ in range(rounds):
        for i, (g, _) in enumerate(graphs):
            with torch.cuda.stream(lanes[i % nl]):
                g.replay()
    torch.cuda.synchronize()
    for (g, out), ref in zip(graphs, serial):
        worst = max(worst, float((out - ref).abs().max()))
    again = 0.0
    for (g, out), ref in zip(graphs, serial):
        g.replay()
        torch.cuda.synchronize()
        again = max(again, float((out - ref).abs().max()))
    print(json.dumps({"stage": name, "concurrent_vs_serial": worst, "serial_again": again}), flush=True)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="MNSRF")
    ap.add_argument("--hint", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--vocab", type=int, default=20000)
    a = ap.parse_args()
    w = Multitask(default_args(a.model, src_vocab_size=a.vocab, tgt_vocab_size=40))
    fill_module_(w.network, 23)
    w.cuda()
    w.id_check_interval = 0
    net = w.network.eval()
    exs = batches(a.batches, 16, 7, 10, 4, 64, a.vocab)
    lanes = [torch.cuda.Stream() for _ in range(a.lanes)]
    lib.set_batches_in_flight(a.hint or a.lanes, lanes)
    with torch.no_grad():
        probe("predict", lambda ex: w.predict(ex, suggest=False)["click_scores"], exs, lanes, a.rounds)
        if a.model != "CARS":
            probe("encode.mem", lambda ex: net.encode(ex["source_words"], ex["source_lens"])[0], exs, lanes, a.rounds)
            probe("encode.sess", lambda ex: net.encode(ex["source_words"], ex["source_lens"])[1], exs, lanes, a.rounds)
            probe("rank_document", lambda ex: net.rank_document(ex["source_words"], None, None, ex["document_words"], ex["document_lens"],
                                                                source_len=ex["source_lens"]), exs, lanes, a.rounds)


if __name__ == "__main__":
    main()
