#!/usr/bin/env python
"""bench.py -- ranked (query,doc) pairs/sec of the encode-and-rank hot path on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]        (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], the config the metric is quoted on that fits one GPU):
    match_tensor ranker, batch = 32 queries x 10 candidates, q_len 4, doc_len 64, emb_dim 300, fp32,
    synthetic MSMARCO-shaped ids (Zipf over a 100 000 x 300 table, seed 1013), full-length sequences.
A "step" = Ranker.predict on one batch already resident in HBM: network forward + softmax over candidates.
N > 1 (weak scaling): the CANDIDATE axis is sharded -- every rank scores its own 10 candidates of each of the
32 queries (global candidate set = 10*N per query) and one RCCL all-gather assembles the [32, 10*N] score
matrix on every rank before the softmax.  value = pairs all ranks ranked / max-rank time.

Besides the driver's contract keys the JSON line carries
    roofline     -- the dominant kernel (by summed duration), timed with HIP events on its launch stream in a
                    profiled pass over the same workload right after the timed region (events inside the timed
                    region would distort it); its algorithmic flops/bytes per launch are stated in DESIGN.md;
    cpu_baseline -- the CPU oracle (oracle/neuroir_cpu.py, pinned to the reference) timed on the host cores
                    on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With 4 batches in
# flight plus torch's own streams that leaves no spare queue and lanes serialise behind each other (measured: 2.35 M
# pairs/s at 4 queues, 3.13 M at 8).  Must be set before the HIP runtime initialises, i.e. before `import torch`.
# (With a single lane the default of 4 is kept: 8 queues measured slower there.)
if not ("--streams" in sys.argv and sys.argv[sys.argv.index("--streams") + 1:][:1] == ["1"]):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from context_attentive_ir_amd import lib, sharding, synth  # noqa: E402
from context_attentive_ir_amd.config import default_args  # noqa: E402
from context_attentive_ir_amd.detinit import fill_module_  # noqa: E402
from context_attentive_ir_amd.wrappers import Multitask, Ranker  # noqa: E402

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_FP32_TFLOPS = 157.3     # fp32 vector == fp32 MFMA peak


def algorithmic_bytes_per_pair(N, QL, DL, E=300):
    """every token occurrence gathers its fp32 row once + int64 id, + 4 B score."""
    return DL * (4 * E + 8) + QL * (4 * E + 8) / N + 4


# per-kernel algorithmic work for one launch of the MatchTensor pipeline (derivations in DESIGN.md section 5)
def kernel_work(name, B, N, QL, DL, E=300, F=40, Hq=15, Hd=70, C=50):
    M = B * N
    w = {
        # BiLSTM recurrence with the input projection fused in: 2 dirs x DL steps x 4H x (H + F) MACs per sequence;
        # HBM: x [DL,F] read once per direction, h [DL,2H] written once
        "lstm_mfma_kernel<5,28,3,1>": dict(flops=M * 2 * DL * 2 * 4 * Hd * (Hd + F), bytes=M * DL * (2 * F + 2 * Hd) * 4),
        "lstm_mfma_kernel<5,28,4,2>": dict(flops=M * 2 * DL * 2 * 4 * Hd * (Hd + F), bytes=M * DL * (2 * F + 2 * Hd) * 4),
        "lstm_mfma_kernel<1,16,4,1>": dict(flops=B * 2 * QL * 2 * 4 * Hq * (Hq + F), bytes=B * QL * (2 * F + 2 * Hq) * 4),
        "lstm_rec_kernel[fused]<80>": dict(flops=M * 2 * DL * 2 * 4 * Hd * (Hd + F), bytes=M * DL * (2 * F + 2 * Hd) * 4),
        "lstm_rec_kernel[fused]<16>": dict(flops=B * 2 * QL * 2 * 4 * Hq * (Hq + F), bytes=B * QL * (2 * F + 2 * Hq) * 4),
        # interaction GEMM after folding the query taps: per (i,j) position 15 taps x C channels x 6 filters MACs,
        # + 18->20 1x1 conv; HBM: Pd [DL,C] + ids read once per pair (U is per query, L2-resident)
        "mt_head_kernel": dict(flops=M * QL * DL * 2 * (15 * C * 6 + 45 * 6 + 18 * 20), bytes=M * DL * (C * 4 + 8) + M * 4),
        "gemm_kernel[gather]": dict(flops=(M * DL + B * QL) * 2 * E * F, bytes=(M * DL + B * QL) * (4 * E + 8 + 4 * F)),
        "gemm_kernel": dict(flops=M * DL * 2 * (2 * Hd * C) + B * QL * 2 * (2 * Hq * C), bytes=M * DL * 4 * (2 * Hd + C)),
        "mt_fold_kernel": dict(flops=B * QL * 15 * C * 3 * 6 * 2, bytes=B * QL * 15 * C * 8 * 4),
    }
    return w.get(name)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--model", default="match_tensor", choices=["match_tensor", "esm", "drmm", "duet", "cars", "m_match_tensor", "mnsrf"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cands", type=int, default=10)
    ap.add_argument("--qlen", type=int, default=4)
    ap.add_argument("--dlen", type=int, default=64)
    ap.add_argument("--session", type=int, default=7)
    ap.add_argument("--vocab", type=int, default=100000)
    ap.add_argument("--uniform", action="store_true", help="uniform token ids (worst case for the gather) instead of Zipf")
    ap.add_argument("--nbatches", type=int, default=12, help="distinct resident batches cycled through")
    ap.add_argument("--streams", type=int, default=4, help="batches in flight: step i runs on HIP stream i %% streams")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a captured hipGraph")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="CARS: bf16 = bf16 folded tables + bf16 MFMA recurrence (BASELINE config 5)")
    ap.add_argument("--no-fold", action="store_true", help="CARS: per-batch gather-GEMM instead of the folded embedding table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def build(args, dev):
    kind = args.model.upper()
    extra = dict(max_query_len=args.qlen, max_doc_len=args.dlen) if kind == "DUET" else {}
    margs = default_args(kind, src_vocab_size=args.vocab, **extra)
    wrapper = Multitask(margs) if kind in ("CARS", "M_MATCH_TENSOR", "MNSRF") else Ranker(margs)
    fill_module_(wrapper.network, 1013)
    if kind == "CARS":
        wrapper.network.compute_dtype = args.dtype
        wrapper.network.fold_embeddings = not args.no_fold
    wrapper.cuda()
    wrapper.network.eval()
    return wrapper


def make_batches(args, rank, dev):
    out = []
    for i in range(args.nbatches):
        seed = 1013 + 7919 * i + 104729 * rank
        if args.model in ("cars", "m_match_tensor", "mnsrf"):
            b = synth.session_batch(args.batch, args.session, args.cands, args.qlen, args.dlen, args.vocab, seed)
        else:
            b = synth.ranker_batch(args.batch, args.cands, args.qlen, args.dlen, args.vocab, seed, uniform=args.uniform)
        out.append({k: v.to(dev) for k, v in b.items()})
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()   # (smoke-testing N ranks on a 1-GPU box maps them all to device 0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # "gloo": flow test without RCCL (gathers through the host)
    dist = None
    multi = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))   # BENCH_FORCE_DIST: 1-rank group, exercises the RCCL path
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    L = lib.load()
    model = build(args, dev)
    batches = make_batches(args, rank, dev)
    is_cars = args.model in ("cars", "m_match_tensor", "mnsrf")      # session-structured batch [B,S,N,DL]
    pairs_per_step_rank = args.batch * args.cands * (args.session if is_cars else 1)

    def forward(i):
        """rank-local part of a step: network forward on this rank's candidate shard (captured into a hipGraph)."""
        ex = batches[i % len(batches)]
        if is_cars:
            return model.predict(ex)["click_scores"]
        s = model.scores(ex)
        if multi:
            return s
        out = torch.empty_like(s)
        lib.check(L.nir_softmax_rows(lib.ptr(s), lib.ptr(out), s.shape[0], s.shape[1], lib.stream()), "softmax")
        return out

    def finish(s):
        """cross-rank part (eager): one all-gather of the score shards, then the softmax over all candidates."""
        if not multi or is_cars:
            return s
        if backend == "nccl":
            # blocking gather into this lane's persistent buffer (the lane waits, the other lanes keep the GPU busy; a
            # blocking collective also keeps torch's allocator free of cross-stream bookkeeping), then ONE kernel that
            # does the softmax straight off the rank-major gather buffer
            ln = lanes.index(torch.cuda.current_stream()) if torch.cuda.current_stream() in lanes else 0
            if fbufs[ln] is None:
                fbufs[ln] = (torch.empty(world * s.shape[0], s.shape[1], device=dev),
                             torch.empty(s.shape[0], args.cands * world, device=dev))
            gbuf, probs = fbufs[ln]
            dist.all_gather_into_tensor(gbuf, s.contiguous())
            lib.check(L.nir_softmax_gathered(lib.ptr(gbuf), lib.ptr(probs), None, world, s.shape[0], s.shape[1],
                                             args.cands * world, lib.stream()), "nir_softmax_gathered")
            return probs
        s = sharding.gather_scores(s.cpu(), args.cands * world).to(dev)
        out = torch.empty_like(s)
        lib.check(L.nir_softmax_rows(lib.ptr(s), lib.ptr(out), s.shape[0], s.shape[1], lib.stream()), "softmax")
        return out

    fbufs = []          # per-lane (gather buffer, probabilities), filled once the lanes exist

    def step(i):
        return finish(forward(i))

    # ---- hipGraph replay of the rank-local forward (removes host launch overhead; the collective stays eager) ----
    # Steps are independent batches, so `--streams` of them are kept in flight: step i runs on HIP stream (lane)
    # i % streams.  One batch of 320 pairs cannot fill 256 CUs (the recurrence occupies 214 CUs with 4 waves each), a
    # second batch's kernels co-run in the idle slots.  Every lane has its own workspace (lib.workspace keys on the
    # stream) and its own captured graphs, so lanes share only read-only weights.
    graphs = None
    use_graph = not args.no_graph
    lanes = [torch.cuda.Stream() for _ in range(max(1, args.streams))]
    fbufs.extend([None] * len(lanes))
    # tell the library: with several batches in flight it drops its own query/document fork inside a batch (the side
    # branch made the lanes' graphs compete for hardware queues: 2 lanes 1.66 M pairs/s with, 2.11 M without) and packs
    # the recurrence into fuller workgroups (3.69 M vs 3.36 M at 4 lanes)
    L.nir_set_batches_in_flight(len(lanes))
    lane_of = lambda i: (i % len(batches)) % len(lanes)   # noqa: E731  (a batch/graph always runs on the same lane)
    torch.cuda.set_stream(lanes[0])
    for i in range(max(3, min(args.warmup, 5)) * len(lanes)):
        with torch.cuda.stream(lanes[lane_of(i)]):
            step(i)
    torch.cuda.synchronize()
    if use_graph:
        try:
            graphs = []
            for i in range(len(batches)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=lanes[lane_of(i)]):
                    out = forward(i)
                graphs.append((g, out))
        except Exception as e:  # pragma: no cover - graph capture is an optimisation only
            print("[bench] graph capture unavailable (%s); timing eager launches" % e, file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()

    # N > 1 with ONE batch in flight: the score all-gather of step k is issued asynchronously (RCCL's own stream) and
    # consumed one step later, so it overlaps step k+1's scoring kernels; `drain()` completes the last step inside the
    # timed region.  With several lanes the other lanes already cover a lane's gather latency, and chaining every lane's
    # deferred wait through the single RCCL stream measured slower (1-rank RCCL group: 0.122 vs 0.100 ms/step), so the
    # lane simply waits for its own gather.  BENCH_ASYNC_GATHER=1 / BENCH_SYNC_GATHER=1 force either behaviour.
    want_async = len(lanes) == 1 or bool(os.environ.get("BENCH_ASYNC_GATHER"))
    pipelined = [multi and not is_cars and backend == "nccl" and want_async and not os.environ.get("BENCH_SYNC_GATHER")]
    pending = [[] for _ in lanes]

    # caller-owned gather / probability buffers, two per lane (a lane has at most one gather outstanding)
    gbufs = [[None, None] for _ in lanes]
    pbufs = [[None, None] for _ in lanes]
    gparity = [0 for _ in lanes]

    def drain(lane=None):
        for ln in range(len(lanes)) if lane is None else [lane]:
            with torch.cuda.stream(lanes[ln]):
                while pending[ln]:
                    h, par = pending[ln].pop(0)
                    if pbufs[ln][par] is None:
                        pbufs[ln][par] = torch.empty(h.B, h.N, device=dev)
                    h.softmax(pbufs[ln][par])

    def run(i, only_lane=None):
        ln = lane_of(i) if only_lane is None else only_lane
        with torch.cuda.stream(lanes[ln]):
            if graphs is not None:
                g, out = graphs[i % len(graphs)]
                g.replay()
            else:
                out = forward(i)
            if pipelined[0]:
                try:
                    par = gparity[ln]
                    gparity[ln] ^= 1
                    if gbufs[ln][par] is None:
                        gbufs[ln][par] = torch.empty(world * out.shape[0], out.shape[1], device=dev)
                    h = sharding.ScoreGather(out, args.cands * world, out=gbufs[ln][par])
                except Exception as e:  # pragma: no cover - fall back to the blocking gather
                    print("[bench] async all-gather unavailable (%s); using the blocking gather" % e, file=sys.stderr)
                    pipelined[0] = False
                    finish(out)
                    return
                drain(ln)
                pending[ln].append((h, par))
            else:
                finish(out)

    for i in range(args.warmup):
        run(i)
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(i)
    drain()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3     # host enqueue time per step (diagnostic)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_pairs = pairs_per_step_rank * world * args.steps
    value = total_pairs / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # for reference: the same steps strictly one after another on a single stream (per-batch latency), and a race check:
    # the scores the concurrent replays left in the graphs' output buffers must equal a serial replay's bit for bit
    single_ms = ms_per_step
    overlap_diff = None
    if len(lanes) > 1:
        if graphs is not None and not pipelined[0]:
            conc = [out.clone() for _, out in graphs]
            torch.cuda.synchronize()
            overlap_diff = 0.0
            for j, (g, out) in enumerate(graphs):
                g.replay()
                torch.cuda.synchronize()
                overlap_diff = max(overlap_diff, float((out - conc[j]).abs().max()))
        ns = max(10, min(args.steps, 100))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(ns):
            run(i, only_lane=0)
        drain()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - ts) / ns * 1e3
        if dist:
            dist.barrier()

    # ---- second number (never `value`): ids start in pinned HOST memory; every step copies them H2D into the static
    # buffers of a captured hipGraph (context_attentive_ir_amd/graph_runner.py) and replays it
    h2d_value = None
    if world == 1:
        try:
            from context_attentive_ir_amd.graph_runner import GraphedPredictor
            gps = [GraphedPredictor(model, batches[0], queue_ahead=len(lanes) == 1) for _ in lanes]   # one predictor (stream + static inputs) per lane
            # every batch packed into one pinned host buffer (what inputters.*_batchify produces): ONE H2D copy per step
            host = [gps[0].pack({k: v.cpu() for k, v in b.items()}) for b in batches]
            nh = max(10, min(args.steps, 400))
            for i in range(3 * len(gps)):
                gps[i % len(gps)].predict(host[i % len(host)], clone=False)
            torch.cuda.synchronize()
            th = time.perf_counter()
            for i in range(nh):
                gps[i % len(gps)].predict(host[i % len(host)], clone=False)
            torch.cuda.synchronize()
            h2d_value = pairs_per_step_rank * nh / (time.perf_counter() - th)
        except Exception as e:  # pragma: no cover - secondary figure only
            print("[bench] H2D-inclusive figure unavailable: %s" % e, file=sys.stderr)

    # ---- profiled pass: HIP events around every kernel of the library, same workload -------------------
    roofline = None
    import ctypes
    nprof = max(10, min(args.steps, 50))
    os.environ["NIR_NO_FORK"] = "1"   # time every kernel in isolation (no query/document stream overlap)
    if rank == 0:
        L.nir_profile_enable(1)
    for i in range(nprof):            # every rank runs the steps (the all-gather is collective); rank 0 records
        step(i)
    torch.cuda.synchronize()
    L.nir_profile_enable(0)
    os.environ.pop("NIR_NO_FORK", None)
    if rank == 0:
        buf = ctypes.create_string_buffer(1 << 16)
        L.nir_profile_report(buf, len(buf))
        kern = {}
        for line in buf.value.decode().strip().splitlines():
            name, cnt, ms = line.rsplit(",", 2)
            kern[name] = (int(cnt), float(ms))
        if kern:
            dom = max(kern, key=lambda k: kern[k][1])
            cnt, ms = kern[dom]
            launches_per_step = cnt / nprof
            avg_us = ms / cnt * 1e3
            roofline = {"kernel": dom, "avg_us": round(avg_us, 3), "launches_per_step": launches_per_step,
                        "kernels_us_per_step": {k: round(v[1] / nprof * 1e3, 2) for k, v in sorted(kern.items())}}
            work = kernel_work(dom, args.batch, args.cands, args.qlen, args.dlen) if args.model == "match_tensor" else None
            if work:
                fl = work["flops"] / launches_per_step
                by = work["bytes"] / launches_per_step
                tf = fl / (avg_us * 1e-6) / 1e12
                gbs = by / (avg_us * 1e-6) / 1e9
                if fl / by > PEAK_FP32_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
                    roofline.update(bound="mfma", achieved=round(tf, 4), peak=PEAK_FP32_TFLOPS, unit="TFLOP/s",
                                    frac=round(tf / PEAK_FP32_TFLOPS, 5))
                else:
                    roofline.update(bound="hbm", achieved=round(gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                    frac=round(gbs / PEAK_HBM_GBS, 5))
                roofline["alg_flops_per_launch"] = fl
                roofline["alg_bytes_per_launch"] = by
            elif args.model in ("esm", "drmm"):
                by = algorithmic_bytes_per_pair(args.cands, args.qlen, args.dlen) * args.batch * args.cands
                gbs = by / (avg_us * 1e-6) / 1e9
                roofline.update(bound="hbm", achieved=round(gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                                frac=round(gbs / PEAK_HBM_GBS, 5), alg_bytes_per_launch=by)
                if gbs > PEAK_HBM_GBS:
                    roofline["note"] = ("algorithmic bytes/s above the HBM peak: repeated (Zipf) ids are served from L2/MALL, "
                                        "not HBM; run with --uniform --vocab 2000000 for the HBM-resident figure")
            # HBM bytes per launch from the committed PMC capture of this same workload (profiles/traffic.json:
            # separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 FETCH correction); null when no capture matches
            roofline["traffic"] = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                default_wl = (args.model, args.batch, args.cands, args.qlen, args.dlen) == ("match_tensor", 32, 10, 4, 64)
                if default_wl and dom in tj["kernels"]:
                    roofline["traffic"] = tj["kernels"][dom]["bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                pass
            # whole-step HBM fraction BASELINE.json asks for (algorithmic bytes of SURVEY.md 8d x pairs/s)
            bpp = algorithmic_bytes_per_pair(args.cands, args.qlen, args.dlen)
            roofline["step_hbm_GBps"] = round(value / world * bpp / 1e9, 2)
            roofline["step_hbm_frac"] = round(value / world * bpp / 1e9 / PEAK_HBM_GBS, 5)
            # ... and "FLOP/s / peak next to it" (SURVEY.md 8d): algorithmic flops per pair of the reference's op
            # sequence at the named config shapes (MatchTensor C2 1.76e7, CARS C3 6.72e7, DUET C4 2.08e8, DRMM C4 8.7e5)
            named = {("match_tensor", 4, 64): 1.76e7, ("m_match_tensor", 4, 64): 1.76e7, ("cars", 4, 64): 6.72e7,
                     ("duet", 4, 290): 2.08e8, ("drmm", 4, 290): 8.7e5}
            fpp = named.get((args.model, args.qlen, args.dlen))
            if fpp:
                roofline["step_alg_TFLOPs"] = round(value / world * fpp / 1e12, 2)
                roofline["step_flop_frac"] = round(value / world * fpp / 1e12 / PEAK_FP32_TFLOPS, 5)

    # ---- CPU baseline: the oracle (pinned port of the reference) on the host cores ---------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import neuroir_cpu as O
        sd = {k: v.detach().cpu().float() for k, v in model.network.state_dict().items()}
        ex = {k: v.cpu() for k, v in batches[0].items()}
        ncores = torch.get_num_threads()
        if args.model == "mnsrf":
            fn = lambda: torch.softmax(O.mnsrf_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"]), -1)  # noqa: E731
        elif args.model == "m_match_tensor":
            fn = lambda: torch.softmax(O.m_match_tensor_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"]), -1)  # noqa: E731
        elif is_cars:
            fn = lambda: O.predict_softmax(O.cars_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"], ex["document_labels"]))  # noqa: E731
        else:
            f = O.MODEL_FNS[args.model.upper()]
            fn = lambda: O.predict_softmax(f(sd, ex["que_rep"], ex["que_len"], ex["doc_rep"], ex["doc_len"]))  # noqa: E731
        ref = fn()
        gpu = step(0).cpu()
        maxdiff = float((gpu - ref.view_as(gpu)).abs().max())
        # be fair to the CPU: tiny per-op tensors oversubscribe a big host, so probe a few thread counts first
        avail = os.cpu_count() or ncores
        best_t, best_rate = ncores, 0.0
        for t in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(t)
            fn()
            n0, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 0.75:
                fn(); n0 += 1
            rate = n0 / (time.perf_counter() - t0)
            if rate > best_rate:
                best_t, best_rate = t, rate
        torch.set_num_threads(best_t)
        n, t1 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t1
            if dt > args.cpu_seconds or n >= 5000:
                break
        cpu = {"value": round(n * pairs_per_step_rank / dt, 1), "unit": "pairs/s", "cores": best_t, "kind": "port",
               "sample": "%d batches of the same %s workload in %.1f s (oracle/neuroir_cpu.py = pinned port of the reference, "
                         "torch %s CPU, best of {8,16,32,64} threads = %d; host has %d logical cores)"
                         % (n, args.model, dt, torch.__version__, best_t, avail),
               "max_abs_diff_vs_gpu_softmax": maxdiff}

    if rank == 0:
        cfg = {"workload": "%s ranker, batch=%d queries x %d candidates%s, q_len=%d, doc_len=%d, emb_dim=300, vocab=%d, fp32, full-length %s ids"
                           % (args.model, args.batch * (world if is_cars else 1), args.cands * (1 if is_cars else world), (" x session %d" % args.session) if is_cars else "",
                              args.qlen, args.dlen, args.vocab, "uniform" if args.uniform else "Zipf"),
               "global_batch_pairs": pairs_per_step_rank * world,
               "parallelism": "single GPU" if world == 1 else
                              ("x%d independent per-rank session batches, no collective (sharded CARS = Multitask.parallelize)" % world)
                              if is_cars else ("candidate-sharded x%d + RCCL all-gather of scores%s"
                                               % (world, " (async, consumed one step later)" if pipelined[0] else "")),
               "hipgraph": graphs is not None,
               "batches_in_flight": len(lanes),
               "host_enqueue_ms_per_step": round(host_ms, 5),
               "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
               "ms_per_step_one_batch_in_flight": round(single_ms, 5),
               "overlapped_vs_serial_max_abs_diff": overlap_diff,
               "pairs_per_s_with_host_ids_h2d": None if h2d_value is None else round(h2d_value, 1)}
        line = {"metric": "ranked (query,doc) pairs/sec", "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg,
                "roofline": roofline, "cpu_baseline": cpu}
        result_line = json.dumps(line)
    else:
        result_line = None
    if dist:
        dist.destroy_process_group()
    # RCCL writes its version banner through C stdio (buffered when stdout is a pipe): flush it out first so that the
    # JSON line is the LAST line of rank 0's stdout
    import ctypes as _ct
    _ct.CDLL(None).fflush(None)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
