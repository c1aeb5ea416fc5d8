"""GPU (-m gpu): bench.py end to end -- the single-GPU contract line, and the N = 2 strong-scaling flow (two ranks on the one
visible GPU, BENCH_BACKEND=gloo: the collectives go through the host, the sharding / gather / softmax code path is the one the
driver's 8-GPU RCCL run takes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _last_json(out):
    return json.loads(out.strip().splitlines()[-1])


def _detail(d):
    return json.load(open(os.path.join(ROOT, d["config"]["detail"])))


def test_bench_single_gpu_line():
    """the record as the driver reads it: `python bench.py --steps 20 --warmup 5`, stdout FOLLOWED by stderr -- the last line of that is ONE
    compact JSON object (< 4 KB) with the contract keys + roofline + cpu_baseline, stderr of a successful run is empty, a short region is
    repeated (`reps`) with `steps` = the argument; the full sub-records are in the detail file the line names."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--sub", "C1_esm,C4_drmm", "--cpu-seconds", "2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stderr.strip() == "", out.stderr[-2000:]
    last = (out.stdout + out.stderr).strip().splitlines()[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    assert KEYS | {"sub", "reps"} <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["reps"] >= 3 and d["value"] > 0 and d["config"]["workload"].startswith("cars")
    assert abs(d["value"] - 1120 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3         # value, ms_per_step and the C3 step (1 120 pairs) agree
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1.5 and r["kernel"] and r["avg_us"] > 0 and r["peak"] > 0 and r["unit"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["max_abs_diff_vs_gpu_softmax"] < 1e-4 and d["cpu_baseline"]["cores"] >= 1
    subs = {e["name"]: e for e in d["sub"]}
    assert subs["C1_esm"]["pairs_per_s"] > 0 and subs["C1_esm"]["bound"] == "hbm"
    assert {"hist_rows_differ", "pairs_differ", "map_delta_vs_oracle"} <= set(subs["C4_drmm"])       # the DRMM parity gap rides with its record
    full = _detail(d)
    assert full["sub"]["C1_esm"]["roofline"]["kernel"] and full["headline"]["roofline"]["kernels_us_per_step"]


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """plain `python bench.py --gpus 2` (no launcher around it): bench.py starts the two ranks itself (both on the one GPU, BENCH_BACKEND=gloo)
    and n_gpus is what the process group counted; with the RCCL backend and one visible device it refuses instead of measuring one GPU."""
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_NO_WEAK="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--sub", "none"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout + out.stderr)
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["shard_axis"] == "pair" and d["value"] > 0
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("BENCH_BACKEND")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--sub", "none"],
                             capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert out.returncode != 0 and "RCCL needs one device per rank" in out.stderr
    # a launcher world that is not --gpus is refused as well
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "12", "--sub", "none"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", BENCH_BACKEND="gloo"))
    assert out.returncode != 0 and "refusing" in out.stderr


def test_bench_two_rank_stream_record():
    """the multi-rank C5 stream as a bench record (2 ranks on the one GPU, gloo): both modes, every rank the same number of rounds."""
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_NO_WEAK="1", BENCH_H2D_SECONDS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for mode in ("batch", "pair"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C5_stream", "--streams", "2"],
                             capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, BENCH_STREAM_MODE=mode, BENCH_STREAM_SESSIONS="6000"))
        assert out.returncode == 0, out.stderr[-2000:]
        d = _last_json(out.stdout + out.stderr)
        assert d["n_gpus"] == 2 and d["value"] > 0 and ("mode '%s'" % mode) in d["config"]["parallelism"], d["config"]


@pytest.mark.parametrize("axis,needle,port", [("auto", "pair axis in 2 contiguous chunks = 8 whole sessions", "29541"),
                                              ("candidate", "session-sharded tail", "29543")])
def test_bench_two_ranks_strong_scaling_flow(axis, needle, port):
    """bench.py --gpus 2 as the driver launches it (two ranks on the one GPU, gloo): the sharded CARS headline in both shard axes, a sharded
    ranker sub-record, the labelled weak-scaling figure."""
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SHARD_AXIS=axis)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--sub", "C2_match_tensor" if axis == "auto" else "none"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["world_size"] == 2 and needle in d["config"]["parallelism"], d["config"]["parallelism"]
    assert d["config"]["shard_axis"] == ("pair" if axis == "auto" else "candidate")
    assert d["config"]["weak_scaling_pairs_per_s"] > 0
    if axis == "auto":
        sub = _detail(d)["sub"]["C2_match_tensor"]
        assert sub["world_size"] == 2 and "5 per rank" in sub["parallelism"] and sub["pairs_per_s"] > 0 and sub["shard_axis"] == "candidate"


def test_bench_rccl_lane_graphs_with_remainder_groups_replay_live_inputs():
    """The pair-axis path of an N > 1 run over RCCL (one GPU: BENCH_EMULATE_WORLD=8 over a real 1-rank RCCL group): lane graphs of KG merged steps
    plus REMAINDER groups captured on the way (warm-up 8 = 3 + 3 + 2 steps at the C5 shape).  Until round 6 the macro-batched input tensors a
    graph reads were locals of the capturing function: freed on return, overwritten -- or unmapped -- under later replays (intermittent
    "Memory access fault by GPU", deterministic for this step pattern)."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", BENCH_EMULATE_WORLD="8", BENCH_NO_H2D="1", MASTER_PORT="29551")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_BACKEND"):
        env.pop(k, None)
    for cfg, extra in (("C5_cars_bf16", ["--steps", "24", "--warmup", "8"]), ("C3_cars", ["--steps", "12", "--warmup", "6"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--sub", "none", "--no-cpu-baseline"] + extra,
                             capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, (cfg, out.returncode, out.stderr[-1500:])
        d = _last_json(out.stdout + out.stderr)
        assert d["emulated"] is True and d["emulated_world"] == 8 and d["value"] > 0 and d["config"]["shard_axis"] == "pair"


def test_two_rank_sharding_reproduces_single_rank_scores():
    """Ranker.parallelize() / Multitask.parallelize() with 2 ranks (gloo, both on the one GPU): the candidate-sharded predict
    equals the unsharded one on every rank."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "tests", "sharded_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SHARDED_OK" in out.stdout


def test_stream_device_gather_over_one_rank_rccl_group():
    """The RCCL form of the sharded stream (graph_runner.StreamingSessionPredictor, gather = "device": per lane a communication stream runs
    [ wait for the replay | all_gather_into_tensor | D2H of the gathered block ], per-slot gather buffers, the compute stream moves on) over a
    real 1-rank RCCL process group: both modes, macro 2 -- every batch delivered, equal to the plain single-stream results, identical MAP."""
    env = dict(os.environ, SHARD_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29549",
           os.path.join(ROOT, "tests", "sharded_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SHARDED_OK" in out.stdout


@pytest.mark.parametrize("axis,emulate", [("auto", "1"), ("candidate", "4"), ("pair", "8")])
def test_bench_rccl_sharded_step_is_graph_replayed(axis, emulate):
    """The RCCL path of the sharded CARS step over a real (1-rank) RCCL process group (BENCH_FORCE_DIST; BENCH_EMULATE_WORLD gives this
    process rank 0's share of a larger world): hipGraph replays around eager collectives -- pair axis: one graph + the all-gather of the
    probabilities; candidate axis: the software-pipelined segment graphs + ONE all-to-all per step."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_PORT="29553", BENCH_SHARD_AXIS=axis, BENCH_EMULATE_WORLD=emulate, BENCH_NO_H2D="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--sub", "none", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    if out.returncode < 0:       # killed by a signal (seen once in round 5: SIGABRT with an empty log, not reproduced on the next box): one retry, log kept
        print("bench.py died with signal %d; rank log tail:\n%s" % (-out.returncode, open(os.path.join(ROOT, "bench_stderr.rank0.log")).read()[-2000:]))
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stderr[-2000:], open(os.path.join(ROOT, "bench_stderr.rank0.log")).read()[-2000:])
    assert "graph capture unavailable" not in open(os.path.join(ROOT, "bench_stderr.rank0.log")).read()
    d = _last_json(out.stdout)
    assert d["scaling"] == "strong" and d["config"]["hipgraph"] is True and d["value"] > 0
    assert ("pair axis" in d["config"]["parallelism"]) == (axis != "candidate")
    assert _detail(d)["headline"]["host_enqueue_ms_per_step"] < d["ms_per_step"] * 1.05


def test_sharded_stages_reproduce_predict():
    """Multitask.shard_stage_a / shard_stage_b (the two captured halves) with the gather done by hand for two simulated ranks equal the
    unsharded predict."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from context_attentive_ir_amd import sharding, synth
    from context_attentive_ir_amd.config import default_args
    from context_attentive_ir_amd.detinit import fill_module_
    from context_attentive_ir_amd.wrappers import Multitask
    V, B, S, N = 1500, 3, 4, 7
    mt = Multitask(default_args("CARS", src_vocab_size=V, tgt_vocab_size=300))
    fill_module_(mt.network, 1013)
    mt.cuda()
    ex = {k: v.cuda() for k, v in synth.session_batch(B, S, N, 4, 20, V, seed=3, full_length=False).items()}
    ref = mt.predict(ex, suggest=False)["click_scores"]
    world, parts = 2, []
    for rank in range(world):
        d, l = sharding.shard_session_candidates(ex["document_words"], ex["document_lens"], world, rank)
        pq, pl = mt.shard_stage_a(ex, d, l)
        parts.append(pl.reshape(B * S, -1))
    gathered = torch.cat(parts, 0).contiguous()
    got = mt.shard_stage_b(pq, gathered, ex["document_labels"], N)
    assert float((got - ref).abs().max()) < 1e-6
    # sharded ranker MLP: every simulated rank scores its own slice (padded with a repeat of its last candidate), the slices are
    # concatenated rank-major like the second all-gather leaves them, shard_stage_c applies the softmax over the N real candidates
    per = parts[0].shape[1] // pq.shape[2]
    slices = []
    for rank in range(world):
        own = parts[rank].view(B, S, per, -1)
        slices.append(mt.shard_stage_b(pq, gathered, ex["document_labels"], N, own=own))
    probs = mt.shard_stage_c(torch.cat(slices, 0).contiguous(), torch.empty(B * S, N, device="cuda"), N).view(B, S, N)
    assert float((probs - ref).abs().max()) < 1e-6


def test_rank_session_candidate_slice_matches_full_scores():
    """nir_cars_rank_session_shard: scoring a slice of the pooled candidates (clicks / sessions over all of them) gives exactly the
    columns of the full score matrix -- a candidate's score depends on the session state and its own pooled vector only."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import build_model
    from context_attentive_ir_amd import synth
    V, B, S, N = 1200, 4, 5, 9
    m = build_model("CARS", vocab=V, device="cuda")
    ex = {k: v.cuda() for k, v in synth.session_batch(B, S, N, 5, 24, V, seed=8, full_length=False).items()}
    pooled, _, _ = m.encode(ex["source_words"], ex["source_lens"])
    docs = m.encode_document(ex["document_words"], ex["document_lens"])
    full = m._rank_session(pooled, docs, ex["document_labels"])[0]
    for lo, hi in ((0, 3), (3, 9), (8, 9), (0, 9)):
        part = m._rank_session(pooled, docs, ex["document_labels"], rank_docs=docs[:, :, lo:hi].contiguous())[0]
        assert part.shape == (B, S, hi - lo)
        assert torch.equal(part, full[:, :, lo:hi])
