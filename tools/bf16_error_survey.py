"""Achieved error of the bf16 CARS path (BASELINE config 5) against the ORACLE over the shapes the tests use: max |score diff|, max |softmax
prob diff|, MAP / MAP@10 deltas.  The bounds in tests/test_gpu_fold.py (BF16_SCORE_TOL, BF16_PROB_TOL) are set to <= 2x these maxima.
    python tools/bf16_error_survey.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_model, cpu_state_dict  # noqa: E402
from oracle import neuroir_cpu as O  # noqa: E402
from context_attentive_ir_amd import synth  # noqa: E402
from context_attentive_ir_amd.eval import ltorank  # noqa: E402


def main():
    out = []
    for (B, S, N, QL, DL, V, seed) in [(4, 7, 10, 6, 64, 3000, 11), (2, 3, 50, 6, 64, 3000, 5), (3, 4, 9, 4, 64, 3000, 11), (16, 7, 10, 4, 64, 5000, 8),
                                       (4, 7, 50, 4, 64, 100000, 2), (8, 5, 10, 4, 64, 100000, 3), (6, 7, 10, 4, 64, 20000, 4)]:
        m = build_model("CARS", vocab=V, device="cuda")
        ex = synth.session_batch(B, S, N, QL, DL, V, seed=seed, full_length=False)
        sd = cpu_state_dict(m)
        ref = O.cars_scores(sd, ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"], ex["document_labels"])
        dex = {k: v.cuda() for k, v in ex.items()}
        rec = {"shape": [B, S, N, QL, DL], "vocab": V}
        for dt in ("f32", "bf16"):
            m.compute_dtype = dt
            pooled, _, _ = m.encode(dex["source_words"], dex["source_lens"])
            s = m.rank_document(pooled, dex["document_words"], dex["document_lens"], dex["document_labels"])[0].cpu()
            lab = ex["document_labels"].reshape(-1, N).numpy()
            r_ref, r_got = ref.reshape(-1, N).numpy(), s.reshape(-1, N).numpy()
            a_ref, a_got = np.argsort(-r_ref, 1, kind="stable"), np.argsort(-r_got, 1, kind="stable")
            rec[dt] = {"score_max_abs_diff": float((s - ref).abs().max()), "prob_max_abs_diff": float((torch.softmax(s, -1) - torch.softmax(ref, -1)).abs().max()),
                       "map_delta": float(ltorank.MAP(a_got, lab) - ltorank.MAP(a_ref, lab)),
                       "rows_with_a_different_order": int((a_ref != a_got).any(1).sum()), "rows": int(a_ref.shape[0]),
                       "min_gap_of_reordered_rows": float(np.abs(np.diff(np.sort(r_ref, 1), axis=1)).min(1)[(a_ref != a_got).any(1)].min()) if (a_ref != a_got).any() else None}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    worst = {k: max(r["bf16"][k] for r in out) for k in ("score_max_abs_diff", "prob_max_abs_diff")}
    worst["abs_map_delta"] = max(abs(r["bf16"]["map_delta"]) for r in out)
    print("WORST " + json.dumps(worst))
    od = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(od):
        json.dump({"records": out, "worst_bf16": worst}, open(os.path.join(od, "bf16_error_survey_r05.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
