"""The opt-in "f32_split2" precision tier (NIR_DTYPE_F32_SPLIT2: h as ONE fp16 term in the recurrent product and the attention GEMM) against the
ORACLE, next to the default fp32-accurate path, at shapes where the tier's kernels run (>= 512 sequences of T = 64), default and trained-scale
weights.  python tools/split2_error_survey.py"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_model, cpu_state_dict  # noqa: E402
from oracle import neuroir_cpu as O  # noqa: E402
from context_attentive_ir_amd import lib, synth  # noqa: E402


def kernels_of(fn):
    L = lib.load()
    L.nir_profile_enable(1)
    fn()
    torch.cuda.synchronize()
    L.nir_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    L.nir_profile_report(buf, len(buf))
    return sorted({ln.rsplit(",", 2)[0].split("[")[0] for ln in buf.value.decode().strip().splitlines()})


def main():
    out = []
    for factor in (1.0, 2.0, 4.0, 1.0 / 64):
        for (B, S, N, QL, DL, V, seed) in [(8, 7, 10, 4, 64, 20000, 3), (2, 6, 50, 4, 64, 100000, 5)]:
            m = build_model("CARS", vocab=V, device="cuda")
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if "emb_luts" not in n:
                        p.mul_(factor)
            ex = synth.session_batch(B, S, N, QL, DL, V, seed=seed, full_length=True)
            sd = cpu_state_dict(m)
            args = (ex["source_words"], ex["source_lens"], ex["document_words"], ex["document_lens"], ex["document_labels"])
            ref = O.cars_scores(sd, *args)
            dex = {k: v.cuda() for k, v in ex.items()}
            rec = {"shape": [B, S, N, QL, DL], "vocab": V, "weight_factor": factor, "score_scale": float(ref.abs().max())}
            for dt in ("f32", "f32_split2"):
                m.compute_dtype = dt

                def run():
                    pooled, _, _ = m.encode(dex["source_words"], dex["source_lens"])
                    return m.rank_document(pooled, dex["document_words"], dex["document_lens"], dex["document_labels"], want_states=False)[0]
                s = run().cpu()
                rec[dt] = {"score_max_abs_diff_vs_oracle": float((s - ref).abs().max()), "prob_max_abs_diff": float((torch.softmax(s, -1) - torch.softmax(ref, -1)).abs().max()),
                           "kernels": [k for k in kernels_of(run) if k.startswith("lstm16") or k.startswith("attn_pool")]}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    od = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(od):
        json.dump(out, open(os.path.join(od, "split2_error_survey_r05.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
