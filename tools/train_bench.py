"""Only the two training sub-records of bench.py (train_C3_cars_update, train_C2_match_tensor_update) -- a 40 s run instead of the full line -- with
optional module-level switches of autograd.py set first:  python tools/train_bench.py [NAME=VALUE ...]   e.g. PACKED_WGRAD=1 SPLIT_TRAIN_FWD=0"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from context_attentive_ir_amd import autograd as A  # noqa: E402


def main():
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        setattr(A, k, int(v, 0))
    sys.argv = ["bench.py", "--no-cpu-baseline"]
    args = bench.parse()
    env = bench.Env(1)
    head = dict(bench.CONFIGS[bench.HEADLINE])
    recs = [("train_C3_cars_update", "CARS", head), ("train_C2_match_tensor_update", "MATCH_TENSOR", dict(bench.CONFIGS["C2_match_tensor"]))]
    if os.environ.get("TRAIN_ALL"):             # the other two session models on the C3 session shape
        recs += [("train_X3_mnsrf_update", "MNSRF", dict(bench.CONFIGS["X3_mnsrf"])), ("train_X3_m_match_tensor_update", "M_MATCH_TENSOR", dict(bench.CONFIGS["X3_m_match_tensor"]))]
    for name, kind, c in recs:
        r = bench.train_record(kind, dict(c), args, env)
        print(json.dumps({"name": name, **{k: r.get(k) for k in ("ms_per_step", "hipgraph", "eager_ms_per_step", "hip_kernel_ms_per_step", "top_kernels_ms_per_step", "error")}}), flush=True)


if __name__ == "__main__":
    main()
