"""Only bench.py's C3_cars_with_decode sub-record (full Multitask.predict: ranking + greedy decode, macro-batches of 8, 4 in flight).
python tools/decode_bench.py [--streams N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]          # e.g. --streams 1: one macro-batch in flight (isolated kernel times under rocprofv3)
    args = bench.parse()
    env = bench.Env(1)
    r = bench.decode_record(dict(bench.CONFIGS[bench.HEADLINE]), args, env)
    print(json.dumps({k: r.get(k) for k in ("ms_per_step", "pairs_per_s", "suggested_queries_per_s", "graph_predictions_equal_eager",
                                             "eager_one_in_flight_ms_per_step", "error")}))


if __name__ == "__main__":
    main()
