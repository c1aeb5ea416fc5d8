#!/bin/bash
# round 3, first GPU pass: envelope tests, full gpu suite, default bench, emulated 8-rank sharded CARS
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_envelope.py -x -q -m gpu > gpurun_out/envelope.log 2>&1; echo "envelope rc=$?"
tail -5 gpurun_out/envelope.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_envelope.py > gpurun_out/gputests.log 2>&1; echo "gpu rc=$?"
tail -5 gpurun_out/gputests.log
timeout 900 python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/bench_default.log
for W in 1 8; do
  BENCH_FORCE_DIST=1 BENCH_EMULATE_WORLD=$W timeout 600 python bench.py --sub none --no-cpu-baseline > gpurun_out/emu_c3_w$W.log 2>gpurun_out/emu_c3_w$W.err; echo "emu W=$W rc=$?"
  tail -c 900 gpurun_out/emu_c3_w$W.log; tail -3 gpurun_out/emu_c3_w$W.err
done
