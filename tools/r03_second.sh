#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/rccl_capture_probe.py > gpurun_out/rccl_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/rccl_probe.log
timeout 900 python -m pytest tests/test_gpu_envelope.py -q -m gpu > gpurun_out/envelope.log 2>&1; echo "envelope rc=$?"
tail -8 gpurun_out/envelope.log
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_envelope.py > gpurun_out/gputests.log 2>&1; echo "gpu rc=$?"
tail -12 gpurun_out/gputests.log
