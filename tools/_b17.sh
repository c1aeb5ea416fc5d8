#!/bin/bash
cp context_attentive_ir_amd/libneuroir_hip.so /tmp/base.so
for v in V0; do
  cp gpurun_in_$v.so context_attentive_ir_amd/libneuroir_hip.so
  echo "== $v"; python -m pytest tests/test_gpu_parity.py -m gpu -q -k "duet_oracle or duet_golden or duet_fused" 2>&1 | tail -1
  python tools/_b15.py 2>&1 | tail -1
done
cp /tmp/base.so context_attentive_ir_amd/libneuroir_hip.so
