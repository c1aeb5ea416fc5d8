#!/bin/bash
# build libneuroir variants of duet_fused.hip with extra -D flags: tools/_variants.sh V1 V2 ...
cd /root/repo/context_attentive_ir_amd
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -DDF_TIMING -DDF_$v -O3 -std=c++17 -fPIC -Wno-unused-result -c csrc/duet_fused.hip -o /tmp/duet_fused_$v.o || exit 1
  objs=$(ls csrc/_obj/*.o | grep -v duet_fused.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/gpurun_in_$v.so $objs /tmp/duet_fused_$v.o || exit 1
  echo built $v
done
