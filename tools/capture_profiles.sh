#!/bin/bash
# Capture the rocprofv3 evidence of the bench workload on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/capture_profiles.sh'
# Outputs land in gpurun_out/prof/; tools/derive_traffic.py condenses them into profiles/.
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa trace domains).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# NIR_LSTM_MFMA_S=4: the recurrence layout the default (4 batches in flight) run uses, so the kernels are the same ones
SERIAL="env NIR_LSTM_MFMA_S=4 python $REPO/bench.py --streams 1 --no-graph --steps 50 --warmup 5 --no-cpu-baseline"
# 1. per-kernel durations, one batch in flight, eager launches (every launch attributed, kernels not stretched by overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o s -- $SERIAL > $OUT/stats_serial.log 2>&1
# 2. the default command (4 batches in flight, hipGraph replay): durations include co-running kernels of other batches
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o s -- python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats_default.log 2>&1
# 3. counters, one pass per set
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $SERIAL > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $SERIAL > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $SERIAL > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $OUT/pmc_mfma -o p -- $SERIAL > /dev/null 2>&1
# 4. the bench line of this build (default command)
python $REPO/bench.py 2>/dev/null | tail -1 > $OUT/bench_line.json
ls -R $OUT | head -40
