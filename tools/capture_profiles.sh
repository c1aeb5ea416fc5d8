#!/bin/bash
# rocprofv3 evidence for every bench.py configuration (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/capture_profiles.sh r05 [config ...]'
# First the default `python bench.py` (its compact line + bench_detail.json are kept beside the captures); then per config:
# (1) --kernel-trace --stats of the serial eager run -- one lane, nothing co-running, every launch attributed -- with the library's batches-in-flight
#     hint set to the TIMED run's lane count (--in-flight-hint 4), so that the capture launches the same kernel instantiations the timed run does
#     (round 3 captured lstm16_pt_h2_kernel<3,2,16> for C2 while the line named <3,5,4>);
# (2) FETCH_SIZE pass; (3) WRITE_SIZE pass; (4) SQ/MFMA pass.  Counter passes use --kernel-trace only (never combined with sys/hip/hsa trace domains).
# tools/derive_profiles.py condenses the traces and FAILS when a record's dominant kernel is missing from its capture.
set -u
PREFIX=${1:-r05}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${PREFIX}prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_DETAIL=$OUT/bench_detail.json python $REPO/bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
export BENCH_NO_H2D=1
CFGS=${@:-C3_cars C1_esm C2_match_tensor NS_match_tensor_50 NS_cars_50 C4_duet C4_drmm C4_esm_hbm C5_cars_bf16 X3_mnsrf X3_m_match_tensor C3_cars_split2}
for c in $CFGS; do
  steps=20; case $c in C4_duet|C5_cars_bf16|NS_cars_50) steps=6;; esac
  RUN="python $REPO/bench.py --config $c --sub none --streams 1 --in-flight-hint 4 --no-graph --steps $steps --warmup 3 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/stats -o s -- $RUN > $OUT/$c.stats.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/$c/fetch -o p -- $RUN > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/$c/write -o p -- $RUN > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/$c/sq -o p -- $RUN > /dev/null 2>&1
  # round 6: the issue / stall counters behind DESIGN 10's stall table (own pass: kernel-trace only)
  case $c in C3_cars|C5_cars_bf16|X3_mnsrf)
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d $OUT/$c/stall -o p -- $RUN > /dev/null 2>&1;;
  esac
  echo "captured $c"
done
# keep the merged-back payload small: the per-dispatch traces are condensed on the box
python $REPO/tools/derive_profiles.py $OUT; rc=$?
python $REPO/tools/derive_profiles.py --check $OUT $OUT/bench_detail.json; rc2=$?
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
exit $((rc + rc2))
