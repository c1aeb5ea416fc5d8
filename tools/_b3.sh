cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear or duet or cars or rnn_encoder" 2>&1 | tail -4
python tools/bench_gemm.py 71680 256 256
python tools/bench_gemm.py 921600 300 900
python tools/bench_gemm.py 71680 1024 300 gather
python tools/bench_gemm.py 4096 4096 4096
NIR_EXACT_F32=1 python tools/bench_gemm.py 71680 256 256
NIR_EXACT_F32=1 python tools/bench_gemm.py 921600 300 900
for cfg in "cars --batch 16 --cands 10" "duet --batch 64 --cands 50 --dlen 290 --qlen 8 --steps 20"; do
python bench.py --model $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['config']['ms_per_step_one_batch_in_flight'], d['roofline']['kernels_us_per_step'])"
done
