cd $GRAFT_REPO_ROOT
python tools/bench_gemm.py 71680 256 256
python tools/bench_gemm.py 921600 300 900
python tools/bench_gemm.py 71680 1024 300 gather
python tools/bench_gemm.py 4096 4096 4096
