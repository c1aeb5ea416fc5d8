cd $GRAFT_REPO_ROOT
for s in "1120 512 1024" "1120 256 256" "112 256 1280" "112 1026 256" "448 256 256" "22400 512 1024"; do
python tools/bench_gemm.py $s
NIR_NO_GEMM16=1 python tools/bench_gemm.py $s
done
