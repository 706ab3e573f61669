set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3e}; mkdir -p $O
MS=65536,4096 timeout 300 python tools/bench_gemm8.py 2>&1 | grep -v amdgpu.ids > $O/gemm8_mfma_shapes.txt; grep -v "ok$" $O/gemm8_mfma_shapes.txt | tail -40
GPTQ_MM1_PAIR_PF=2 timeout 120 python tools/bench_pair_mm1.py 2>&1 | grep "PF=" > $O/pair_mm1.txt
GPTQ_MM1_PAIR_PF=4 timeout 120 python tools/bench_pair_mm1.py 2>&1 | grep "PF=" >> $O/pair_mm1.txt; cat $O/pair_mm1.txt
timeout 500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
