set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3c}; mkdir -p $O
timeout 400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -30 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
MS=65536,4096,2048,1024 timeout 200 python tools/bench_gemm8.py 2>&1 | grep -v amdgpu.ids > $O/gemm8_run2.txt; tail -25 $O/gemm8_run2.txt
ITER=4000 timeout 200 python tools/stress_backward_small_m.py 2>&1 | grep -v amdgpu.ids > $O/stress_backward.txt; tail -4 $O/stress_backward.txt
timeout 100 ./tools/ovlab > $O/ovlab_run2.txt 2>&1; grep -c "BIT-EXACT" $O/ovlab_run2.txt
