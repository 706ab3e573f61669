cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
SHAPES=4096x12288 MS=8,16,32,128 timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
