cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
SHAPES=4096x12288,4096x8192 MS=33,48 timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids | cut -c1-160
