cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
timeout 300 python tools/bench_engine_act_order.py 2>&1 | grep -v amdgpu.ids | tail -6
