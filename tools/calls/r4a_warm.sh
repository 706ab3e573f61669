# round 4, call 1: corrected warm-weights table + run-ahead prefetcher sweep (tools/warmlab.hip), then the GPU tests on the ticked kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 120 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 tools/warmlab A > $O/warmlab_A.txt 2>&1; tail -5 $O/warmlab_A.txt
timeout 300 tools/warmlab B > $O/warmlab_B.txt 2>&1; cat $O/warmlab_B.txt
timeout 300 tools/warmlab C > $O/warmlab_C.txt 2>&1; cat $O/warmlab_C.txt
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
