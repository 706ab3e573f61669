# round 4, call 3: early-issue variant of the tail prefetch; bench.py --gpus 2 without a launcher
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 600 python tools/bench_tail_prefetch.py 0 1 4 8 32 0 2>&1 | grep -v amdgpu.ids > $O/tail_prefetch_early_issue.txt; cat $O/tail_prefetch_early_issue.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "bench_gpus2" 2>&1 | tail -5
