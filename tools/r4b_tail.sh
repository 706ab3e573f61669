# round 4, call 2: tail prefetch sweep on the LLaMA-7B decode pass
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python tools/bench_tail_prefetch.py 2>&1 | grep -v amdgpu.ids > $O/tail_prefetch.txt; cat $O/tail_prefetch.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "stripe" 2>&1 | tail -3
