# round 4, call 5: LM-head GEMV (parity + timing), routes between 1025 and 4095 rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "dense_matvec" 2>&1 | tail -3
timeout 300 python tools/bench_lm_head.py 2>&1 | grep -v amdgpu.ids > $O/lm_head.txt; cat $O/lm_head.txt
timeout 300 python -m pytest tests/test_gpu_model.py -q -x -k "engine" 2>&1 | tail -3
ALL_ROUTES=1 MS=1025,1536,2048,3072,4095 timeout 900 python tools/bench_mid_prefill.py 2>&1 | grep -v amdgpu.ids > $O/routes_1k_4k.txt; cat $O/routes_1k_4k.txt
