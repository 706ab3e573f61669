cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -q -k "released or long_context or attention or prepared_layer_abi" 2>&1 | tail -15 > $O/pytest.txt; tail -6 $O/pytest.txt
MS=2600,4096 timeout 300 python tools/bench_released_prefill.py > $O/released_prefill.txt 2>/dev/null; cat $O/released_prefill.txt
