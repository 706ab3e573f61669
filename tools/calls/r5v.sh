cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_batch.py -q -m gpu -k "test_layer_decode_norm_and_residual and 4096-4096" 2>&1 | tail -2; done > $O/repeat.txt 2>&1; cat $O/repeat.txt
timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu > $O/pytest_batch.txt 2>&1; tail -4 $O/pytest_batch.txt
