# round 5, call 1: the batched decode path -- new tests first (fail fast), then the whole GPU suite, then the operator's times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -40 > $O/pytest_batch.txt; tail -25 $O/pytest_batch.txt
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_batch.py 2>&1 | tail -30 > $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt
timeout 400 python tools/bench_layer_decode.py > $O/layer_decode.txt 2> $O/layer_decode.err; cat $O/layer_decode.txt; tail -3 $O/layer_decode.err
