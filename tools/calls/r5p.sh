set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "stripe_gemm or prefill or small_batches or layer" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
MS=129,192,256,320,384,512,640 timeout 600 python tools/bench_mid_prefill.py > $O/mid_sliced.txt 2>&1; cat $O/mid_sliced.txt
GPTQ_SGS_MT=0 MS=129,192,256,320,384,512,640 timeout 600 python tools/bench_mid_prefill.py > $O/mid_unsliced.txt 2>&1; cat $O/mid_unsliced.txt
