set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -q -m gpu -k "gemm or prefill or released or route or memory or layer_abi or prepared" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python -c "
import sys, json
sys.path[:0]=['.','gptq-for-llama_amd']
import bench
print(json.dumps(bench.prompt_leg('cuda:0')))
" > $O/prompt_leg.txt 2>&1; tail -c 5500 $O/prompt_leg.txt
