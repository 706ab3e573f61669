set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm8_tile or prefill_routes or fused_mlp_prefill or transpose or backward" > $O/pytest_tile.txt 2>&1; tail -5 $O/pytest_tile.txt
timeout 600 python tools/bench_gemm8_tile.py 2304 2560 3072 3584 4096 5000 8192 > $O/gemm8_tile.txt 2>&1; cat $O/gemm8_tile.txt
timeout 600 python -c "
import sys, json
sys.path[:0]=['.','gptq-for-llama_amd']
import bench
print(json.dumps(bench.prompt_leg('cuda:0')))
" > $O/prompt_leg.txt 2>&1; tail -c 6000 $O/prompt_leg.txt
