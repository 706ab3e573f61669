set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
timeout 120 tools/probes/stream3 > $O/stream3.txt 2>&1; cat $O/stream3.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm8_tile or prefill_routes or fused_mlp_prefill" > $O/pytest_tile.txt 2>&1; tail -5 $O/pytest_tile.txt
timeout 600 python tools/bench_gemm8_tile.py > $O/gemm8_tile.txt 2>&1; cat $O/gemm8_tile.txt
