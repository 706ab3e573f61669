cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y; mkdir -p $O
for D in 3 5 8 11; do
echo "DU $D"
GPTQ_DECODE_DU_LONG=$D MS=1,4 SHAPES=11008x4096 timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head
done > $O/down_du.txt; cat $O/down_du.txt
