cd $GRAFT_REPO_ROOT
O=gpurun_out/r5z; mkdir -p $O
for D in 0 2 3 4 6 8; do
echo "CDU $D"
GPTQ_DECODE_CDU=$D MS=1,4 SHAPES= timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep "12288\|pair"
done > $O/cdu.txt; cat $O/cdu.txt
