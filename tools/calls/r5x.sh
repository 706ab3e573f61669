cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; mkdir -p $O
for V in base c2; do
E=""; [ "$V" = "c2" ] && E="GPTQ_DECODE_C_MIN=256"
env $E MS=1,4 SHAPES=4096x8192,4096x6144,5120x15360,5120x5120 timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head > $O/layer_$V.txt; cat $O/layer_$V.txt
done
