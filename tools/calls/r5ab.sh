cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ab; mkdir -p $O
for T in 128 192 256; do
echo "MT_TILES $T"
GPTQ_SGS_MT_TILES=$T MS=129,192,256,384,512,640 timeout 600 python tools/bench_mid_prefill.py 2>/dev/null | cut -c1-150
done > $O/mt_tiles.txt; cat $O/mt_tiles.txt
