# round 4: the four-wave instances of the wave-private small-batch kernel after their x prefetch depth was cut (6 -> 2 / 1 sub-slots): default
# dispatch against GPTQ_MM3W4=0 on the multi-round shapes, 33..128 rows, 4- and 3-bit
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n_w4pf; mkdir -p $O; : > $O/w4pf.txt
run() { echo "## $*" >> $O/w4pf.txt; env "$@" timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep mfma_us | cut -c1-150 >> $O/w4pf.txt; }
run SHAPES=4096x8192,4096x12288,4096x11008 MS=33,48,49,64,65,80,96,112,128
run GPTQ_MM3W4=0 SHAPES=4096x8192,4096x12288,4096x11008 MS=33,48,49,64,65,80,96,112,128
run BITS=3 SHAPES=4096x8192,4096x12288 MS=48,64,96
run BITS=3 GPTQ_MM3W4=0 SHAPES=4096x8192,4096x12288 MS=48,64,96
cat $O/w4pf.txt | cut -c1-80
timeout 900 python -m pytest tests -q -m gpu -x -k "small_batches or stripe_mm or batches or fuzz" 2>&1 | tail -3
