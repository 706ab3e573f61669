# round 4, call 7: K slices of the 128-row fused tile GEMM (33..128 rows on multi-round / long-K shapes) vs the K-slice kernel of the 16-row tiles
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
run() { echo "## $*" >> $O/sgs.txt; env "$@" timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids >> $O/sgs.txt; }
run SHAPES=4096x12288,4096x11008,11008x4096 MS=33,48,64,96,128
run GPTQ_SGS=0 SHAPES=4096x12288,4096x11008,11008x4096 MS=33,48,64,96,128
run GPTQ_SGS=2 SHAPES=4096x12288 MS=16,32,64,128 SKS=2,3,4
run GPTQ_SGS=2 SHAPES=11008x4096 MS=16,32,64,128 SKS=4,6,8,11
run GPTQ_SGS=2 SHAPES=4096x4096 MS=64,128 SKS=4,8
run SHAPES=4096x8192 MS=8,16,32,48,64
run GPTQ_MM3C=0 GPTQ_SGS=0 SHAPES=4096x8192 MS=8,16,32,48,64
cat $O/sgs.txt | cut -c1-400
