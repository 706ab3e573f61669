# round 4, call 6: several stripes per workgroup on the same staged x (stripe_mm3_kernel C = 2 / 3) vs the K-slice schedule with forced slice counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
run() { echo "## $*" >> $O/mm3c.txt; env "$@" timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids >> $O/mm3c.txt; }
run SHAPES=4096x12288,4096x11008 MS=8,16,32,48,64 SKS=1,2,3,4
run GPTQ_MM3C=0 SHAPES=4096x12288,4096x11008 MS=8,16,32,48,64
run GPTQ_MM3C_PF=1 SHAPES=4096x12288,4096x11008 MS=32,48,64
run SHAPES=4096x12288,4096x11008 MS=96,128 SKS=1,2,3,4
run SHAPES=11008x4096 MS=32,64,128 SKS=2,4,6,8
run GPTQ_MM3C=2 SHAPES=4096x8192 MS=16,32,64
cat $O/mm3c.txt
