# round 4: four-wave / 512-register instances of the wave-private small-batch kernel (GPTQ_MM3W4=2 forces them) against the default dispatch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
run() { echo "## $*" >> $O/w4.txt; env "$@" timeout 300 python tools/bench_stripe_mm.py 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $O/w4.txt; }
run GPTQ_MM3W4=2 SHAPES=4096x4096,4096x8192,4096x12288,4096x11008 MS=32,48,64,96,128
run GPTQ_MM3W4=0 SHAPES=4096x4096,4096x8192,4096x12288,4096x11008 MS=32,48,64,96,128
cat $O/w4.txt
