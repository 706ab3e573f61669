# round 5, call 10: straight-line K-slice instances (all of a slice's row blocks in flight) -- parity, A/B, kernel split
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
GPTQ_MM_KSC=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "stripe_mm or small_batches" 2>&1 | tail -5 > $O/pytest_mm.txt; tail -3 $O/pytest_mm.txt
for KSC in 2 0; do
echo "--- GPTQ_MM_KSC=$KSC" | tee -a $O/ksc_ab.txt
GPTQ_MM_KSC=$KSC MS=17,32,48,64 SHAPES=4096x12288,4096x11008,11008x4096,4096x8192,4096x4096 timeout 400 python tools/bench_stripe_mm.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'M', d['M'], 'mfma_us', d.get('mfma_us'), 'relerr %.1e' % d.get('mfma_relerr', -1))" | tee -a $O/ksc_ab.txt
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j
for M in 64 32; do
GPTQ_MM_KSC=2 timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/kt_$M -- python $R/tools/run_small_batch_once.py 4096 12288 $M > $O/kt_$M.txt 2>&1
ST=$(find $O/kt_$M -name "*kernel_stats.csv" | head -1); echo "== M $M"; grep -i "stripe_mm" "$ST" | cut -c1-160 | tee -a $O/kernel_times.txt
rm -rf $O/kt_$M
done
