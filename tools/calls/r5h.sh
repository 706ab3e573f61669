# round 5, call 8: K slices with two stripes per wave (17 .. 64 rows on multi-round shapes / long K): parity, then A/B against round 4's schedules
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "stripe_mm or small_batches or fuzz_shapes" 2>&1 | tail -8 > $O/pytest_mm.txt; tail -4 $O/pytest_mm.txt
for KSC in 1 0; do
echo "--- GPTQ_MM_KSC=$KSC" | tee -a $O/ksc_ab.txt
GPTQ_MM_KSC=$KSC MS=17,32,48,64 SHAPES=4096x12288,4096x11008,11008x4096,4096x8192 timeout 400 python tools/bench_stripe_mm.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'M', d['M'], 'mfma_us', d.get('mfma_us'), 'relerr %.1e' % d.get('mfma_relerr', -1))" | tee -a $O/ksc_ab.txt
done
