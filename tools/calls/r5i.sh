# round 5, call 9: where the time of the small-batch tiles goes -- per-kernel durations and PMC counters (LDS / MFMA / waits) of the C = 2 K-slice
# kernel against round 4's schedule, 4096 x 12288 at 64 and 32 rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O
for M in 64 32; do for KSC in 1 0; do
GPTQ_MM_KSC=$KSC timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/kt_${M}_$KSC -- python $R/tools/run_small_batch_once.py 4096 12288 $M > $O/kt_${M}_$KSC.txt 2>&1
ST=$(find $O/kt_${M}_$KSC -name "*kernel_stats.csv" | head -1); echo "== M $M KSC $KSC"; grep -i "stripe_mm\|stripe_gemm" "$ST" | cut -c1-200 | sed 's/gptq::(anonymous namespace):://' | tee -a $O/kernel_times.txt
rm -rf $O/kt_${M}_$KSC
done; done
for KSC in 1 0; do
GPTQ_MM_KSC=$KSC timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --kernel-trace -f csv -d $O/pmc_$KSC -- python $R/tools/run_small_batch_once.py 4096 12288 64 1 > $O/pmc_$KSC.txt 2>&1
CC=$(find $O/pmc_$KSC -name "*counter_collection.csv" | head -1)
python - "$CC" $KSC <<'PY' | tee -a $O/pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'stripe_mm' in k:
        acc[k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print('KSC', sys.argv[2], k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
rm -rf $O/pmc_$KSC
done
